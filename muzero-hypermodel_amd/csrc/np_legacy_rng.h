// np_legacy_rng.h -- numpy legacy RandomState streams, one per search tree.
//
// The reference draws all of its randomness from numpy's global legacy generator
// (self_play.py:22 seed; :474 dirichlet; :372 tie-break choice; :237,:244 action sampling; :217
// random opponent).  To give tree e the exact behaviour of reference worker `config.seed + e`
// each tree owns a clone of that generator:
//   * the MT19937 core (seeding, twist, tempering) and the bounded-integer draw used by
//     `numpy.random.choice(list)` are integer-only and run on BOTH host and device (the select
//     kernel breaks UCB ties on-device);
//   * everything that needs libm transcendentals (gamma / Dirichlet) or visit-count powers
//     (select_action) runs on the host against the same glibc libm numpy itself calls, on a host
//     mirror of the stream that is kept in step with the device copy by word counts;
//   * the Dirichlet draw for shape <= 1 (every reference config: root_dirichlet_alpha 0.1 ... 0.3) ALSO runs on the
//     device (DeviceStream below, on glibc's log / pow restated in glibc_libm.h), so that a batch of moves needs
//     no pre-drawn noise from the host.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "glibc_libm.h"

#if defined(__HIPCC__)
#define MZ_HD __host__ __device__
#else
#define MZ_HD
#endif

namespace mz {

constexpr int kMtN = 624;
constexpr int kMtM = 397;

// ---- MT19937 core over caller-provided storage (key[624] + pos) -------------------------------
MZ_HD inline void mt_seed(uint32_t* key, int32_t* pos, uint32_t seed) {
    // numpy.random.seed(int) -> _legacy_seeding -> mt19937_seed (Knuth's init_genrand)
    uint32_t prev = seed;
    key[0] = prev;
    for (int i = 1; i < kMtN; ++i) {
        prev = 1812433253u * (prev ^ (prev >> 30)) + static_cast<uint32_t>(i);
        key[i] = prev;
    }
    *pos = kMtN;
}

MZ_HD inline uint32_t mt_mix(uint32_t hi, uint32_t lo, uint32_t far) {
    const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

MZ_HD inline void mt_regenerate(uint32_t* key) {
    int k = 0;
    for (; k < kMtN - kMtM; ++k) key[k] = mt_mix(key[k], key[k + 1], key[k + kMtM]);
    for (; k < kMtN - 1; ++k) key[k] = mt_mix(key[k], key[k + 1], key[k + kMtM - kMtN]);
    key[kMtN - 1] = mt_mix(key[kMtN - 1], key[0], key[kMtM - 1]);
}

MZ_HD inline uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

MZ_HD inline uint32_t mt_next(uint32_t* key, int32_t* pos) {
    int32_t p = *pos;
    if (p >= kMtN) {
        mt_regenerate(key);
        p = 0;
    }
    const uint32_t y = key[p];
    *pos = p + 1;
    return mt_temper(y);
}

// smallest 2^m - 1 >= v
MZ_HD inline uint32_t mask_for(uint32_t v) {
    v |= v >> 1;
    v |= v >> 2;
    v |= v >> 4;
    v |= v >> 8;
    v |= v >> 16;
    return v;
}

// RandomState.randint(0, n) as reached from choice(list-of-n): masked rejection on 32-bit words;
// n == 1 consumes nothing.  `words` counts the draws.
MZ_HD inline uint32_t mt_below(uint32_t* key, int32_t* pos, uint32_t n, uint32_t* words) {
    const uint32_t top = n - 1u;
    if (top == 0u) return 0u;
    const uint32_t mask = mask_for(top);
    uint32_t v;
    do {
        v = mt_next(key, pos) & mask;
        ++*words;
    } while (v > top);
    return v;
}

// ---- the legacy distributions of the exploration noise over caller-provided MT19937 storage ------------
// (usable on the device: log / pow are glibc's, restated; the same code runs on the host in the tests)
struct DeviceStream {
    uint32_t* key;
    int32_t pos;
    uint32_t words;   // 32-bit words drawn through this object
    // optional copy of key[win_lo, win_hi) in fast memory (the device kernel stages the words a draw is likely to
    // consume in LDS with all loads in flight, instead of one dependent global load per word)
    const uint32_t* window = nullptr;
    int32_t win_lo = 0, win_hi = 0;
    MZ_HD uint32_t next_word() {
        if (pos >= win_lo && pos < win_hi) {
            const uint32_t y = window[pos - win_lo];
            ++pos;
            return mt_temper(y);
        }
        if (pos >= kMtN) win_hi = 0;  // the regeneration rewrites the block the window was copied from
        return mt_next(key, &pos);
    }
    MZ_HD double uniform() {  // legacy_double: 53 random bits from two words
        const int32_t a = static_cast<int32_t>(next_word() >> 5);
        const int32_t b = static_cast<int32_t>(next_word() >> 6);
        words += 2u;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    MZ_HD double exponential() { return -libm::glibc_log(1.0 - uniform()); }
    // legacy_standard_gamma for shape <= 1 (the shape > 1 branch needs legacy_gauss and its cached value: host only)
    MZ_HD double gamma_le1(double shape) {
        if (shape == 1.0) return exponential();
        if (shape == 0.0) return 0.0;
        for (;;) {
            const double u = uniform();
            const double v = exponential();
            if (u <= 1.0 - shape) {
                const double x = libm::glibc_pow(u, 1. / shape);
                if (x <= v) return x;
            } else {
                const double y = -libm::glibc_log((1 - u) / shape);
                const double x = libm::glibc_pow(1.0 - shape + shape * y, 1. / shape);
                if (x <= (v + y)) return x;
            }
        }
    }
    // RandomState.dirichlet([alpha] * k), alpha <= 1
    MZ_HD void dirichlet(double alpha, int k, double* out) {
        double acc = 0.0;
        for (int j = 0; j < k; ++j) {
            out[j] = gamma_le1(alpha);
            acc = acc + out[j];
        }
        const double inv = 1 / acc;
        for (int j = 0; j < k; ++j) out[j] = out[j] * inv;
    }
};

// ---- host stream: the full legacy distribution set the path needs -----------------------------
struct HostStream {
    uint32_t key[kMtN];
    int32_t pos = kMtN;
    int32_t has_gauss = 0;
    double gauss = 0.0;
    uint64_t words = 0;  // 32-bit words drawn since construction / seeding
    // Speculative draws (mzmcts_moves_prepare) must be undoable: while `twist_backup` is set, the block of
    // 624 words is copied there the first time it is about to be regenerated (with the word count at that
    // moment), which is all a later restore to ANY earlier or later position needs.
    uint32_t* twist_backup = nullptr;
    bool twisted = false;
    uint64_t twist_words = 0;

    void seed(uint32_t s) {
        mt_seed(key, &pos, s);
        has_gauss = 0;
        gauss = 0.0;
        words = 0;
    }
    uint32_t u32() {
        if (pos == kMtN && twist_backup && !twisted) {
            std::memcpy(twist_backup, key, sizeof(key));
            twisted = true;
            twist_words = words;
        }
        ++words;
        return mt_next(key, &pos);
    }
    // n draws whose values nobody looks at: what remains of the current block of 624 words is stepped over, further
    // blocks are regenerated and stepped over (the state n calls of u32() leave behind, without tempering n words)
    void skip(uint64_t n) {
        while (n > 0) {
            if (pos >= kMtN) {
                if (twist_backup && !twisted) {
                    std::memcpy(twist_backup, key, sizeof(key));
                    twisted = true;
                    twist_words = words;
                }
                mt_regenerate(key);
                pos = 0;
            }
            const uint64_t step = n < static_cast<uint64_t>(kMtN - pos) ? n : static_cast<uint64_t>(kMtN - pos);
            pos += static_cast<int32_t>(step);
            words += step;
            n -= step;
        }
    }
    // legacy_double: 53 random bits from two words
    double uniform() {
        const int32_t a = static_cast<int32_t>(u32() >> 5);
        const int32_t b = static_cast<int32_t>(u32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    uint32_t below(uint32_t n) {
        uint32_t w = 0;
        const uint32_t v = mt_below(key, &pos, n, &w);
        words += w;
        return v;
    }
    double exponential() { return -std::log(1.0 - uniform()); }
    double normal() {  // legacy_gauss: polar Box-Muller with a one-value cache
        if (has_gauss) {
            const double t = gauss;
            has_gauss = 0;
            gauss = 0.0;
            return t;
        }
        double x1, x2, r2;
        do {
            x1 = 2.0 * uniform() - 1.0;
            x2 = 2.0 * uniform() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        const double f = std::sqrt(-2.0 * std::log(r2) / r2);
        gauss = f * x1;
        has_gauss = 1;
        return f * x2;
    }
    double gamma(double shape) {  // legacy_standard_gamma
        if (shape == 1.0) return exponential();
        if (shape == 0.0) return 0.0;
        if (shape < 1.0) {
            for (;;) {
                const double u = uniform();
                const double v = exponential();
                if (u <= 1.0 - shape) {
                    const double x = std::pow(u, 1. / shape);
                    if (x <= v) return x;
                } else {
                    const double y = -std::log((1 - u) / shape);
                    const double x = std::pow(1.0 - shape + shape * y, 1. / shape);
                    if (x <= (v + y)) return x;
                }
            }
        }
        const double b = shape - 1. / 3.;
        const double c = 1. / std::sqrt(9 * b);
        for (;;) {
            double x, v;
            do {
                x = normal();
                v = 1.0 + c * x;
            } while (v <= 0.0);
            v = v * v * v;
            const double u = uniform();
            if (u < 1.0 - 0.0331 * (x * x) * (x * x)) return b * v;
            if (std::log(u) < 0.5 * x * x + b * (1. - v + std::log(v))) return b * v;
        }
    }
    // RandomState.dirichlet([alpha] * k)
    void dirichlet(double alpha, int k, double* out) {
        double acc = 0.0;
        for (int j = 0; j < k; ++j) {
            out[j] = gamma(alpha);
            acc = acc + out[j];
        }
        const double inv = 1 / acc;
        for (int j = 0; j < k; ++j) out[j] = out[j] * inv;
    }
    // RandomState.choice(n, p=p): normalised cumulative sum, one uniform, right-bisect
    int choice_p(const double* p, int n) {
        double total = 0.0;
        for (int i = 0; i < n; ++i) total += p[i];
        const double u = uniform();
        double run = 0.0;
        int idx = 0;
        for (; idx < n; ++idx) {
            run += p[idx];
            if (!(run / total <= u)) break;
        }
        return idx;
    }
    // SelfPlay.select_action (self_play.py:223-246) -> chosen slot
    int select_action(const int32_t* visits, int n, double temperature) {
        if (temperature == 0) {
            int best = 0;
            for (int i = 1; i < n; ++i)
                if (visits[i] > visits[best]) best = i;
            return best;
        }
        if (std::isinf(temperature)) return static_cast<int>(below(static_cast<uint32_t>(n)));
        double stack_buf[64];
        double* w = n <= 64 ? stack_buf : new double[n];
        double total = 0;
        for (int i = 0; i < n; ++i) {
            w[i] = std::pow(static_cast<double>(visits[i]), 1 / temperature);
            total = total + w[i];
        }
        for (int i = 0; i < n; ++i) w[i] = w[i] / total;
        const int pick = choice_p(w, n);
        if (w != stack_buf) delete[] w;
        return pick;
    }
};
}  // namespace mz
