// Host-side check of csrc/glibc_libm.h against the libm of this machine: bit-equal `log` and `pow` over the domain
// numpy's legacy gamma sampler reaches (and the subnormal / underflow tail of pow).  Built and run by
// tests/test_glibc_libm.py:  g++ -O2 -std=c++17 -ffp-contract=off -mfma glibc_libm_check.cpp -lm
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "glibc_libm.h"

static uint64_t s[2] = {0x123456789abcdefULL, 0xfedcba987654321ULL};
static inline uint64_t rnd() {
    uint64_t s1 = s[0];
    const uint64_t s0 = s[1];
    s[0] = s0;
    s1 ^= s1 << 23;
    s[1] = s1 ^ s0 ^ (s1 >> 18) ^ (s0 >> 5);
    return s[1] + s0;
}
static inline double u01() { return static_cast<double>(rnd() >> 11) * (1.0 / 9007199254740992.0); }
static inline uint64_t bits(double x) {
    uint64_t u;
    std::memcpy(&u, &x, 8);
    return u;
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? std::atol(argv[1]) : 20000000L;
    long bad_log = 0, bad_pow = 0, special = 0;
    static const double shapes[8] = {0.25, 0.1, 0.3, 0.03, 0.15, 0.5, 0.7, 0.9};
    for (long t = 0; t < n; ++t) {
        const double u = u01();
        double x;
        switch (t & 3) {
            case 0: x = 1.0 - u; break;                                        // standard_exponential's argument
            case 1: x = u * 4.0 + 1e-300; break;
            case 2: x = std::ldexp(u + 0.5, static_cast<int>(rnd() % 2000) - 1000); break;
            default: x = (1.0 - u) / shapes[(t >> 2) & 7]; break;               // the shape < 1 branch's argument
        }
        if (t % 1000003 == 0) x = std::ldexp(u + 0.5, -1060);                    // subnormal
        const double a = std::log(x), b = mz::libm::glibc_log(x);
        if (bits(a) != bits(b)) {
            if (bad_log < 5) std::printf("log(%a): libm %a here %a\n", x, a, b);
            ++bad_log;
        }
        const double shape = shapes[(t >> 2) & 7];
        const double y = 1. / shape;
        double px;
        switch (t & 3) {
            case 0: px = u; break;                                              // pow(U, 1 / shape)
            case 1: px = 1.0 - shape + shape * -std::log((1 - u * shape) / shape > 0 ? (u * shape) : 0.5); break;
            case 2: px = std::ldexp(u + 0.5, -static_cast<int>(rnd() % 60)); break;   // tiny bases: subnormal results
            default: px = u * 8.0; break;
        }
        if (t % 999983 == 0) px = 0.0;
        const double c = std::pow(px, y), d = mz::libm::glibc_pow(px, y);
        if (c < 1e-200) ++special;
        if (bits(c) != bits(d)) {
            if (bad_pow < 5) std::printf("pow(%a, %a): libm %a here %a\n", px, y, c, d);
            ++bad_pow;
        }
    }
    std::printf("{\"samples\": %ld, \"log_mismatches\": %ld, \"pow_mismatches\": %ld, \"pow_results_below_1e-200\": %ld}\n", n,
                bad_log, bad_pow, special);
    return (bad_log || bad_pow) ? 1 : 0;
}
