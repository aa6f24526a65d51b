// Host-side check of np_legacy_rng.h DeviceStream (the Dirichlet sampler the GPU runs: glibc's log / pow restated,
// csrc/glibc_libm.h) against HostStream (the same sampler on this machine's libm, pinned to numpy by fixture G7):
// identical rows, word counts and stream positions over many seeds, shapes and row lengths.  Built and run by
// tests/test_glibc_libm.py:  g++ -O2 -std=c++17 -ffp-contract=off -mfma device_stream_check.cpp -lm
#include <cstdio>
#include <cstring>

#include "np_legacy_rng.h"

int main() {
    static const double alphas[] = {0.25, 0.1, 0.3, 0.03, 0.5, 0.9, 1.0};
    static const int lengths[] = {2, 4, 7, 9, 18, 121};
    long rows = 0, bad = 0;
    for (unsigned seed = 0; seed < 400; ++seed) {
        for (double alpha : alphas) {
            for (int k : lengths) {
                mz::HostStream host;
                host.seed(seed * 2654435761u + 17u);
                uint32_t key[mz::kMtN];
                std::memcpy(key, host.key, sizeof(key));
                mz::DeviceStream dev{key, host.pos, 0u, nullptr, 0, 0};
                double want[121], got[121];
                for (int draw = 0; draw < 3; ++draw) {                 // (three draws: the 624-word block is regenerated on the way)
                    const uint64_t before = host.words;
                    host.dirichlet(alpha, k, want);
                    const uint32_t dev_before = dev.words;
                    dev.dirichlet(alpha, k, got);
                    ++rows;
                    if (std::memcmp(want, got, sizeof(double) * k) != 0 || host.words - before != dev.words - dev_before ||
                        host.pos != dev.pos || std::memcmp(host.key, key, sizeof(key)) != 0) {
                        if (bad < 5) std::printf("seed %u alpha %g k %d draw %d differs\n", seed, alpha, k, draw);
                        ++bad;
                    }
                }
            }
        }
    }
    std::printf("{\"rows\": %ld, \"mismatches\": %ld}\n", rows, bad);
    return bad ? 1 : 0;
}
