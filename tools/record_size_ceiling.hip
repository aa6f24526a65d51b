// record_size_ceiling.hip -- how the random-access rate of this GPU's memory system depends on the size of the
// record: R-byte records (R = 16 ... 512, 16 bytes per lane, R/16 adjacent lanes per record) at random R-aligned
// (16- and 32-byte records: 64-byte-aligned) places of a multi-GiB table, addresses known up front, 8 records in
// flight per lane group.  Also read-modify-write of the same records (what a backup does to a path node) and plain
// stores of them with no load before (what a backup does when the descent has handed it the records).
// Decides the child-block geometry of the lock-step tree kernels: is the limit requests per
// second (then bigger useful records win) or bytes per second (then smaller sectors win)?
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/record_size_ceiling tools/record_size_ceiling.hip
//   tools/_bin/record_size_ceiling [table GiB = 8]          one JSON line per (pattern, record size)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(call)                                                            \
    do {                                                                       \
        hipError_t err__ = (call);                                             \
        if (err__ != hipSuccess) {                                             \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(err__)); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {  // splitmix64 finaliser
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

constexpr int kUnroll = 8;

// LANES lanes per record (16 B each); records sit at multiples of SLOT bytes (SLOT >= 16 * LANES).
template <int LANES, int SLOT, int MODE>  // MODE 0 = read, 1 = read-modify-write, 2 = write only
__global__ __launch_bounds__(256) void record_kernel(uint4* __restrict__ table, uint64_t n_slots, int rounds,
                                                     uint32_t* __restrict__ sink) {
    const uint64_t group = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) / LANES;
    const int j = threadIdx.x % LANES;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint4 v[kUnroll];
        uint64_t at[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const uint64_t slot = mix(group * 0x10001ull + static_cast<uint64_t>(r) * kUnroll + u) % n_slots;
            at[u] = slot * (SLOT / 16) + j;
            if (MODE != 2) v[u] = table[at[u]];
            else v[u] = uint4{static_cast<uint32_t>(slot), 1u, static_cast<uint32_t>(r), 0u};
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            acc ^= v[u].x ^ v[u].z;
            if (MODE != 0) {
                v[u].y += 1u;
                table[at[u]] = v[u];
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;  // keeps the loads alive
}

template <int LANES, int SLOT, int MODE>
static int run(uint4* table, uint64_t table_bytes, uint32_t* sink, hipEvent_t a, hipEvent_t b) {
    const uint64_t n_slots = table_bytes / SLOT;
    const uint64_t groups = 1ull << 22;
    const int rounds = 2;
    const dim3 grid(static_cast<unsigned>(groups * LANES / 256));
    record_kernel<LANES, SLOT, MODE><<<grid, dim3(256)>>>(table, n_slots, rounds, sink);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a));
        record_kernel<LANES, SLOT, MODE><<<grid, dim3(256)>>>(table, n_slots, rounds, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double records = static_cast<double>(groups) * rounds * kUnroll;
    const double bytes = records * 16.0 * LANES * (MODE == 1 ? 2.0 : 1.0);
    std::printf("{\"pattern\": \"%s\", \"record_bytes\": %d, \"slot_bytes\": %d, \"records\": %.0f, \"ms\": %.4f, "
                "\"useful_GB_per_s\": %.1f, \"G_records_per_s\": %.2f}\n",
                MODE == 1 ? "read_modify_write" : (MODE == 2 ? "write" : "read"), 16 * LANES, SLOT, records, best, bytes / (best * 1e-3) / 1e9,
                records / (best * 1e-3) / 1e9);
    std::fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? std::atof(argv[1]) : 8.0;
    const uint64_t table_bytes = static_cast<uint64_t>(gib * 1024.0 * 1024.0 * 1024.0);
    uint4* table = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc(&table, table_bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(table, 1, table_bytes));
    CHECK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    if (run<1, 64, 0>(table, table_bytes, sink, a, b)) return 1;
    if (run<2, 64, 0>(table, table_bytes, sink, a, b)) return 1;
    if (run<4, 64, 0>(table, table_bytes, sink, a, b)) return 1;
    if (run<8, 128, 0>(table, table_bytes, sink, a, b)) return 1;
    if (run<16, 256, 0>(table, table_bytes, sink, a, b)) return 1;
    if (run<32, 512, 0>(table, table_bytes, sink, a, b)) return 1;
    if (run<1, 64, 1>(table, table_bytes, sink, a, b)) return 1;
    if (run<2, 64, 1>(table, table_bytes, sink, a, b)) return 1;
    if (run<4, 64, 1>(table, table_bytes, sink, a, b)) return 1;
    if (run<8, 128, 1>(table, table_bytes, sink, a, b)) return 1;
    if (run<1, 64, 2>(table, table_bytes, sink, a, b)) return 1;
    if (run<2, 64, 2>(table, table_bytes, sink, a, b)) return 1;
    if (run<4, 64, 2>(table, table_bytes, sink, a, b)) return 1;
    if (run<8, 128, 2>(table, table_bytes, sink, a, b)) return 1;
    CHECK(hipFree(table));
    CHECK(hipFree(sink));
    return 0;
}
