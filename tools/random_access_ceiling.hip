// random_access_ceiling.hip -- what the memory system of this GPU sustains for the access pattern of the
// lock-step tree kernels: 64-byte records at random addresses of a multi-GiB table.
//
//   independent   every lane group issues UNROLL record loads at once (addresses known up front): the ceiling
//                 for randomly placed 64-byte reads with as much memory-level parallelism as the GPU holds
//   chained       every lane group walks a chain: the address of hop h+1 comes out of the record of hop h,
//                 as a tree descent's does (child index -> next record); parallelism = resident lane groups
//
// A record is read the way select reads one: a lane group of 4 lanes, 16 bytes per lane.  Stand-alone program
// (no library, no torch):   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/random_access_ceiling tools/random_access_ceiling.hip
//                           tools/_bin/random_access_ceiling [table GiB = 8] [hops = 9]
// prints one JSON line per pattern.  Used for DESIGN.md section 5 (measured ceiling beside select / expand_backup).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(call)                                                                              \
    do {                                                                                         \
        hipError_t err__ = (call);                                                               \
        if (err__ != hipSuccess) {                                                               \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(err__));                   \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {  // splitmix64 finaliser
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

// every record's first 8 bytes: the index of the record a chain visits next
__global__ void fill_kernel(uint4* table, uint64_t n_records) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_records * 4; i += stride) {
        const uint64_t rec = i >> 2;
        const uint64_t next = mix(rec) % n_records;
        table[i] = make_uint4(static_cast<uint32_t>(next), static_cast<uint32_t>(next >> 32), static_cast<uint32_t>(i), 0u);
    }
}

constexpr int kUnroll = 8;

__global__ __launch_bounds__(256) void independent_kernel(const uint4* __restrict__ table, uint64_t n_records, int rounds,
                                                          uint32_t* __restrict__ sink) {
    const uint64_t group = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 2;
    const int j = threadIdx.x & 3;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const uint64_t rec = mix(group * 0x10001ull + static_cast<uint64_t>(r) * kUnroll + u) % n_records;
            v[u] = table[rec * 4 + j];
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) acc ^= v[u].x ^ v[u].z;
    }
    if (acc == 0x12345678u) sink[0] = acc;  // keeps the loads alive
}

// G lanes per chain, 64 / G bytes per lane (select at 2 actions: G = 2, two 16-byte loads per lane)
template <int G>
__global__ __launch_bounds__(256) void chained_kernel(const uint4* __restrict__ table, uint64_t n_records, int hops,
                                                      uint32_t* __restrict__ sink) {
    const uint64_t group = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) / G;
    const int j = threadIdx.x % G;
    const int leader = (threadIdx.x & 63) & ~(G - 1);
    uint64_t rec = mix(group) % n_records;
    uint32_t acc = 0;
    for (int h = 0; h < hops; ++h) {
        uint4 v[4 / G];
#pragma unroll
        for (int q = 0; q < 4 / G; ++q) v[q] = table[rec * 4 + j * (4 / G) + q];
#pragma unroll
        for (int q = 0; q < 4 / G; ++q) acc ^= v[q].z;
        // lane 0 of the group holds the link; hand it to the others (as select hands on the arg-max child)
        const uint32_t lo = __shfl(v[0].x, leader, 64);
        const uint32_t hi = __shfl(v[0].y, leader, 64);
        rec = (static_cast<uint64_t>(hi) << 32) | lo;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int G>
static int run_chained(const uint4* table, uint64_t n_records, int hops, uint32_t* sink, double gib, int shift, hipEvent_t a,
                       hipEvent_t b) {
    const uint64_t groups = 1ull << shift;
    const dim3 grid(static_cast<unsigned>(groups * G / 256));
    chained_kernel<G><<<grid, dim3(256)>>>(table, n_records, hops, sink);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a));
        chained_kernel<G><<<grid, dim3(256)>>>(table, n_records, hops, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double bytes = static_cast<double>(groups) * hops * 64.0;
    std::printf("{\"pattern\": \"chained\", \"lanes_per_chain\": %d, \"table_gib\": %.2f, \"chains\": %.0f, \"hops\": %d, "
                "\"ms\": %.4f, \"GB_per_s\": %.1f}\n",
                G, gib, static_cast<double>(groups), hops, best, bytes / (best * 1e-3) / 1e9);
    return 0;
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? std::atof(argv[1]) : 8.0;
    const int hops = argc > 2 ? std::atoi(argv[2]) : 9;
    const uint64_t n_records = static_cast<uint64_t>(gib * 1024.0 * 1024.0 * 1024.0) / 64;
    uint4* table = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc(&table, n_records * 64));
    CHECK(hipMalloc(&sink, 64));
    fill_kernel<<<dim3(8192), dim3(256)>>>(table, n_records);
    CHECK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));

    // independent: 2^22 lane groups x 16 records
    {
        const uint64_t groups = 1ull << 22;
        const int rounds = 2;
        const dim3 grid(static_cast<unsigned>(groups * 4 / 256));
        independent_kernel<<<grid, dim3(256)>>>(table, n_records, rounds, sink);  // warm-up (TLB, clocks)
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(a));
            independent_kernel<<<grid, dim3(256)>>>(table, n_records, rounds, sink);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        const double bytes = static_cast<double>(groups) * rounds * kUnroll * 64.0;
        std::printf("{\"pattern\": \"independent\", \"table_gib\": %.2f, \"records\": %.0f, \"ms\": %.4f, \"GB_per_s\": %.1f}\n", gib,
                    bytes / 64.0, best, bytes / (best * 1e-3) / 1e9);
    }
    // chained: 2^20 chains (the lock-step measurement's tree count) and 2^22 (every CU full for the whole launch)
    for (int shift : {20, 22}) {
        if (run_chained<2>(table, n_records, hops, sink, gib, shift, a, b)) return 1;
        if (run_chained<4>(table, n_records, hops, sink, gib, shift, a, b)) return 1;
    }
    CHECK(hipFree(table));
    CHECK(hipFree(sink));
    return 0;
}
