// kernel_common.h -- helpers shared by the kernel translation units (mcts_kernels.hip, fused_narrow.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "np_legacy_rng.h"
#include "tree_layout.h"

namespace mz {

// Diagnostic build only (-DMZ_STAMPS, tools/stamp_fused.py): per-phase cycle sums of the fused kernel,
// written to a debug buffer that nothing else reads.  The production library never defines it.
#ifdef MZ_STAMPS
static __device__ unsigned long long g_stamp_sums[16];  // one copy per translation unit
#define MZ_STAMP_DECL unsigned long long stamp_prev = __builtin_readcyclecounter(), stamp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define MZ_STAMP(slot)                                                   \
    do {                                                                 \
        const unsigned long long now__ = __builtin_readcyclecounter();   \
        stamp_acc[slot] += now__ - stamp_prev;                           \
        stamp_prev = now__;                                              \
    } while (0)
#define MZ_STAMP_FLUSH                                                                        \
    do {                                                                                      \
        if (threadIdx.x == 0)                                                                 \
            for (int s__ = 0; s__ < 16; ++s__) atomicAdd(&g_stamp_sums[s__], stamp_acc[s__]);  \
    } while (0)
#else
#define MZ_STAMP_DECL
#define MZ_STAMP(slot)
#define MZ_STAMP_FLUSH
#endif

// per-search reset shared by the root kernels (MinMaxStats(), max_tree_depth; self_play.py:317-319)
__device__ __forceinline__ void reset_search_state(const TreeParams& p, int e, const uint32_t* rng_skip) {
    // advance the device RNG past the words the host mirror consumed (Dirichlet draw of this
    // move, action sampling of the previous one)
    const uint32_t skip = rng_skip ? rng_skip[e] : 0u;
    if (skip) {
        uint32_t* key = p.mt_key + static_cast<size_t>(e) * kMtN;
        int32_t pos = p.mt_pos[e];
        for (uint32_t i = 0; i < skip; ++i) (void)mt_next(key, &pos);
        p.mt_pos[e] = pos;
    }
    p.min_max[e] = MinMax{INFINITY, -INFINITY};
    p.root_value_sum[e] = 0.0;
    p.max_depth[e] = 0;
    p.depth_sum[e] = 0;
    p.tie_words[e] = 0u;
    p.path_len[e] = 0;
    p.leaf_parent[e] = 0;
}

// Launch with optional HIP events bound to the dispatch itself (hipExtLaunchKernel records the
// kernel's own start / end timestamps into the events, so profiling-mode timings are kernel
// durations, comparable with rocprofv3's, not launch-to-launch gaps).
template <typename Kernel, typename... Args>
static inline void launch_kernel(Kernel kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                          const LaunchTiming* timing, Args... args) {
    if (timing && timing->start)
        hipExtLaunchKernelGGL(kernel, grid, block, static_cast<uint32_t>(lds), stream, timing->start, timing->stop, 0u,
                              args...);
    else
        kernel<<<grid, block, lds, stream>>>(args...);
}


}  // namespace mz
