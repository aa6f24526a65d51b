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

// Per-move control block of the whole-move kernels.  A plain search passes the noise / rng_skip rows and
// leaves the rest null.  A BATCH of moves queued back to back without host round trips
// (mzmcts_moves_*, include/mzmcts.h) additionally lets the kernel
//   * sample the move's action itself (SelfPlay.select_action, self_play.py:223-246) from the tree's RNG
//     stream, so the next kernels (environment step, next search) can follow without the host;
//   * refuse to search an env whose pre-drawn exploration noise has become invalid: the host draws the
//     noise of move m+1 before move m has run, assuming how many RNG words search m spends on tie-breaks
//     (they come ahead of the Dirichlet draw in the stream); when it spent a different number, the env
//     stalls -- sticky flag, nothing consumed, nothing written -- and the host redraws from the true stream
//     position after collecting the batch;
//   * write the move's results into that move's slot of an output ring.
struct MoveCtl {
    const double* noise;          // [E][A] Dirichlet rows of this move, or null
    const uint32_t* rng_skip;     // [E] words the host consumed since the device copy last moved
    const double* temperature;    // [E]; null = the host samples the action
    const int32_t* move_limit;    // [E] first move of the batch that must NOT be searched; null = no limit
    uint8_t* stall;               // [E] sticky within a batch; null = never stall
    int32_t move_index;
    const uint32_t* expected_ties; // [E] tie-break words the previous search of the batch was assumed to consume;
                                  //     null = this is the first move of the batch (nothing was assumed)
    // this move's slot of the output ring (any may be null)
    int32_t* actions;             // [E] sampled action (-1: not searched)
    int32_t* visits;              // [E][A] root children visit counts, by child slot
    double* root_value_sum;       // [E]
    float* root_predicted;        // [E]
    int32_t* max_depth;           // [E]
    uint32_t* tie_words;          // [E] RNG words the search consumed (tie-breaks)
    uint32_t* sample_words;       // [E] RNG words the action sampling consumed
    int32_t* depth_sum;           // [E] sum of the S descent depths
    // play_game's temperature rule (self_play.py:152-158): temperature only while len(action_history) < threshold, i.e.
    // while fewer than threshold - 1 moves of the env's current game were played; afterwards the best action (T = 0)
    int32_t* game_moves;          // [E] moves played in env e's current game (the kernel counts its own move); null = no rule
    int32_t temperature_threshold; // 0 = none
};

// The temperature select_action is called with for env e's move (see MoveCtl::game_moves).
__device__ __forceinline__ double move_temperature(const MoveCtl& ctl, int e) {
    const double t = ctl.temperature[e];
    if (ctl.game_moves && ctl.temperature_threshold > 0 && ctl.game_moves[e] + 1 >= ctl.temperature_threshold) return 0.0;
    return t;
}

// Optional extras of move_inputs_kernel (mcts_kernels.hip): a second copy of the root inputs into the engine's own arrays
// (lock-step moves) and the per-game move counters of the temperature-threshold rule.
struct MoveInputsExtra {
    int32_t* own_legal = nullptr;       // [E][A]
    int32_t* own_nlegal = nullptr;      // [E]
    int32_t* own_to_play = nullptr;     // [E]
    int32_t* game_moves = nullptr;      // [E]
    const uint8_t* finished = nullptr;  // [E] the env kernels' `done` of the move before, or null
};

// Group-uniform: true = this env must not be searched in this move (see MoveCtl).
__device__ __forceinline__ bool move_stalled(const TreeParams& p, const MoveCtl& ctl, int e, int j) {
    if (!ctl.stall) return false;
    bool stalled = ctl.stall[e] != 0;
    if (!stalled && ctl.expected_ties && p.tie_words[e] != ctl.expected_ties[e]) stalled = true;
    if (!stalled && ctl.move_limit && ctl.move_index >= ctl.move_limit[e]) stalled = true;
    if (stalled && j == 0) {
        ctl.stall[e] = 1;
        if (ctl.actions) ctl.actions[e] = -1;
    }
    return stalled;
}

// Temperatures whose visit_count ** (1 / T) is exact integer arithmetic everywhere: 1 / T an integer k in 1..4
// (the reference's schedules use 1, 0.5, 0.25: cartpole.py:118-128).  Returns k, or 0.
__host__ __device__ inline int exact_inverse_temperature(double temperature) {
    const double inv = 1.0 / temperature;
    const int k = static_cast<int>(inv);
    return (inv == static_cast<double>(k) && k >= 1 && k <= 4) ? k : 0;
}

// SelfPlay.select_action (self_play.py:223-246) on the device copy of the stream, for the temperatures
// whose arithmetic is exact everywhere: 0 (arg-max), +inf (choice without p) and 1/k (visits ** k by
// multiplication -- libm's pow returns the same exactly representable integers --, normalise, cumulative
// sum, one legacy double, right-bisect: RandomState.choice(p=...)).
// Returns the chosen child slot, or -2 for a temperature the host must handle (pow()).
template <typename VisitOf>
__device__ __forceinline__ int device_select_action(VisitOf visit_of, int n, double temperature, uint32_t* key,
                                                    int32_t* pos, uint32_t* words) {
    if (temperature == 0.0) {
        int best = 0, best_v = visit_of(0);
        for (int i = 1; i < n; ++i) {
            const int v = visit_of(i);
            if (v > best_v) {
                best_v = v;
                best = i;
            }
        }
        return best;
    }
    if (isinf(temperature)) return static_cast<int>(mt_below(key, pos, static_cast<uint32_t>(n), words));
    const int k = exact_inverse_temperature(temperature);
    if (k == 0) return -2;
    auto weight = [&](int i) {
        const double v = static_cast<double>(visit_of(i));
        double w = v;
        for (int r = 1; r < k; ++r) w = w * v;
        return w;
    };
    double total = 0.0;
    for (int i = 0; i < n; ++i) total = total + weight(i);
    double total_p = 0.0;
    for (int i = 0; i < n; ++i) total_p += weight(i) / total;
    const int32_t a = static_cast<int32_t>(mt_next(key, pos) >> 5);
    const int32_t b = static_cast<int32_t>(mt_next(key, pos) >> 6);
    *words += 2u;
    const double u = (a * 67108864.0 + b) / 9007199254740992.0;
    double run = 0.0;
    int idx = 0;
    for (; idx < n; ++idx) {
        run += weight(idx) / total;
        if (!(run / total_p <= u)) break;
    }
    return idx;
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
