// mcts_kernels.hip -- batched MCTS kernels for CDNA4 (gfx950, wave64).
//
// Mapping: a tree is owned by a group of G adjacent lanes of ONE wavefront (G = pow2 >= min(A,64),
// or wider on request), one lane per child (lanes loop when A > 64).  A 64-thread workgroup (one
// wave) carries 64/G trees; tree -> workgroup is identical in every kernel, so a tree's blocks are
// re-touched from the same XCD (workgroups are dealt round-robin over the 8 XCDs) and stay in that
// XCD's L2.
//
// Two execution shapes share the device functions of tree_device.h / fc_net_device.h:
//   lock-step   select_kernel -> (any inference engine, e.g. PyTorch-ROCm) -> expand_backup_kernel,
//               tree in HBM; general networks.
//   fused       search_fused_fc_kernel: one launch per MOVE for fully-connected networks -- every
//               workgroup keeps its trees (child blocks, path, hidden states, activations) and the
//               network weights in LDS and runs root inference + all S simulations without leaving
//               the CU; trees never synchronise with each other because they are independent.
//
// Arithmetic contract (bit-exact with the reference's Python floats): every UCB / backup
// operation is an IEEE fp64 +,-,*,/ in the reference's order (compiled with -ffp-contract=off, so
// nothing is fused); log/sqrt of the parent visit count come from a host libm table staged in LDS;
// ties are detected with == on the fp64 scores and broken with the tree's MT19937 stream exactly as
// numpy.random.choice does.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <type_traits>

#include "fc_net_device.h"
#include "kernel_common.h"
#include "np_legacy_rng.h"
#include "tree_device.h"
#include "tree_layout.h"

namespace mz {

constexpr int kThreads = 64;   // one wavefront per workgroup
constexpr int kMaxChunks = 4;  // A <= 256

__device__ __forceinline__ void stage_pbc_table(double* table, const TreeParams& p) {
    for (int i = threadIdx.x; i <= p.S; i += kThreads) {
        table[i] = p.pbc_log[i];
        table[p.S + 1 + i] = p.pbc_sqrt[i];
    }
}

// -------------------------------------------------------------------------------------------------
// select: descend every tree from its root to a leaf (self_play.py:321-335, 364-379) and gather the
// parent hidden state + action of the leaf into the inference batch (self_play.py:339-343).
// -------------------------------------------------------------------------------------------------
// The descent of tree e for simulation `sim` (pbc_table staged; group-uniform control flow).
template <int G, int CH, bool FUSE_GATHER>
__device__ __forceinline__ void select_tree(const TreeParams& p, int sim, const double* pbc_table, int e, int j, int group_base,
                                            int n_root, const MinMax& mm, int32_t mt_pos, float* __restrict__ hidden_out,
                                            int64_t* __restrict__ action_out) {
    if (n_root == 0) {  // inactive tree: keep the batch row defined
        if (j == 0) {
            p.path_len[e] = 0;
            p.leaf_parent[e] = 0;
            if (action_out) action_out[e] = 0;
        }
        if (FUSE_GATHER && hidden_out)
            for (int i = j; i < p.H; i += G) hidden_out[static_cast<size_t>(e) * p.H + i] = 0.f;
        return;
    }

    const GlobalTree tree = global_tree(p, e);
    uint32_t words = 0;
    // (Tried for paired lines: fetching the whole 128-byte line per level and taking hops into its other half from
    // registers -- 294 us against 258 us at E = 2^20: the second half already comes out of L2, the registers cost a
    // wavefront per SIMD.)
    const Descent d = descend<G, CH>(tree, pbc_table, p.S, p.A, sim, n_root, mm, p.discount, p.P == 2,
                                     p.mt_key + static_cast<size_t>(e) * kMtN, mt_pos, words, j, group_base,
                                     p.path_ties ? p.path_ties + e : nullptr, p.E, p.error_flag);
    if (j == 0) {
        p.path_len[e] = d.depth;
        p.leaf_parent[e] = d.parent;
        // Where the node this simulation expands will live: normally line (sim + 1, e); with paired lines the FIRST
        // child a node expands (the node has not been descended through before: one visit, none for the root) takes the
        // free half of that node's own line, if the node owns its line (is not itself such a guest).
        const bool first_child = d.parent_visits == (d.parent == 0 ? 0 : 1);
        const bool paired = p.line_stride != p.block_stride;
        p.leaf_loc[e] = (paired && first_child && (d.parent_loc & 1) == 0) ? (d.parent_loc | 1) : 2 * (sim + 1);
        if (words) {
            p.mt_pos[e] = mt_pos;
            p.tie_words[e] += words;
        }
        if (action_out)
            action_out[e] = (d.depth == 1) ? p.root_action[static_cast<size_t>(e) * p.A + d.slot] : d.slot;
    }
    if (FUSE_GATHER && hidden_out) {
        const float* src = p.hidden + (static_cast<size_t>(d.parent) * p.E + e) * p.H;
        float* dst = hidden_out + static_cast<size_t>(e) * p.H;
        if ((p.H & 3) == 0) {
            const float4* s4 = reinterpret_cast<const float4*>(src);
            float4* d4 = reinterpret_cast<float4*>(dst);
            for (int i = j; i < p.H / 4; i += G) d4[i] = s4[i];
        } else {
            for (int i = j; i < p.H; i += G) dst[i] = src[i];
        }
    }
}

// (8 waves per SIMD: a descent is a chain of dependent loads, the trees in flight are what hides their latency)
template <int G, int CH, bool FUSE_GATHER>
__global__ __launch_bounds__(kThreads, 8) void select_kernel(TreeParams p, int sim, float* __restrict__ hidden_out,
                                                          int64_t* __restrict__ action_out) {
    extern __shared__ double pbc_table[];  // [2][S+1]
    constexpr int kTrees = kThreads / G;
    const int e = blockIdx.x * kTrees + threadIdx.x / G;
    const int j = threadIdx.x % G;
    const int group_base = threadIdx.x - j;  // lane of the group leader inside the wave
    // Per-tree control words first, the table after: at HBM scale every dependent round trip to memory costs
    // microseconds, so the tree's own loads go out before the (cache-resident) table is staged, and the root
    // block is touched now so that the descent's first record is on its way while the table lands in LDS.
    const bool in_range = e < p.E;
    int n_root = 0;
    MinMax mm{};
    int32_t mt_pos = 0;
    if (in_range) {
        n_root = p.root_children[e];
        mm = p.min_max[e];
        if (j == 0) mt_pos = p.mt_pos[e];
        (void)*reinterpret_cast<const volatile int32_t*>(p.blocks + static_cast<size_t>(e) * p.line_stride + (16u * j) % p.block_stride);
    }
    stage_pbc_table(pbc_table, p);
    __syncthreads();
    if (!in_range) return;
    select_tree<G, CH, FUSE_GATHER>(p, sim, pbc_table, e, j, group_base, n_root, mm, mt_pos, hidden_out, action_out);
}

// -------------------------------------------------------------------------------------------------
// select for MANY trees per wavefront (an experiment for HBM scale, off unless mzmcts_set_select_queue asks for it).
// A wavefront of select_kernel lasts as long as the DEEPEST of its 64/G descents while the mean descent is about half
// as long: half of the lane groups idle behind a finished tree.  Here a wavefront owns `trees_per_wave` consecutive trees and its
// lane groups draw them from a wavefront-local queue: one loop iteration = one level of whatever tree each group is
// in, a group whose descent ended takes the next tree of the queue in the same iteration.  Per-tree inputs (min-max
// bounds, root child count, RNG cursor) are staged in LDS up front and the per-tree outputs collected there, so a
// tree change costs no dependent global load; the hidden-state gather runs at the end with all rows in flight.
// Trees are independent, so the order they are descended in changes nothing (tests: bit-identical to select_kernel).
// -------------------------------------------------------------------------------------------------
union QueueSlot {       // per tree of the wavefront: input, then output of the same tree
    MinMax mm;          // in
    int4 out;           // out: {path_len, leaf_parent, leaf_loc, action}
};
constexpr size_t kQueueBytesPerTree = sizeof(QueueSlot) + 2 * sizeof(int32_t);   // + root child count, RNG cursor

template <int G, int CH, bool FUSE_GATHER>
__global__ __launch_bounds__(kThreads, 8) void select_queue_kernel(TreeParams p, int sim, int trees_per_wave,
                                                                   float* __restrict__ hidden_out,
                                                                   int64_t* __restrict__ action_out) {
    extern __shared__ double queue_smem[];  // pb_c table [2][S+1] | QueueSlot [T] | n_root i32 [T] | mt_pos i32 [T]
    constexpr int kTrees = kThreads / G;
    double* pbc_table = queue_smem;
    QueueSlot* slots = reinterpret_cast<QueueSlot*>(queue_smem + 2 * (p.S + 1));
    int32_t* slot_n_root = reinterpret_cast<int32_t*>(slots + trees_per_wave);
    int32_t* slot_mt_pos = slot_n_root + trees_per_wave;
    const int base = blockIdx.x * trees_per_wave;
    const int count = (p.E - base < trees_per_wave) ? p.E - base : trees_per_wave;
    for (int t = threadIdx.x; t < count; t += kThreads) {
        slots[t].mm = p.min_max[base + t];
        slot_n_root[t] = p.root_children[base + t];
        slot_mt_pos[t] = p.mt_pos[base + t];
    }
    stage_pbc_table(pbc_table, p);
    __syncthreads();

    const int j = threadIdx.x % G;
    const int group_base = threadIdx.x - j;
    const int span = child_span<G>(p.A);
    const bool two_player = p.P == 2;
    const bool paired = p.line_stride != p.block_stride;
    int next = kTrees;           // first tree of the queue nobody has taken yet (wavefront-uniform)
    int t = threadIdx.x / G;     // this group's tree, relative to base
    bool active = t < count;
    // the descent this group is in
    int e = base + t, n_children = 0, k = 0, loc = 0, N = sim, depth = 0;
    MinMax mm{};
    int32_t mt_pos = 0;
    uint32_t words = 0;
    if (active) {
        mm = slots[t].mm;
        n_children = slot_n_root[t];
        mt_pos = slot_mt_pos[t];
    }
    while (__ballot(active) != 0ull) {
        bool finished = false;
        if (active) {
            const GlobalTree tree = global_tree(p, e);
            int4 out{0, 0, 0, 0};
            if (n_children == 0) {  // inactive tree: keep the batch row defined
                finished = true;
            } else {
                const LevelPick pick = select_level<G, CH>(tree, pbc_table, p.S, p.A, loc, N, n_children, span, mm, p.discount,
                                                           two_player, p.mt_key + static_cast<size_t>(e) * kMtN, mt_pos, words,
                                                           j, group_base, p.error_flag);
                if (j == 0) {
                    tree.path_store(depth, (loc << 8) | pick.slot);
                    if (p.path_ties) p.path_ties[static_cast<size_t>(depth) * p.E + e] = pick.n_ties;
                }
                ++depth;
                if (pick.child < 0 || depth > sim) {
                    finished = true;
                    if (pick.child >= 0 && j == 0) atomicOr(p.error_flag, 2);  // inconsistent tree (see descend)
                    // where the node this simulation expands will live (see select_tree)
                    const bool first_child = N == (k == 0 ? 0 : 1);
                    out.x = depth;
                    out.y = k;
                    out.z = (paired && first_child && (loc & 1) == 0) ? (loc | 1) : 2 * (sim + 1);
                    out.w = (depth == 1) ? p.root_action[static_cast<size_t>(e) * p.A + pick.slot] : pick.slot;
                    if (j == 0 && words) {
                        p.mt_pos[e] = mt_pos;
                        p.tie_words[e] += words;
                    }
                } else {
                    k = pick.child;
                    loc = pick.loc;
                    N = pick.visits;
                    n_children = p.A;
                }
            }
            if (finished && j == 0) slots[t].out = out;
        }
        // groups whose descent ended take the next trees of the queue, in lane order
        const unsigned long long fin = __ballot(finished && j == 0);
        if (finished) {
            const int rank = __popcll(fin & ((1ull << group_base) - 1ull));
            t = next + rank;
            active = t < count;
            if (active) {
                e = base + t;
                mm = slots[t].mm;
                n_children = slot_n_root[t];
                mt_pos = slot_mt_pos[t];
                words = 0;
                k = 0;
                loc = 0;
                N = sim;
                depth = 0;
            }
        }
        next += __popcll(fin);
    }
    __syncthreads();

    // per-tree results, coalesced
    for (int r = threadIdx.x; r < count; r += kThreads) {
        const int4 out = slots[r].out;
        p.path_len[base + r] = out.x;
        p.leaf_parent[base + r] = out.y;
        p.leaf_loc[base + r] = out.z;
        if (action_out) action_out[base + r] = out.w;
    }
    // hidden-state gather (self_play.py:339-343): every row's loads in flight together
    if (FUSE_GATHER && hidden_out) {
        constexpr int kBatch = 4;
        if ((p.H & 3) == 0) {
            const int h4 = p.H / 4;
            const int pieces = count * h4;
            for (int i0 = threadIdx.x; i0 < pieces; i0 += kThreads * kBatch) {
                float4 v[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const int i = i0 + u * kThreads;
                    v[u] = float4{0.f, 0.f, 0.f, 0.f};
                    if (i < pieces) {
                        const int r = i / h4, c = i - r * h4;
                        if (slots[r].out.x > 0)
                            v[u] = reinterpret_cast<const float4*>(
                                p.hidden + (static_cast<size_t>(slots[r].out.y) * p.E + base + r) * p.H)[c];
                    }
                }
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const int i = i0 + u * kThreads;
                    if (i < pieces) reinterpret_cast<float4*>(hidden_out + static_cast<size_t>(base) * p.H)[i] = v[u];
                }
            }
        } else {
            const int pieces = count * p.H;
            for (int i0 = threadIdx.x; i0 < pieces; i0 += kThreads * kBatch) {
                float v[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const int i = i0 + u * kThreads;
                    v[u] = 0.f;
                    if (i < pieces) {
                        const int r = i / p.H, c = i - r * p.H;
                        if (slots[r].out.x > 0) v[u] = p.hidden[(static_cast<size_t>(slots[r].out.y) * p.E + base + r) * p.H + c];
                    }
                }
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const int i = i0 + u * kThreads;
                    if (i < pieces) hidden_out[static_cast<size_t>(base) * p.H + i] = v[u];
                }
            }
        }
    }
}

// Stand-alone gather for large hidden states (ResNet planes): one workgroup row per tree,
// 16-byte lanes, fully coalesced on both sides.
__global__ __launch_bounds__(256) void gather_hidden_kernel(TreeParams p, float* __restrict__ hidden_out) {
    const int e = blockIdx.x;
    const int k = p.leaf_parent[e];
    const float* src = p.hidden + (static_cast<size_t>(k) * p.E + e) * p.H;
    float* dst = hidden_out + static_cast<size_t>(e) * p.H;
    if ((p.H & 3) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < p.H / 4; i += gridDim.y * blockDim.x) d4[i] = s4[i];
    } else {
        for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < p.H; i += gridDim.y * blockDim.x) dst[i] = src[i];
    }
}

// The same gather laid out as the residual networks' dynamics input (reference models.py:553-568): row e =
// [parent hidden state (H floats = channels x plane) | one plane filled with action / action_space_size].
__global__ __launch_bounds__(256) void gather_dynamics_input_kernel(TreeParams p, const int64_t* __restrict__ action,
                                                                    float* __restrict__ out, int plane,
                                                                    float action_space) {
    const int e = blockIdx.x;
    const int k = p.leaf_parent[e];
    const float* src = p.hidden + (static_cast<size_t>(k) * p.E + e) * p.H;
    float* dst = out + static_cast<size_t>(e) * (p.H + plane);
    const int row = p.H + plane;
    const float fill = static_cast<float>(action[e]) / action_space;
    if (((p.H | row) & 1) == 0) {   // rows of an even number of floats: 8-byte accesses (the output rows are 8-byte aligned)
        const float2* s2 = reinterpret_cast<const float2*>(src);
        float2* d2 = reinterpret_cast<float2*>(dst);
        const int h2 = p.H / 2;
        for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < row / 2; i += gridDim.y * blockDim.x)
            d2[i] = i < h2 ? s2[i] : float2{fill, fill};
        return;
    }
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < row; i += gridDim.y * blockDim.x) dst[i] = i < p.H ? src[i] : fill;
}

// The same for SHORT rows (3x3 boards: 153 floats): a 256-thread block per env leaves most lanes idle and makes E tiny
// workgroups; here a block takes `envs_per_block` consecutive envs and walks their rows as one flat, fully coalesced
// index range (e = t / row by a multiply: exact for t < envs_per_block * row <= 4096).
__global__ __launch_bounds__(256) void gather_dynamics_rows_kernel(TreeParams p, const int64_t* __restrict__ action,
                                                                   float* __restrict__ out, int plane, float action_space,
                                                                   int envs_per_block, uint32_t row_magic) {
    const int row = p.H + plane;
    const int e0 = blockIdx.x * envs_per_block;
    const int envs = (p.E - e0 < envs_per_block) ? p.E - e0 : envs_per_block;
    float* dst = out + static_cast<size_t>(e0) * row;
    for (int t = threadIdx.x; t < envs * row; t += 256) {
        const int le = static_cast<int>((static_cast<uint32_t>(t) * row_magic) >> 20);
        const int i = t - le * row;
        const int e = e0 + le;
        dst[t] = i < p.H ? p.hidden[(static_cast<size_t>(p.leaf_parent[e]) * p.E + e) * p.H + i]
                         : static_cast<float>(action[e]) / action_space;
    }
}

// Contiguous slab copy (network output -> pool slab) when the caller could not write in place.
__global__ __launch_bounds__(256) void copy_slab_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (; i < n / 4; i += stride) d4[i] = s4[i];
    } else {
        for (; i < n; i += stride) dst[i] = src[i];
    }
}

// -------------------------------------------------------------------------------------------------
// expand_roots: root.expand over the legal actions + exploration noise (self_play.py:293-315,
// 452-477) and reset of the per-search state.
// -------------------------------------------------------------------------------------------------
template <int G, int CH, bool INJECTED>
__global__ __launch_bounds__(kThreads) void expand_roots_kernel(TreeParams p, const float* __restrict__ value_logits,
                                                                const float* __restrict__ reward_logits,
                                                                const float* __restrict__ policy_logits,
                                                                const float* __restrict__ root_hidden,
                                                                const double* __restrict__ inj_reward,
                                                                const double* __restrict__ inj_priors,
                                                                const double* __restrict__ noise,  // [E][A] or null
                                                                const uint32_t* __restrict__ rng_skip) {
    constexpr int kTrees = kThreads / G;
    const int e = blockIdx.x * kTrees + threadIdx.x / G;
    const int j = threadIdx.x % G;
    if (e >= p.E) return;
    const int n_children = p.root_children[e];
    if (j == 0) reset_search_state(p, e, rng_skip);
    if (n_children == 0) return;

    double reward = 0.0;
    float predicted = 0.f;
    if (INJECTED) {
        reward = inj_reward[e];
    } else {
        predicted = support_to_scalar_group<G>(value_logits + static_cast<size_t>(e) * p.F, p.F, p.support, j);
        if (reward_logits)
            reward = static_cast<double>(
                support_to_scalar_group<G>(reward_logits + static_cast<size_t>(e) * p.F, p.F, p.support, j));
    }
    if (j == 0) {
        p.root_reward[e] = reward;
        p.root_predicted[e] = predicted;
    }

    float logit[CH];
    bool valid[CH];
    double prior[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int child = c * G + j;
        valid[c] = child < n_children;
        logit[c] = 0.f;
        prior[c] = 0.0;
        if (valid[c]) {
            if (INJECTED)
                prior[c] = inj_priors[static_cast<size_t>(e) * p.A + child];
            else
                logit[c] = policy_logits[static_cast<size_t>(e) * p.A + p.root_action[static_cast<size_t>(e) * p.A + child]];
        }
    }
    if (!INJECTED) group_softmax<G, CH>(logit, valid, prior);
    write_root_children<G, CH>(global_tree(p, e), p.A, n_children, prior,
                               noise ? noise + static_cast<size_t>(e) * p.A : nullptr, p.noise_frac, j);
    if (root_hidden) {
        const float* src = root_hidden + static_cast<size_t>(e) * p.H;
        float* dst = p.hidden + static_cast<size_t>(e) * p.H;  // slab 0
        for (int i = j; i < p.H; i += G) dst[i] = src[i];
    }
}

// -------------------------------------------------------------------------------------------------
// expand_backup: decode value/reward, expand the leaf over the full action space, back the value
// up the search path with min-max statistics (self_play.py:344-356, 407-431, 452-466, 560-562).
// -------------------------------------------------------------------------------------------------
// expand + backup of tree e for simulation `sim`; returns (in every lane of the group) the min-max statistics it leaves.
template <int G, int CH, bool INJECTED>
__device__ __forceinline__ MinMax expand_backup_tree(const TreeParams& p, int sim, int e, int j, StagedNode* staged,
                                                     const float* __restrict__ value_logits,
                                                     const float* __restrict__ reward_logits,
                                                     const float* __restrict__ policy_logits,
                                                     const double* __restrict__ inj_value,
                                                     const double* __restrict__ inj_reward,
                                                     const double* __restrict__ inj_priors) {
    const int depth = p.path_len[e];  // >= 1

    double value;
    float reward_f;
    if (INJECTED) {
        value = inj_value[e];
        reward_f = static_cast<float>(inj_reward[e]);
    } else {
        float value_f;
        support_to_scalar_pair<G>(value_logits + static_cast<size_t>(e) * p.F, reward_logits + static_cast<size_t>(e) * p.F,
                                  p.F, p.support, j, value_f, reward_f);
        value = static_cast<double>(value_f);
    }
    float logit[CH];
    bool valid[CH];
    double prior[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int child = c * G + j;
        valid[c] = child < p.A;
        logit[c] = 0.f;
        prior[c] = 0.0;
        if (valid[c]) {
            if (INJECTED)
                prior[c] = inj_priors[static_cast<size_t>(e) * p.A + child];
            else
                logit[c] = policy_logits[static_cast<size_t>(e) * p.A + child];
        }
    }
    if (!INJECTED) group_softmax<G, CH>(logit, valid, prior);

    const GlobalTree tree = global_tree(p, e);
    const int loc_new = p.leaf_loc[e];                   // chosen by select (own line of slab sim + 1, or the parent's)
    write_children<G, CH>(tree, loc_new, p.A, prior, j);

    MinMax mm = p.min_max[e];
    double root_value_sum = (j == 0) ? p.root_value_sum[e] : 0.0;
    const double root_reward = (j == 0) ? p.root_reward[e] : 0.0;
    backup<G>(tree, depth, sim, value, reward_f, p.P == 2, p.discount, mm, root_value_sum, root_reward, staged, j, loc_new);
    if (j == 0) {
        p.root_value_sum[e] = root_value_sum;
        p.min_max[e] = mm;
        if (depth > p.max_depth[e]) p.max_depth[e] = depth;
        p.depth_sum[e] += depth;
    }
    mm.minimum = __shfl(mm.minimum, 0, G);               // the leader owns the running statistics
    mm.maximum = __shfl(mm.maximum, 0, G);
    return mm;
}

template <int G, int CH, bool INJECTED>
__global__ __launch_bounds__(kThreads) void expand_backup_kernel(TreeParams p, int sim,
                                                                 const float* __restrict__ value_logits,
                                                                 const float* __restrict__ reward_logits,
                                                                 const float* __restrict__ policy_logits,
                                                                 const double* __restrict__ inj_value,
                                                                 const double* __restrict__ inj_reward,
                                                                 const double* __restrict__ inj_priors) {
    constexpr int kTrees = kThreads / G;
    __shared__ StagedNode staged[kTrees][kStageLevels];
    const int tree_in_block = threadIdx.x / G;
    const int e = blockIdx.x * kTrees + tree_in_block;
    const int j = threadIdx.x % G;
    if (e >= p.E) return;
    if (p.root_children[e] == 0) return;
    (void)expand_backup_tree<G, CH, INJECTED>(p, sim, e, j, staged[tree_in_block], value_logits, reward_logits, policy_logits,
                                              inj_value, inj_reward, inj_priors);
}

// -------------------------------------------------------------------------------------------------
// One step of the lock-step loop in ONE launch: expand + backup of simulation `sim`, then the descent of simulation
// sim + 1 for the same trees.  63 % of the nodes a descent visits lie on the path the backup has just rewritten (CartPole
// traces): in one launch they are served by the XCD's L2 instead of a second trip to HBM, and a launch boundary goes.
// -------------------------------------------------------------------------------------------------
template <int G, int CH, bool INJECTED, bool FUSE_GATHER>
__global__ __launch_bounds__(kThreads) void expand_backup_select_kernel(TreeParams p, int sim,
                                                                        const float* __restrict__ value_logits,
                                                                        const float* __restrict__ reward_logits,
                                                                        const float* __restrict__ policy_logits,
                                                                        const double* __restrict__ inj_value,
                                                                        const double* __restrict__ inj_reward,
                                                                        const double* __restrict__ inj_priors,
                                                                        float* __restrict__ hidden_out,
                                                                        int64_t* __restrict__ action_out) {
    extern __shared__ double pbc_table[];  // [2][S+1]
    constexpr int kTrees = kThreads / G;
    __shared__ StagedNode staged[kTrees][kStageLevels];
    const int tree_in_block = threadIdx.x / G;
    const int e = blockIdx.x * kTrees + tree_in_block;
    const int j = threadIdx.x % G;
    const int group_base = threadIdx.x - j;
    stage_pbc_table(pbc_table, p);
    __syncthreads();
    if (e >= p.E) return;
    const int n_root = p.root_children[e];
    MinMax mm{};
    if (n_root != 0)
        mm = expand_backup_tree<G, CH, INJECTED>(p, sim, e, j, staged[tree_in_block], value_logits, reward_logits,
                                                 policy_logits, inj_value, inj_reward, inj_priors);
    // The descent reads records that lanes of THIS wavefront have just stored (a tree belongs to one lane group): the
    // wavefront's vector-memory operations reach its CU's L1 in program order, so ordering them inside the wavefront is
    // enough -- no other CU touches these records during the launch, and lines cached by earlier launches were dropped
    // at the launch boundary.
    group_memory_fence();
    const int32_t mt_pos = (j == 0) ? p.mt_pos[e] : 0;
    select_tree<G, CH, FUSE_GATHER>(p, sim + 1, pbc_table, e, j, group_base, n_root, mm, mt_pos, hidden_out, action_out);
}

// add_exploration_noise's draw (self_play.py:468-477: numpy.random.dirichlet([alpha] * len(actions))) on every
// tree's OWN stream, on the device: the legacy gamma sampler over glibc's log / pow (np_legacy_rng.h DeviceStream,
// glibc_libm.h), one thread per tree.  First steps over the words the host mirror consumed since the device copy last
// moved (rng_skip, zeroed here so that the search's own reset does not skip them again).  The rows go where a host
// draw would have been uploaded; expand_roots / the whole-move kernels read them from there.
constexpr int kNoiseWindow = 96;
__global__ __launch_bounds__(kThreads) void root_noise_kernel(TreeParams p, uint32_t* __restrict__ rng_skip) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.E) return;
    const int n = p.root_children[e];
    double* row = p.noise_rows + static_cast<size_t>(e) * p.A;
    for (int i = 0; i < p.A; ++i) row[i] = 0.0;
    p.noise_words[e] = 0u;
    if (n == 0) return;  // inactive: the stream is left alone
    uint32_t* key = p.mt_key + static_cast<size_t>(e) * kMtN;
    int32_t pos = p.mt_pos[e];
    const uint32_t skip = rng_skip ? rng_skip[e] : 0u;
    for (uint32_t i = 0; i < skip; ++i) (void)mt_next(key, &pos);
    if (rng_skip) rng_skip[e] = 0u;
    // the words the draw is likely to consume, staged in LDS with every load in flight (a gamma variate takes >= 4
    // words; read one by one out of global memory each is a dependent round trip)
    __shared__ uint32_t window[kThreads][kNoiseWindow + 1];
    const int ahead = (kMtN - pos < kNoiseWindow) ? kMtN - pos : kNoiseWindow;
#pragma unroll 8
    for (int i = 0; i < ahead; ++i) window[threadIdx.x][i] = key[pos + i];
    DeviceStream stream{key, pos, 0u, window[threadIdx.x], pos, pos + (ahead > 0 ? ahead : 0)};
    stream.dirichlet(p.noise_alpha, n, row);
    p.mt_pos[e] = stream.pos;
    p.noise_words[e] = stream.words;
}

// One move of a batch whose inputs live on the device (mzmcts_moves_prepare_device): records what the coming search is
// run with -- legal set, player to move: the caller's environment kernels will overwrite them for the next move -- and, for
// the envs that will be searched (active, not stalled, inside their move limit: the conditions of move_stalled), steps
// over the host's pending words and draws the exploration noise as root_noise_kernel does.
__global__ __launch_bounds__(kThreads) void move_inputs_kernel(TreeParams p, const uint32_t* __restrict__ rng_skip,
                                                               const uint8_t* __restrict__ stall,
                                                               const int32_t* __restrict__ move_limit, int move_index,
                                                               int draw_noise, int32_t* __restrict__ nlegal_out,
                                                               int32_t* __restrict__ to_play_out,
                                                               uint32_t* __restrict__ words_out,
                                                               int32_t* __restrict__ legal_out, MoveInputsExtra extra) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.E) return;
    const int n = p.root_children[e];
    nlegal_out[e] = n;
    to_play_out[e] = p.root_to_play[e];
    for (int i = 0; i < p.A; ++i) legal_out[static_cast<size_t>(e) * p.A + i] = p.root_action[static_cast<size_t>(e) * p.A + i];
    if (extra.own_nlegal) {   // lock-step moves: the engine's own root inputs (what its captured simulation loop reads)
        extra.own_nlegal[e] = n;
        extra.own_to_play[e] = p.root_to_play[e];
        for (int i = 0; i < p.A; ++i) extra.own_legal[static_cast<size_t>(e) * p.A + i] = p.root_action[static_cast<size_t>(e) * p.A + i];
    }
    // an env whose game ended with the move before starts a new one: its move counter restarts (MoveCtl::game_moves)
    if (extra.game_moves && extra.finished && extra.finished[e]) extra.game_moves[e] = 0;
    words_out[e] = 0u;
    double* row = p.noise_rows + static_cast<size_t>(e) * p.A;
    for (int i = 0; i < p.A; ++i) row[i] = 0.0;
    if (n == 0 || (stall && stall[e]) || (move_limit && move_index >= move_limit[e])) return;
    uint32_t* key = p.mt_key + static_cast<size_t>(e) * kMtN;
    int32_t pos = p.mt_pos[e];
    const uint32_t skip = rng_skip ? rng_skip[e] : 0u;
    for (uint32_t i = 0; i < skip; ++i) (void)mt_next(key, &pos);
    if (draw_noise) {
        __shared__ uint32_t window[kThreads][kNoiseWindow + 1];
        const int ahead = (kMtN - pos < kNoiseWindow) ? kMtN - pos : kNoiseWindow;
#pragma unroll 8
        for (int i = 0; i < ahead; ++i) window[threadIdx.x][i] = key[pos + i];
        DeviceStream stream{key, pos, 0u, window[threadIdx.x], pos, pos + (ahead > 0 ? ahead : 0)};
        stream.dirichlet(p.noise_alpha, n, row);
        pos = stream.pos;
        words_out[e] = stream.words;
    }
    p.mt_pos[e] = pos;
}

// numpy.random.seed(seeds[e]) for every stream, on the device copy.
__global__ __launch_bounds__(256) void seed_streams_kernel(uint32_t* __restrict__ keys, int32_t* __restrict__ pos,
                                                           const uint32_t* __restrict__ seeds, int E) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int32_t ps;
    mt_seed(keys + static_cast<size_t>(e) * kMtN, &ps, seeds[e]);
    pos[e] = ps;
}

// -------------------------------------------------------------------------------------------------
// Fully-connected network, lock-step form: the same device functions the fused kernel uses, one
// launch per inference over the [E, .] batch (so the two shapes can be compared bit for bit).
// -------------------------------------------------------------------------------------------------
template <int G, bool INITIAL>
__global__ __launch_bounds__(kThreads) void fc_inference_kernel(FcNet net, const float* __restrict__ weights, int E,
                                                                const float* __restrict__ in,        // obs or hidden
                                                                const int64_t* __restrict__ action,  // recurrent only
                                                                float* __restrict__ value_logits,
                                                                float* __restrict__ reward_logits,
                                                                float* __restrict__ policy_logits,
                                                                float* __restrict__ hidden_out) {
    extern __shared__ __attribute__((aligned(16))) float fc_smem[];  // [padded weights][64/G][scratch_floats]
    constexpr int kTrees = kThreads / G;
    float* w_lds = fc_smem;
    NeuronDesc* table = reinterpret_cast<NeuronDesc*>(fc_smem + ((net.n_weights_lds + 3) & ~3));
    stage_fc_weights(net, weights, w_lds, threadIdx.x, kThreads);
    __syncthreads();
    build_fc_tables(net, w_lds, table, G, threadIdx.x, kThreads);
    __syncthreads();
    const int init_entries = phase_list_entries(net.init_pre, net.n_init_pre, G) +
                             phase_list_entries(net.init_post, net.n_init_post, G);
    const int tree_in_block = threadIdx.x / G;
    const int e = blockIdx.x * kTrees + tree_in_block;
    const int j = threadIdx.x % G;
    if (e >= E) return;
    float* scratch = reinterpret_cast<float*>(table + fc_table_entries(net, G)) +
                     static_cast<size_t>(tree_in_block) * net.scratch_floats;
    fc_clear_scratch<G>(net, scratch, j);
    if (INITIAL) {
        fc_initial<G>(net, table, w_lds, scratch, in + static_cast<size_t>(e) * net.obs, j);
        for (int i = j; i < net.F; i += G)
            reward_logits[static_cast<size_t>(e) * net.F + i] = (i == net.F / 2) ? 0.f : -INFINITY;
    } else {
        fc_recurrent<G>(net, table + init_entries, w_lds, scratch, in + static_cast<size_t>(e) * net.enc,
                        static_cast<int>(action[e]), j);
        for (int i = j; i < net.F; i += G) reward_logits[static_cast<size_t>(e) * net.F + i] = scratch[net.off_reward + i];
    }
    for (int i = j; i < net.F; i += G) value_logits[static_cast<size_t>(e) * net.F + i] = scratch[net.off_value + i];
    for (int i = j; i < net.A; i += G) policy_logits[static_cast<size_t>(e) * net.A + i] = scratch[net.off_policy + i];
    for (int i = j; i < net.enc; i += G) hidden_out[static_cast<size_t>(e) * net.enc + i] = scratch[net.off_norm + i];
}

// -------------------------------------------------------------------------------------------------
// Fused whole-move search for fully-connected networks (MCTS.run, self_play.py:261-362, for every
// tree of the workgroup, in ONE launch).
//
// LDS per workgroup:  pb_c tables | network weights | per tree { child blocks (S+1) x stride,
//                     path S x 4 B, activation scratch, optionally hidden states (S+1) x enc x 4 B }
// Nothing is exchanged between trees, so there is no barrier after the initial staging.  Hidden
// states are also streamed to the HBM pool (fire-and-forget) and, at the end, the child blocks are
// copied out slab by slab, so readout / export see exactly what the lock-step kernels leave.
// -------------------------------------------------------------------------------------------------
template <int G, int CH>
__global__ __launch_bounds__(kThreads) void search_fused_fc_kernel(TreeParams p, FcNet net, FusedLayout lay,
                                                                   const float* __restrict__ weights,
                                                                   const float* __restrict__ observations,  // [E][obs]
                                                                   MoveCtl ctl, int n_sims) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    MZ_STAMP_DECL
    double* pbc_table = reinterpret_cast<double*>(smem);
    float* w_lds = reinterpret_cast<float*>(smem + lay.off_weights);
    const NeuronDesc* fc_table = reinterpret_cast<const NeuronDesc*>(smem + lay.off_table);
    const NeuronDesc* rec_table = reinterpret_cast<const NeuronDesc*>(smem + lay.off_rec_table);
    stage_pbc_table(pbc_table, p);
    stage_fc_weights(net, weights, w_lds, threadIdx.x, kThreads);
    __syncthreads();
    build_fc_tables(net, w_lds, reinterpret_cast<NeuronDesc*>(smem + lay.off_table), G, threadIdx.x, kThreads);
    __syncthreads();
    MZ_STAMP(0);

    constexpr int kTrees = kThreads / G;
    const int tree_in_block = threadIdx.x / G;
    const int e = blockIdx.x * kTrees + tree_in_block;
    const int j = threadIdx.x % G;
    const int group_base = threadIdx.x - j;
    if (e >= p.E) return;
    const int n_root = p.root_children[e];
    if (move_stalled(p, ctl, e, j)) return;
    const double* noise = ctl.noise;
    if (j == 0) reset_search_state(p, e, ctl.rng_skip);
    if (n_root == 0) {
        if (j == 0 && ctl.actions) ctl.actions[e] = -1;
        return;
    }

    uint8_t* region = smem + lay.off_trees + static_cast<size_t>(tree_in_block) * lay.tree_bytes;
    const LdsTree tree{region, p.block_stride, p.links_offset, reinterpret_cast<int32_t*>(region + lay.off_path)};
    float* scratch = reinterpret_cast<float*>(region + lay.off_scratch);
    const bool hidden_in_lds = lay.off_hidden != 0xffffffffu;
    float* hidden_lds = hidden_in_lds ? reinterpret_cast<float*>(region + lay.off_hidden) : nullptr;
    const int H = net.enc;
    const bool two_player = p.P == 2;

    // ---- root: initial inference, root.expand over the legal actions, exploration noise ----------
    fc_clear_scratch<G>(net, scratch, j);
    fc_initial<G>(net, fc_table, w_lds, scratch, observations + static_cast<size_t>(e) * net.obs, j);
    const float predicted = support_to_scalar_group<G>(scratch + net.off_value, net.F, net.support, j);
    {
        float logit[CH];
        bool valid[CH];
        double prior[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int child = c * G + j;
            valid[c] = child < n_root;
            prior[c] = 0.0;
            logit[c] = valid[c] ? scratch[net.off_policy + p.root_action[static_cast<size_t>(e) * p.A + child]] : 0.f;
        }
        group_softmax<G, CH>(logit, valid, prior);
        write_root_children<G, CH>(tree, p.A, n_root, prior, noise ? noise + static_cast<size_t>(e) * p.A : nullptr,
                                   p.noise_frac, j);
    }
    for (int i = j; i < H; i += G) {
        const float h = scratch[net.off_norm + i];
        if (hidden_in_lds) hidden_lds[i] = h;
        p.hidden[static_cast<size_t>(e) * H + i] = h;  // slab 0
    }
    group_memory_fence();

    MinMax mm{INFINITY, -INFINITY};
    double root_value_sum = 0.0;
    const double root_reward = 0.0;  // log(one_hot(centre)) decodes to exactly 0
    int32_t mt_pos = (j == 0) ? p.mt_pos[e] : 0;
    uint32_t words = 0;
    int max_depth = 0;
    int64_t depth_sum = 0;
    uint32_t* mt_key = p.mt_key + static_cast<size_t>(e) * kMtN;
    MZ_STAMP(1);

    // ---- S simulations, entirely inside the CU -----------------------------------------------------
    for (int sim = 0; sim < n_sims; ++sim) {
        const Descent d = descend<G, CH>(tree, pbc_table, p.S, p.A, sim, n_root, mm, p.discount, two_player, mt_key,
                                         mt_pos, words, j, group_base, nullptr, 0, p.error_flag);
        MZ_STAMP(2);
        const int action = (d.depth == 1) ? p.root_action[static_cast<size_t>(e) * p.A + d.slot] : d.slot;
        const float* parent_hidden = hidden_in_lds
                                         ? hidden_lds + static_cast<size_t>(d.parent) * H
                                         : p.hidden + (static_cast<size_t>(d.parent) * p.E + e) * H;
#ifdef MZ_STAMPS
        {
            float* x = scratch;
            for (int i = j; i < net.enc; i += G) x[i] = parent_hidden[i];
            for (int a = j; net.enc + a < ((net.enc + net.A + 3) & ~3); a += G) x[net.enc + a] = (a == action) ? 1.f : 0.f;
            group_memory_fence();
            MZ_STAMP(8);
            const NeuronDesc* tb = rec_table;
            for (int ph = 0; ph < net.n_rec_pre; ++ph) {
                run_phase<G>(net.rec_pre[ph], tb, w_lds, scratch, j);
                tb += phase_passes(net.rec_pre[ph], G) * 4 * G;
                MZ_STAMP(9 + ph);
            }
            unit_rescale<G>(scratch + net.off_raw, scratch + net.off_norm, net.enc, j);
            MZ_STAMP(12);
            for (int ph = 0; ph < net.n_rec_post; ++ph) {
                run_phase<G>(net.rec_post[ph], tb, w_lds, scratch, j);
                tb += phase_passes(net.rec_post[ph], G) * 4 * G;
                MZ_STAMP(13 + ph);
            }
        }
#else
        fc_recurrent<G>(net, rec_table, w_lds, scratch, parent_hidden, action, j);
#endif
        MZ_STAMP(3);
        float value_f, reward_f;
        support_to_scalar_pair<G>(scratch + net.off_value, scratch + net.off_reward, net.F, net.support, j, value_f,
                                  reward_f);
        const double value = static_cast<double>(value_f);
        float logit[CH];
        bool valid[CH];
        double prior[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int child = c * G + j;
            valid[c] = child < p.A;
            prior[c] = 0.0;
            logit[c] = valid[c] ? scratch[net.off_policy + child] : 0.f;
        }
        group_softmax<G, CH>(logit, valid, prior);
        MZ_STAMP(4);
        const int k_new = sim + 1;
        write_children<G, CH>(tree, 2 * k_new, p.A, prior, j);   // (a tree in LDS keeps block k at slab k)
        for (int i = j; i < H; i += G) {
            const float h = scratch[net.off_norm + i];
            if (hidden_in_lds) hidden_lds[static_cast<size_t>(k_new) * H + i] = h;
            p.hidden[(static_cast<size_t>(k_new) * p.E + e) * H + i] = h;
        }
        group_memory_fence();
        MZ_STAMP(5);
        // the activation scratch is dead once the heads are decoded: it doubles as the backup's hand-over area
        backup<G>(tree, d.depth, sim, value, reward_f, two_player, p.discount, mm, root_value_sum, root_reward,
                  reinterpret_cast<StagedNode*>(scratch), j, 2 * k_new);
        group_memory_fence();
        // the leader owns the running min-max statistics; every lane scores its child with them
        mm.minimum = __shfl(mm.minimum, 0, G);
        mm.maximum = __shfl(mm.maximum, 0, G);
        if (d.depth > max_depth) max_depth = d.depth;
        depth_sum += d.depth;
        MZ_STAMP(6);
    }

    // ---- publish: per-tree statistics and the child blocks (what readout / export_tree read) -------
    if (j == 0) {
        p.root_reward[e] = root_reward;
        p.root_predicted[e] = predicted;
        p.root_value_sum[e] = root_value_sum;
        p.min_max[e] = mm;
        p.max_depth[e] = max_depth;
        p.depth_sum[e] = depth_sum;
        uint32_t sample_words = 0;
        if (ctl.temperature) {  // SelfPlay.select_action on the tree's own stream (kernel_common.h)
            const ChildLinks* root_links = tree.links(0);
            const int slot = device_select_action([&](int i) { return root_links[i].visits; }, n_root, move_temperature(ctl, e),
                                                  mt_key, &mt_pos, &sample_words);
            if (ctl.actions) ctl.actions[e] = slot >= 0 ? p.root_action[static_cast<size_t>(e) * p.A + slot] : slot;
            if (ctl.game_moves) ctl.game_moves[e] += 1;        // this env's game is one move longer
        }
        if (words | sample_words) p.mt_pos[e] = mt_pos;
        if (words) p.tie_words[e] = words;
        if (ctl.tie_words) ctl.tie_words[e] = words;
        if (ctl.sample_words) ctl.sample_words[e] = sample_words;
        if (ctl.root_value_sum) ctl.root_value_sum[e] = root_value_sum;
        if (ctl.root_predicted) ctl.root_predicted[e] = predicted;
        if (ctl.max_depth) ctl.max_depth[e] = max_depth;
        if (ctl.depth_sum) ctl.depth_sum[e] = static_cast<int32_t>(depth_sum);
    }
    if (ctl.visits)
        for (int c = j; c < p.A; c += G)
            ctl.visits[static_cast<size_t>(e) * p.A + c] = (c < n_root) ? tree.links(0)[c].visits : 0;
    // (in LDS a block is two member arrays, in HBM an array of 32-byte child records: word i of the one is stats[i] or
    // links[i - A], word 2 c / 2 c + 1 of the other the stats / links of child c)
    for (int k = 0; k <= n_sims; ++k) {
        const uint4* src = reinterpret_cast<const uint4*>(region + static_cast<size_t>(k) * p.block_stride);
        uint4* dst = reinterpret_cast<uint4*>(p.blocks + (static_cast<size_t>(k) * p.E + e) * p.line_stride);   // own line, half 0
        for (int i = j; i < 2 * p.A; i += G) dst[i < p.A ? 2 * i : 2 * (i - p.A) + 1] = src[i];
    }
    MZ_STAMP(7);
    MZ_STAMP_FLUSH;
}

#ifdef MZ_STAMPS
hipError_t read_stamp_sums(unsigned long long* out, bool reset) {
    hipError_t err = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp_sums), sizeof(unsigned long long) * 16);
    if (err == hipSuccess && reset) {
        unsigned long long zeros[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        err = hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_sums), zeros, sizeof(zeros));
    }
    return err;
}
#endif

// -------------------------------------------------------------------------------------------------
// launchers
// -------------------------------------------------------------------------------------------------
int default_group_width(int A) {
    int g = 1;
    while (g < A && g < 64) g <<= 1;
    return g;
}

template <int V>
using IntC = std::integral_constant<int, V>;

// Calls fn(IntC<G>{}, IntC<CH>{}) for the tree's lane-group width / chunk count.
template <typename Fn>
static void dispatch_group(const TreeParams& p, Fn&& fn) {
    if (p.A > 64) {
        fn(IntC<64>{}, IntC<kMaxChunks>{});
        return;
    }
    switch (p.group) {
        case 1:
            if (p.chunks == 2) fn(IntC<1>{}, IntC<2>{}); else fn(IntC<1>{}, IntC<1>{});
            break;
        case 2: fn(IntC<2>{}, IntC<1>{}); break;
        case 4:
            if (p.chunks == 2) fn(IntC<4>{}, IntC<2>{}); else fn(IntC<4>{}, IntC<1>{});
            break;
        case 8:
            if (p.chunks == 2) fn(IntC<8>{}, IntC<2>{}); else fn(IntC<8>{}, IntC<1>{});
            break;
        case 16: fn(IntC<16>{}, IntC<1>{}); break;
        case 32: fn(IntC<32>{}, IntC<1>{}); break;
        default: fn(IntC<64>{}, IntC<1>{}); break;
    }
}

static inline int tree_grid(const TreeParams& p) {
    const int trees = kThreads / p.group;
    return (p.E + trees - 1) / trees;
}

constexpr int kFuseGatherMaxFloats = 64;

// Trees per wavefront of the select kernel (mzmcts_set_select_queue): 64/G, one descent per lane group, unless the
// caller asks for a wavefront-local queue (select_queue_kernel).  Measured at E = 2^20 (profiles/r02_select_queue.jsonl):
// 258 us without a queue, 297 / 309 / 271 us with 64 / 128 / 256 trees per wavefront -- the kernel runs at the memory
// system's request rate, not at the length of its longest descent, so the queue stays off unless requested.
static int select_queue_trees(const TreeParams& p, int requested) {
    const int per_wave = kThreads / p.group;
    if (p.A > 64 || requested <= per_wave) return per_wave;
    return requested;
}

hipError_t launch_select(const TreeParams& p, int sim, float* hidden_out, int64_t* action_out, int queue_request,
                         hipStream_t stream, const LaunchTiming* timing) {
    const size_t lds = sizeof(double) * 2 * (static_cast<size_t>(p.S) + 1);
    const int grid = tree_grid(p);
    const bool fuse = p.H <= kFuseGatherMaxFloats;
    const int queue_trees = select_queue_trees(p, queue_request);
    dispatch_group(p, [&](auto g, auto ch) {
        constexpr int G = decltype(g)::value;
        constexpr int CH = decltype(ch)::value;
        if (queue_trees > kThreads / G) {
            const size_t qlds = lds + kQueueBytesPerTree * static_cast<size_t>(queue_trees);   // (lds = 16 (S + 1) bytes)
            const int qgrid = (p.E + queue_trees - 1) / queue_trees;
            if (fuse)
                launch_kernel(select_queue_kernel<G, CH, true>, dim3(qgrid), dim3(kThreads), qlds, stream, timing, p, sim,
                              queue_trees, hidden_out, action_out);
            else
                launch_kernel(select_queue_kernel<G, CH, false>, dim3(qgrid), dim3(kThreads), qlds, stream, timing, p, sim,
                              queue_trees, hidden_out, action_out);
        } else if (fuse)
            launch_kernel(select_kernel<G, CH, true>, dim3(grid), dim3(kThreads), lds, stream, timing, p, sim, hidden_out,
                          action_out);
        else
            launch_kernel(select_kernel<G, CH, false>, dim3(grid), dim3(kThreads), lds, stream, timing, p, sim, hidden_out,
                          action_out);
    });
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    if (!fuse && hidden_out) {
        const int per_row = (p.H + 3) / 4;
        int gy = (per_row + 255) / 256;
        if (gy > 8) gy = 8;
        gather_hidden_kernel<<<dim3(p.E, gy), dim3(256), 0, stream>>>(p, hidden_out);
        err = hipGetLastError();
    }
    return err;
}

hipError_t launch_root_noise(const TreeParams& p, uint32_t* rng_skip, hipStream_t stream) {
    root_noise_kernel<<<dim3((p.E + kThreads - 1) / kThreads), dim3(kThreads), 0, stream>>>(p, rng_skip);
    return hipGetLastError();
}

hipError_t launch_move_inputs(const TreeParams& p, const uint32_t* rng_skip, const uint8_t* stall, const int32_t* move_limit,
                              int move_index, bool draw_noise, int32_t* nlegal_out, int32_t* to_play_out, uint32_t* words_out,
                              int32_t* legal_out, const MoveInputsExtra& extra, hipStream_t stream) {
    move_inputs_kernel<<<dim3((p.E + kThreads - 1) / kThreads), dim3(kThreads), 0, stream>>>(
        p, rng_skip, stall, move_limit, move_index, draw_noise ? 1 : 0, nlegal_out, to_play_out, words_out, legal_out, extra);
    return hipGetLastError();
}

// The end of a lock-step move of a batch (mzmcts_moves_end_lockstep): what the fused whole-move kernels do in their
// epilogue -- SelfPlay.select_action (self_play.py:223-246) on the tree's own stream from the root's visit counts, and the
// move's slot of the output ring -- for trees that live in the HBM pools.  One thread per tree: a few dozen operations.
__global__ __launch_bounds__(kThreads) void lockstep_move_finish_kernel(TreeParams p, MoveCtl ctl) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.E) return;
    const int n = p.root_children[e];
    const HbmChild* root = reinterpret_cast<const HbmChild*>(p.blocks + static_cast<size_t>(e) * p.line_stride);   // line (0, e), half 0
    if (ctl.visits)
        for (int i = 0; i < p.A; ++i) ctl.visits[static_cast<size_t>(e) * p.A + i] = (i < n) ? root[i].links.visits : 0;
    if (n == 0) {
        if (ctl.actions) ctl.actions[e] = -1;
        return;
    }
    uint32_t sample_words = 0;
    int32_t pos = p.mt_pos[e];
    const int slot = device_select_action([&](int i) { return root[i].links.visits; }, n, move_temperature(ctl, e),
                                          p.mt_key + static_cast<size_t>(e) * kMtN, &pos, &sample_words);
    if (sample_words) p.mt_pos[e] = pos;
    if (ctl.actions) ctl.actions[e] = slot >= 0 ? p.root_action[static_cast<size_t>(e) * p.A + slot] : slot;
    if (ctl.game_moves) ctl.game_moves[e] += 1;
    if (ctl.tie_words) ctl.tie_words[e] = p.tie_words[e];
    if (ctl.sample_words) ctl.sample_words[e] = sample_words;
    if (ctl.root_value_sum) ctl.root_value_sum[e] = p.root_value_sum[e];
    if (ctl.root_predicted) ctl.root_predicted[e] = p.root_predicted[e];
    if (ctl.max_depth) ctl.max_depth[e] = p.max_depth[e];
    if (ctl.depth_sum) ctl.depth_sum[e] = static_cast<int32_t>(p.depth_sum[e]);
}

hipError_t launch_lockstep_move_finish(const TreeParams& p, const MoveCtl& ctl, hipStream_t stream) {
    lockstep_move_finish_kernel<<<dim3((p.E + kThreads - 1) / kThreads), dim3(kThreads), 0, stream>>>(p, ctl);
    return hipGetLastError();
}

hipError_t launch_gather_dynamics_input(const TreeParams& p, const int64_t* action, float* out, int plane, int action_space,
                                        hipStream_t stream) {
    const int row = p.H + plane;
    if (row <= 512) {   // short rows: several envs per block (t < 4096: the multiply-shift division by row is exact)
        const int per_block = std::max(1, 2048 / row);
        const uint32_t magic = ((1u << 20) + static_cast<uint32_t>(row) - 1u) / static_cast<uint32_t>(row);
        gather_dynamics_rows_kernel<<<dim3((p.E + per_block - 1) / per_block), dim3(256), 0, stream>>>(
            p, action, out, plane, static_cast<float>(action_space), per_block, magic);
        return hipGetLastError();
    }
    int gy = (row + 255) / 256;
    if (gy > 8) gy = 8;
    gather_dynamics_input_kernel<<<dim3(p.E, gy), dim3(256), 0, stream>>>(p, action, out, plane,
                                                                           static_cast<float>(action_space));
    return hipGetLastError();
}

hipError_t launch_expand_roots(const TreeParams& p, const float* value_logits, const float* reward_logits,
                               const float* policy_logits, const float* root_hidden, const double* inj_reward,
                               const double* inj_priors, const double* noise, const uint32_t* rng_skip,
                               bool injected, hipStream_t stream, const LaunchTiming* timing) {
    const int grid = tree_grid(p);
    dispatch_group(p, [&](auto g, auto ch) {
        constexpr int G = decltype(g)::value;
        constexpr int CH = decltype(ch)::value;
        if (injected)
            launch_kernel(expand_roots_kernel<G, CH, true>, dim3(grid), dim3(kThreads), 0, stream, timing, p, value_logits,
                          reward_logits, policy_logits, root_hidden, inj_reward, inj_priors, noise, rng_skip);
        else
            launch_kernel(expand_roots_kernel<G, CH, false>, dim3(grid), dim3(kThreads), 0, stream, timing, p, value_logits,
                          reward_logits, policy_logits, root_hidden, inj_reward, inj_priors, noise, rng_skip);
    });
    return hipGetLastError();
}

hipError_t launch_expand_backup(const TreeParams& p, int sim, const float* value_logits, const float* reward_logits,
                                const float* policy_logits, const double* inj_value, const double* inj_reward,
                                const double* inj_priors, bool injected, hipStream_t stream,
                                const LaunchTiming* timing) {
    const int grid = tree_grid(p);
    dispatch_group(p, [&](auto g, auto ch) {
        constexpr int G = decltype(g)::value;
        constexpr int CH = decltype(ch)::value;
        if (injected)
            launch_kernel(expand_backup_kernel<G, CH, true>, dim3(grid), dim3(kThreads), 0, stream, timing, p, sim,
                          value_logits, reward_logits, policy_logits, inj_value, inj_reward, inj_priors);
        else
            launch_kernel(expand_backup_kernel<G, CH, false>, dim3(grid), dim3(kThreads), 0, stream, timing, p, sim,
                          value_logits, reward_logits, policy_logits, inj_value, inj_reward, inj_priors);
    });
    return hipGetLastError();
}

// expand_backup(sim) + select(sim + 1) in one launch; the caller still runs the stand-alone gather for large hidden states
hipError_t launch_expand_backup_select(const TreeParams& p, int sim, const float* value_logits, const float* reward_logits,
                                       const float* policy_logits, const double* inj_value, const double* inj_reward,
                                       const double* inj_priors, bool injected, float* hidden_out, int64_t* action_out,
                                       hipStream_t stream, const LaunchTiming* timing) {
    const size_t lds = sizeof(double) * 2 * (static_cast<size_t>(p.S) + 1);
    const int grid = tree_grid(p);
    const bool fuse = p.H <= kFuseGatherMaxFloats;
    dispatch_group(p, [&](auto g, auto ch) {
        constexpr int G = decltype(g)::value;
        constexpr int CH = decltype(ch)::value;
#define MZ_STEP(INJ, FG)                                                                                                  \
    launch_kernel(expand_backup_select_kernel<G, CH, INJ, FG>, dim3(grid), dim3(kThreads), lds, stream, timing, p, sim,    \
                  value_logits, reward_logits, policy_logits, inj_value, inj_reward, inj_priors, hidden_out, action_out)
        if (injected) {
            if (fuse) MZ_STEP(true, true); else MZ_STEP(true, false);
        } else {
            if (fuse) MZ_STEP(false, true); else MZ_STEP(false, false);
        }
#undef MZ_STEP
    });
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    if (!fuse && hidden_out) {
        const int per_row = (p.H + 3) / 4;
        int gy = (per_row + 255) / 256;
        if (gy > 8) gy = 8;
        gather_hidden_kernel<<<dim3(p.E, gy), dim3(256), 0, stream>>>(p, hidden_out);
        err = hipGetLastError();
    }
    return err;
}

hipError_t launch_copy_slab(const float* src, float* dst, size_t n, hipStream_t stream) {
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    copy_slab_kernel<<<dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream>>>(src, dst, n);
    return hipGetLastError();
}

hipError_t launch_seed_streams(uint32_t* keys, int32_t* pos, const uint32_t* seeds, int E, hipStream_t stream) {
    seed_streams_kernel<<<dim3((E + 255) / 256), dim3(256), 0, stream>>>(keys, pos, seeds, E);
    return hipGetLastError();
}

static int host_list_entries(const FcPhase* list, int n, int G) {
    int total = 0;
    for (int p = 0; p < n; ++p) total += (list[p].total_out + 4 * G - 1) / (4 * G) * 4 * G;
    return total;
}
static int host_init_entries(const FcNet& net, int G) {
    return host_list_entries(net.init_pre, net.n_init_pre, G) + host_list_entries(net.init_post, net.n_init_post, G);
}
static int host_table_entries(const FcNet& net, int G) {
    return host_init_entries(net, G) + host_list_entries(net.rec_pre, net.n_rec_pre, G) +
           host_list_entries(net.rec_post, net.n_rec_post, G);
}

hipError_t launch_fc_inference(const TreeParams& p, const FcNet& net, const float* weights, bool initial, const float* in,
                               const int64_t* action, float* value_logits, float* reward_logits, float* policy_logits,
                               float* hidden_out, hipStream_t stream) {
    const int grid = tree_grid(p);
    dispatch_group(p, [&](auto g, auto) {
        constexpr int G = decltype(g)::value;
        const size_t lds = sizeof(float) * (static_cast<size_t>((net.n_weights_lds + 3) & ~3) +
                                            static_cast<size_t>(net.scratch_floats) * (kThreads / G)) +
                           sizeof(NeuronDesc) * static_cast<size_t>(host_table_entries(net, G));
        if (initial)
            fc_inference_kernel<G, true><<<dim3(grid), dim3(kThreads), lds, stream>>>(
                net, weights, p.E, in, action, value_logits, reward_logits, policy_logits, hidden_out);
        else
            fc_inference_kernel<G, false><<<dim3(grid), dim3(kThreads), lds, stream>>>(
                net, weights, p.E, in, action, value_logits, reward_logits, policy_logits, hidden_out);
    });
    return hipGetLastError();
}

// LDS plan of the fused kernel for this (params, net, group); returns false if it cannot fit.
bool plan_fused_layout(const TreeParams& p, const FcNet& net, bool want_hidden_in_lds, size_t lds_limit,
                       FusedLayout* out) {
    FusedLayout lay{};
    auto align16 = [](size_t v) { return (v + 15) / 16 * 16; };
    size_t off = align16(sizeof(double) * 2 * (static_cast<size_t>(p.S) + 1));
    lay.off_weights = static_cast<uint32_t>(off);
    off = align16(off + sizeof(float) * static_cast<size_t>(net.n_weights_lds));
    lay.off_table = static_cast<uint32_t>(off);
    lay.off_rec_table = static_cast<uint32_t>(off + sizeof(NeuronDesc) * static_cast<size_t>(host_init_entries(net, p.group)));
    off = align16(off + sizeof(NeuronDesc) * static_cast<size_t>(host_table_entries(net, p.group)));
    lay.off_trees = static_cast<uint32_t>(off);
    size_t t = static_cast<size_t>(p.S + 1) * p.block_stride;
    lay.off_path = static_cast<uint32_t>(t);
    t = align16(t + sizeof(int32_t) * static_cast<size_t>(p.S));
    lay.off_scratch = static_cast<uint32_t>(t);
    t = align16(t + sizeof(float) * static_cast<size_t>(net.scratch_floats));
    const size_t trees = kThreads / p.group;
    const size_t hidden_bytes = align16(sizeof(float) * static_cast<size_t>(p.S + 1) * net.enc);
    if (want_hidden_in_lds && off + trees * (t + hidden_bytes) <= lds_limit) {
        lay.off_hidden = static_cast<uint32_t>(t);
        t += hidden_bytes;
    } else {
        lay.off_hidden = 0xffffffffu;
    }
    lay.tree_bytes = static_cast<uint32_t>(t);
    lay.total_bytes = static_cast<uint32_t>(off + trees * t);
    *out = lay;
    return off + trees * t <= lds_limit;
}

hipError_t launch_search_fused_fc(const TreeParams& p, const FcNet& net, const FusedLayout& lay, const float* weights,
                                  const float* observations, const MoveCtl& ctl, int n_sims, hipStream_t stream,
                                  const LaunchTiming* timing) {
    const int grid = tree_grid(p);
    hipError_t attr_err = hipSuccess;
    dispatch_group(p, [&](auto g, auto ch) {
        constexpr int G = decltype(g)::value;
        constexpr int CH = decltype(ch)::value;
        auto kernel = search_fused_fc_kernel<G, CH>;
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lay.total_bytes));
        if (attr_err != hipSuccess) return;
        launch_kernel(kernel, dim3(grid), dim3(kThreads), lay.total_bytes, stream, timing, p, net, lay, weights,
                      observations, ctl, n_sims);
    });
    if (attr_err != hipSuccess) return attr_err;
    return hipGetLastError();
}

}  // namespace mz
