// mcts_kernels.hip -- batched MCTS tree kernels for CDNA4 (gfx950, wave64).
//
// Mapping: a tree is owned by a group of G = pow2 >= min(A,64) adjacent lanes of ONE wavefront,
// one lane per child (lanes loop when A > 64).  A 64-thread workgroup (one wave) carries 64/G
// trees; tree -> workgroup is identical in every kernel, so a tree's blocks are re-touched from the
// same XCD (workgroups are dealt round-robin over the 8 XCDs) and stay in that XCD's L2.
//
// Arithmetic contract (bit-exact with the reference's Python floats): every UCB / backup
// operation is an IEEE fp64 +,-,*,/ in the reference's order (this file is compiled with
// -ffp-contract=off, so nothing is fused); log/sqrt of the parent visit count come from a host
// libm table staged in LDS; ties are detected with == on the fp64 scores and broken with the
// tree's MT19937 stream exactly as numpy.random.choice does.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <type_traits>

#include "np_legacy_rng.h"
#include "tree_layout.h"

namespace mz {

constexpr int kThreads = 64;   // one wavefront per workgroup
constexpr int kMaxChunks = 4;  // A <= 256
constexpr int kStageLevels = 16;

template <int G>
__device__ __forceinline__ double group_max(double v) {
#pragma unroll
    for (int m = G / 2; m > 0; m >>= 1) v = fmax(v, __shfl_xor(v, m, G));
    return v;
}
template <int G>
__device__ __forceinline__ float group_maxf(float v) {
#pragma unroll
    for (int m = G / 2; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, G));
    return v;
}
template <int G>
__device__ __forceinline__ float group_sumf(float v) {
#pragma unroll
    for (int m = G / 2; m > 0; m >>= 1) v = v + __shfl_xor(v, m, G);
    return v;
}

__device__ __forceinline__ uint8_t* block_ptr(const TreeParams& p, int k, int e) {
    return p.blocks + (static_cast<size_t>(k) * p.E + e) * p.block_stride;
}

// self_play.py:381-405 ucb_score, one child.
__device__ __forceinline__ double ucb_score(double pb_log, double pb_sqrt, const ChildStats& s,
                                            const ChildLinks& l, double discount, bool two_player,
                                            double mn, double mx) {
    double pb_c = pb_log;
    pb_c = pb_c * (pb_sqrt / static_cast<double>(l.visits + 1));
    const double prior_score = pb_c * s.prior;
    double value_score = 0.0;
    if (l.visits > 0) {
        double q = s.value_sum / static_cast<double>(l.visits);
        if (two_player) q = -q;
        const double v = static_cast<double>(l.reward) + discount * q;
        value_score = (mx > mn) ? (v - mn) / (mx - mn) : v;
    }
    return prior_score + value_score;
}

// -------------------------------------------------------------------------------------------------
// select: descend every tree from its root to a leaf (self_play.py:321-335, 364-379) and gather the
// parent hidden state + action of the leaf into the inference batch (self_play.py:339-343).
// -------------------------------------------------------------------------------------------------
template <int G, int CH, bool FUSE_GATHER>
__global__ __launch_bounds__(kThreads) void select_kernel(TreeParams p, int sim, float* __restrict__ hidden_out,
                                                          int64_t* __restrict__ action_out) {
    extern __shared__ double pbc_table[];  // [2][S+1]
    for (int i = threadIdx.x; i <= p.S; i += kThreads) {
        pbc_table[i] = p.pbc_log[i];
        pbc_table[p.S + 1 + i] = p.pbc_sqrt[i];
    }
    __syncthreads();

    constexpr int kTrees = kThreads / G;
    const int e = blockIdx.x * kTrees + threadIdx.x / G;
    const int j = threadIdx.x % G;
    const int group_base = threadIdx.x - j;  // lane of the group leader inside the wave
    if (e >= p.E) return;
    int n_children = p.root_children[e];
    if (n_children == 0) {  // inactive tree: keep the batch row defined
        if (j == 0) {
            p.path_len[e] = 0;
            p.leaf_parent[e] = 0;
            if (action_out) action_out[e] = 0;
        }
        if (FUSE_GATHER && hidden_out)
            for (int i = j; i < p.H; i += G) hidden_out[static_cast<size_t>(e) * p.H + i] = 0.f;
        return;
    }

    const bool two_player = p.P == 2;
    const MinMax mm = p.min_max[e];
    uint32_t* key = p.mt_key + static_cast<size_t>(e) * kMtN;
    int32_t mt_pos = (j == 0) ? p.mt_pos[e] : 0;
    uint32_t words = 0;

    int k = 0;        // expanded-node index of the current parent
    int N = sim;      // its visit count: the root has been visited once per finished simulation
    int depth = 0;
    int slot = 0;
    for (;;) {
        const uint8_t* blk = block_ptr(p, k, e);
        const ChildStats* stats = reinterpret_cast<const ChildStats*>(blk);
        const ChildLinks* links = reinterpret_cast<const ChildLinks*>(blk + p.links_offset);
        const double pb_log = pbc_table[N];
        const double pb_sqrt = pbc_table[p.S + 1 + N];

        double score[CH];
        ChildLinks lk[CH];
        double best = -INFINITY;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int child = c * G + j;
            score[c] = -INFINITY;
            lk[c] = ChildLinks{0.f, 0, -1, 0};
            if (child < n_children) {
                const ChildStats st = stats[child];
                lk[c] = links[child];
                score[c] = ucb_score(pb_log, pb_sqrt, st, lk[c], p.discount, two_player, mm.minimum, mm.maximum);
                best = fmax(best, score[c]);
            }
        }
        best = group_max<G>(best);

        // tie list in child order (self_play.py:372-378)
        unsigned long long tie_mask[CH];
        int n_ties = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const bool is_max = (c * G + j < n_children) && (score[c] == best);
            const unsigned long long ballot = __ballot(is_max);
            tie_mask[c] = (G == 64) ? ballot : ((ballot >> group_base) & ((1ull << G) - 1ull));
            n_ties += __popcll(tie_mask[c]);
        }
        int pick = 0;
        if (n_ties > 1) {
            uint32_t r = 0;
            if (j == 0) r = mt_below(key, &mt_pos, static_cast<uint32_t>(n_ties), &words);
            pick = static_cast<int>(__shfl(r, 0, G));
        } else if (n_ties == 0) {  // NaN scores: the reference would raise; flag and take slot 0
            if (j == 0) atomicOr(p.error_flag, 1);
            tie_mask[0] = 1ull;
        }
        int sel_visits = 0, sel_child = -1;
        {
            int remaining = pick;
            bool found = false;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                unsigned long long m = tie_mask[c];
                const int cnt = __popcll(m);
                if (!found && remaining < cnt) {
                    for (int i = 0; i < remaining; ++i) m &= m - 1ull;
                    const int bit = __ffsll(static_cast<long long>(m)) - 1;
                    slot = c * G + bit;
                    sel_visits = __shfl(lk[c].visits, bit, G);
                    sel_child = __shfl(lk[c].child_node, bit, G);
                    found = true;
                } else if (!found) {
                    remaining -= cnt;  // (found / remaining are uniform across the group's lanes)
                }
            }
        }
        if (j == 0) {
            p.path[static_cast<size_t>(depth) * p.E + e] = (k << 16) | slot;
            if (p.path_ties) p.path_ties[static_cast<size_t>(depth) * p.E + e] = n_ties;
        }
        ++depth;
        if (sel_child < 0) break;  // reached a node that is not expanded yet
        if (depth > sim) {         // cannot happen on a consistent tree (only sim+1 nodes are expanded);
            if (j == 0) atomicOr(p.error_flag, 2);  // guarantees every wave leaves the loop regardless
            break;
        }
        k = sel_child;
        N = sel_visits;
        n_children = p.A;
    }

    if (j == 0) {
        p.path_len[e] = depth;
        p.leaf_parent[e] = k;
        if (words) {
            p.mt_pos[e] = mt_pos;
            p.tie_words[e] += words;
        }
        if (action_out) action_out[e] = (depth == 1) ? p.root_action[static_cast<size_t>(e) * p.A + slot] : slot;
    }
    if (FUSE_GATHER && hidden_out) {
        const float* src = p.hidden + (static_cast<size_t>(k) * p.E + e) * p.H;
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

// Stand-alone gather for large hidden states (ResNet planes): one workgroup per tree row chunk,
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
// models.py:641-662 support_to_scalar, fp32, torch's operation order; the F logits of one tree are
// spread over the G lanes of its group.
// -------------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ float support_to_scalar_group(const float* __restrict__ logits, int F, int support, int j) {
    float m = -INFINITY;
    for (int i = j; i < F; i += G) m = fmaxf(m, logits[i]);
    m = group_maxf<G>(m);
    float s = 0.f;
    for (int i = j; i < F; i += G) s += expf(logits[i] - m);
    s = group_sumf<G>(s);
    const float inv = 1.0f / s;
    float acc = 0.f;
    for (int i = j; i < F; i += G) acc += static_cast<float>(i - support) * (expf(logits[i] - m) * inv);
    const float x = group_sumf<G>(acc);
    const float u = (fabsf(x) + 1.0f) + 0.001f;
    const float w = 0.004f * u;
    const float r = sqrtf(1.0f + w) - 1.0f;
    const float q = r / 0.002f;
    const float y = q * q - 1.0f;
    const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    return sgn * y;
}

// fp32 softmax over the children of one group (Node.expand, self_play.py:461-463): max, exp,
// sum, multiply by the reciprocal -- torch's CPU kernel order -- widened to fp64 like .tolist().
template <int G, int CH>
__device__ __forceinline__ void group_softmax(const float (&logit)[CH], const bool (&valid)[CH], double (&prior)[CH]) {
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CH; ++c)
        if (valid[c]) m = fmaxf(m, logit[c]);
    m = group_maxf<G>(m);
    float ex[CH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        ex[c] = valid[c] ? expf(logit[c] - m) : 0.f;
        s += ex[c];
    }
    s = group_sumf<G>(s);
    const float inv = 1.0f / s;
#pragma unroll
    for (int c = 0; c < CH; ++c) prior[c] = static_cast<double>(ex[c] * inv);
}

// -------------------------------------------------------------------------------------------------
// expand_roots: root.expand over the legal actions + exploration noise (self_play.py:293-315,
// 452-477) and reset of the per-search state (MinMaxStats(), max_tree_depth; self_play.py:317-319).
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

    if (j == 0) {
        // advance the device RNG past the words the host mirror consumed (Dirichlet draw of this
        // move, action sampling of the previous one)
        uint32_t skip = rng_skip ? rng_skip[e] : 0u;
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

    uint8_t* blk = block_ptr(p, 0, e);
    ChildStats* stats = reinterpret_cast<ChildStats*>(blk);
    ChildLinks* links = reinterpret_cast<ChildLinks*>(blk + p.links_offset);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int child = c * G + j;
        if (child < p.A) {
            double pr = prior[c];
            if (valid[c] && noise) {
                // prior * (1 - frac) + n * frac   (self_play.py:477)
                const double keep = pr * (1 - p.noise_frac);
                const double add = noise[static_cast<size_t>(e) * p.A + child] * p.noise_frac;
                pr = keep + add;
            }
            stats[child] = ChildStats{0.0, pr};
            links[child] = ChildLinks{0.f, 0, -1, 0};
        }
    }
    if (root_hidden) {
        const float* src = root_hidden + static_cast<size_t>(e) * p.H;
        float* dst = p.hidden + static_cast<size_t>(e) * p.H;  // slab 0
        for (int i = j; i < p.H; i += G) dst[i] = src[i];
    }
}

// -------------------------------------------------------------------------------------------------
// expand_backup: decode value/reward, expand the leaf over the full action space, back the value
// up the search path with min-max statistics (self_play.py:344-356, 407-431, 452-466, 560-562).
//
// The backup walks a path whose addresses are all known (written by select), so the group's lanes
// first stage the (value_sum, visits, reward) triples of up to kStageLevels path nodes into LDS with
// independent loads in flight, the group leader then runs the inherently sequential value
// recursion out of LDS, and the lanes write the updated statistics back.
// -------------------------------------------------------------------------------------------------
struct StagedNode {
    double value_sum;
    float reward;
    int32_t visits;
};

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
    const int depth = p.path_len[e];  // >= 1
    const int k_new = sim + 1;

    // ---- decode the network heads -------------------------------------------------------------
    double value;
    float reward_f;
    if (INJECTED) {
        value = inj_value[e];
        reward_f = static_cast<float>(inj_reward[e]);
    } else {
        value = static_cast<double>(
            support_to_scalar_group<G>(value_logits + static_cast<size_t>(e) * p.F, p.F, p.support, j));
        reward_f = support_to_scalar_group<G>(reward_logits + static_cast<size_t>(e) * p.F, p.F, p.support, j);
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

    // ---- expand: children of the new node (slab k_new is written contiguously over trees) ------
    {
        uint8_t* blk = block_ptr(p, k_new, e);
        ChildStats* stats = reinterpret_cast<ChildStats*>(blk);
        ChildLinks* links = reinterpret_cast<ChildLinks*>(blk + p.links_offset);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int child = c * G + j;
            if (valid[c]) {
                stats[child] = ChildStats{0.0, prior[c]};
                links[child] = ChildLinks{0.f, 0, -1, 0};
            }
        }
    }

    // ---- backup ---------------------------------------------------------------------------------
    const bool two_player = p.P == 2;
    const double discount = p.discount;
    MinMax mm = p.min_max[e];
    const double reward = static_cast<double>(reward_f);

    // leaf (tree depth == depth): first visit, value_sum was 0; it is `to_play`'s own node
    const int leaf_packed = p.path[static_cast<size_t>(depth - 1) * p.E + e];
    if (j == 0) {
        uint8_t* blk = block_ptr(p, leaf_packed >> 16, e);
        const int slot = leaf_packed & 0xffff;
        ChildStats* st = reinterpret_cast<ChildStats*>(blk) + slot;
        ChildLinks* lk = reinterpret_cast<ChildLinks*>(blk + p.links_offset) + slot;
        const double vs = 0.0 + value;
        st->value_sum = vs;
        *lk = ChildLinks{reward_f, 1, k_new, 0};
        const double q = vs / 1.0;
        const double seen = two_player ? (reward + discount * -q) : (reward + discount * q);
        mm.maximum = fmax(mm.maximum, seen);
        mm.minimum = fmin(mm.minimum, seen);
        value = (two_player ? -reward : reward) + discount * value;
    }

    // interior path nodes, leaf-side first, kStageLevels at a time
    for (int hi = depth - 2; hi >= 0; hi -= kStageLevels) {
        const int count = (hi + 1 < kStageLevels) ? hi + 1 : kStageLevels;  // levels hi, hi-1, ...
        for (int i = j; i < count; i += G) {
            const int packed = p.path[static_cast<size_t>(hi - i) * p.E + e];
            const uint8_t* blk = block_ptr(p, packed >> 16, e);
            const int slot = packed & 0xffff;
            const ChildStats* st = reinterpret_cast<const ChildStats*>(blk) + slot;
            const ChildLinks lk = *(reinterpret_cast<const ChildLinks*>(blk + p.links_offset) + slot);
            staged[tree_in_block][i] = StagedNode{st->value_sum, lk.reward, lk.visits};
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (j == 0) {
            for (int i = 0; i < count; ++i) {
                StagedNode n = staged[tree_in_block][i];
                const int level = hi - i;  // node's tree depth is level + 1
                const double r = static_cast<double>(n.reward);
                if (!two_player) {
                    n.value_sum += value;
                    n.visits += 1;
                    const double q = n.value_sum / static_cast<double>(n.visits);
                    const double seen = r + discount * q;
                    mm.maximum = fmax(mm.maximum, seen);
                    mm.minimum = fmin(mm.minimum, seen);
                    value = r + discount * value;
                } else {
                    const bool same = ((depth - (level + 1)) & 1) == 0;  // node.to_play == to_play
                    n.value_sum += same ? value : -value;
                    n.visits += 1;
                    const double q = n.value_sum / static_cast<double>(n.visits);
                    const double seen = r + discount * -q;
                    mm.maximum = fmax(mm.maximum, seen);
                    mm.minimum = fmin(mm.minimum, seen);
                    value = (same ? -r : r) + discount * value;
                }
                staged[tree_in_block][i] = n;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = j; i < count; i += G) {
            const int packed = p.path[static_cast<size_t>(hi - i) * p.E + e];
            uint8_t* blk = block_ptr(p, packed >> 16, e);
            const int slot = packed & 0xffff;
            const StagedNode n = staged[tree_in_block][i];
            (reinterpret_cast<ChildStats*>(blk) + slot)->value_sum = n.value_sum;
            (reinterpret_cast<ChildLinks*>(blk + p.links_offset) + slot)->visits = n.visits;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // root (tree depth 0), then per-search statistics
    if (j == 0) {
        double rvs = p.root_value_sum[e];
        const double r = p.root_reward[e];
        const double n_root = static_cast<double>(sim + 1);
        double seen;
        if (!two_player) {
            rvs += value;
            seen = r + discount * (rvs / n_root);
        } else {
            const bool same = (depth & 1) == 0;
            rvs += same ? value : -value;
            seen = r + discount * -(rvs / n_root);
        }
        mm.maximum = fmax(mm.maximum, seen);
        mm.minimum = fmin(mm.minimum, seen);
        p.root_value_sum[e] = rvs;
        p.min_max[e] = mm;
        if (depth > p.max_depth[e]) p.max_depth[e] = depth;
        p.depth_sum[e] += depth;
    }
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
// launchers
// -------------------------------------------------------------------------------------------------
static int group_width(int A) {
    int g = 1;
    while (g < A && g < 64) g <<= 1;
    return g;
}

template <int V>
using IntC = std::integral_constant<int, V>;

// Calls fn(IntC<G>{}, IntC<CH>{}) with the lane-group width / chunk count for A actions.
template <typename Fn>
static void dispatch_group(int A, Fn&& fn) {
    if (A > 64) {
        fn(IntC<64>{}, IntC<kMaxChunks>{});
        return;
    }
    switch (group_width(A)) {
        case 1: fn(IntC<1>{}, IntC<1>{}); break;
        case 2: fn(IntC<2>{}, IntC<1>{}); break;
        case 4: fn(IntC<4>{}, IntC<1>{}); break;
        case 8: fn(IntC<8>{}, IntC<1>{}); break;
        case 16: fn(IntC<16>{}, IntC<1>{}); break;
        case 32: fn(IntC<32>{}, IntC<1>{}); break;
        default: fn(IntC<64>{}, IntC<1>{}); break;
    }
}

static inline int tree_grid(const TreeParams& p) {
    const int g = group_width(p.A);
    const int trees = kThreads / g;
    return (p.E + trees - 1) / trees;
}

constexpr int kFuseGatherMaxFloats = 64;

// Launch with optional HIP events bound to the dispatch itself (hipExtLaunchKernel records the
// kernel's own start / end timestamps into the events, so profiling-mode timings are kernel
// durations, comparable with rocprofv3's, not launch-to-launch gaps).
template <typename Kernel, typename... Args>
static void launch_kernel(Kernel kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                          const LaunchTiming* timing, Args... args) {
    if (timing && timing->start)
        hipExtLaunchKernelGGL(kernel, grid, block, static_cast<uint32_t>(lds), stream, timing->start, timing->stop, 0u,
                              args...);
    else
        kernel<<<grid, block, lds, stream>>>(args...);
}

hipError_t launch_select(const TreeParams& p, int sim, float* hidden_out, int64_t* action_out, hipStream_t stream,
                         const LaunchTiming* timing) {
    const size_t lds = sizeof(double) * 2 * (static_cast<size_t>(p.S) + 1);
    const int grid = tree_grid(p);
    const bool fuse = p.H <= kFuseGatherMaxFloats;
    dispatch_group(p.A, [&](auto g, auto ch) {
        constexpr int G = decltype(g)::value;
        constexpr int CH = decltype(ch)::value;
        if (fuse)
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

hipError_t launch_expand_roots(const TreeParams& p, const float* value_logits, const float* reward_logits,
                               const float* policy_logits, const float* root_hidden, const double* inj_reward,
                               const double* inj_priors, const double* noise, const uint32_t* rng_skip,
                               bool injected, hipStream_t stream, const LaunchTiming* timing) {
    const int grid = tree_grid(p);
    dispatch_group(p.A, [&](auto g, auto ch) {
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
    dispatch_group(p.A, [&](auto g, auto ch) {
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

}  // namespace mz
