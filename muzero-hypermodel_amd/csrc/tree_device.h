// tree_device.h -- device-side building blocks of the search, shared by the lock-step kernels
// (tree in HBM, one kernel per phase) and the fused whole-move kernel (tree resident in LDS).
//
// Every function is written for a lane GROUP: the G = pow2 lanes of one wavefront that own a tree,
// lane j handling child j (+ c*G for c < CH when the action space is wider than the group).  All
// control flow inside these functions depends only on group-uniform values, so the intra-group
// shuffles / ballots are convergent for the group even while other groups of the wave have left.
//
// `Acc` is the storage policy of one tree:
//     ChildStats* stats(int k), ChildLinks* links(int k)      child block k of this tree
//     void path_store(int level, int packed), int path_load(int level)
//     static constexpr bool kInLds                             tree lives in LDS (fused kernel)
#pragma once
#include <hip/hip_runtime.h>

#include "np_legacy_rng.h"
#include "tree_layout.h"

namespace mz {

constexpr int kStageLevels = 16;

// ---- intra-group exchange ---------------------------------------------------------------------------
// partner<M>(v): the value held by the lane this lane pairs with at butterfly step M.  Steps 1 and 2 are
// quad permutes, 4 and 8 the half-row / row mirrors -- all DPP modifiers, i.e. plain VALU moves with no
// trip through the LDS crossbar (a ds_bpermute costs an LDS round trip, ~100+ cycles on the dependent
// chain of a one-wave-per-SIMD kernel); 16 and 32 fall back to ds_bpermute.  Mirrors pair every lane
// with a lane of the opposite half, which is all a reduction of already half-uniform values needs.
template <int M>
__device__ __forceinline__ int partner_bits(int v) {
    if constexpr (M == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (M == 4) return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    else if constexpr (M == 8) return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror
    else return __shfl_xor(v, M, 64);
}
template <int M>
__device__ __forceinline__ float partner(float v) {
    return __builtin_bit_cast(float, partner_bits<M>(__builtin_bit_cast(int, v)));
}
template <int M>
__device__ __forceinline__ double partner(double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = partner_bits<M>(static_cast<int>(bits & 0xffffffffll));
    const int hi = partner_bits<M>(static_cast<int>(bits >> 32));
    return __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

#define MZ_BUTTERFLY(G_, LIMIT_, EXPR)                                  \
    do {                                                                \
        if constexpr ((G_) >= 2) { if ((LIMIT_) >= 2) { constexpr int M = 1; EXPR; } }    \
        if constexpr ((G_) >= 4) { if ((LIMIT_) >= 4) { constexpr int M = 2; EXPR; } }    \
        if constexpr ((G_) >= 8) { if ((LIMIT_) >= 8) { constexpr int M = 4; EXPR; } }    \
        if constexpr ((G_) >= 16) { if ((LIMIT_) >= 16) { constexpr int M = 8; EXPR; } }  \
        if constexpr ((G_) >= 32) { if ((LIMIT_) >= 32) { constexpr int M = 16; EXPR; } } \
        if constexpr ((G_) >= 64) { if ((LIMIT_) >= 64) { constexpr int M = 32; EXPR; } } \
    } while (0)

// max over the first `span` lanes of the group (span = pow2 <= G; lanes beyond it hold -inf and their
// result is not used): the UCB arg-max only ever involves pow2(A) lanes however wide the group is.
template <int G>
__device__ __forceinline__ double group_max(double v, int span = G) {
    MZ_BUTTERFLY(G, span, v = fmax(v, partner<M>(v)));
    return v;
}
template <int G>
__device__ __forceinline__ float group_maxf(float v) {
    MZ_BUTTERFLY(G, G, v = fmaxf(v, partner<M>(v)));
    return v;
}
template <int G>
__device__ __forceinline__ float group_minf(float v) {
    MZ_BUTTERFLY(G, G, v = fminf(v, partner<M>(v)));
    return v;
}
template <int G>
__device__ __forceinline__ float group_sumf(float v) {
    MZ_BUTTERFLY(G, G, v = v + partner<M>(v));
    return v;
}

// wave-level ordering point for values handed between lanes of a group through memory
__device__ __forceinline__ void group_memory_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- storage policies ---------------------------------------------------------------------------
// Blocks are addressed by LOCATION loc = 2 * slab + half (ChildLinks::block_loc); the root's is 0.
//
// In HBM a block is an array of 32-byte CHILD RECORDS { ChildStats, ChildLinks } (tree_layout.h): everything the
// backup reads and rewrites of a path node -- value_sum, reward, visits -- sits in ONE 32-byte sector, the unit the
// memory system fetches and writes back.  Records<T> is a pointer to member T of consecutive records: it indexes
// and offsets like the plain T* the LDS-resident trees hand out (which keep the two members as separate arrays).
template <typename T>
struct Records {
    uint8_t* base;
    __device__ __forceinline__ T& operator[](int child) const {
        return *reinterpret_cast<T*>(base + static_cast<uint32_t>(child) * kChildRecordBytes);
    }
    __device__ __forceinline__ T* operator+(int child) const { return &(*this)[child]; }
};

struct GlobalTree {
    static constexpr bool kInLds = false;
    uint8_t* blocks;       // p.blocks + e * line_stride
    size_t slab_stride;    // E * line_stride
    int32_t* path;         // p.path + e
    int E;
    __device__ __forceinline__ uint8_t* block(int loc) const {
        return blocks + static_cast<size_t>(loc >> 1) * slab_stride + static_cast<size_t>(loc & 1) * 64u;
    }
    __device__ __forceinline__ Records<ChildStats> stats(int loc) const { return Records<ChildStats>{block(loc)}; }
    __device__ __forceinline__ Records<ChildLinks> links(int loc) const {
        return Records<ChildLinks>{block(loc) + sizeof(ChildStats)};
    }
    __device__ __forceinline__ void path_store(int level, int packed) const {
        path[static_cast<size_t>(level) * E] = packed;
    }
    __device__ __forceinline__ int path_load(int level) const { return path[static_cast<size_t>(level) * E]; }
};

__device__ __forceinline__ GlobalTree global_tree(const TreeParams& p, int e) {
    return GlobalTree{p.blocks + static_cast<size_t>(e) * p.line_stride, static_cast<size_t>(p.E) * p.line_stride,
                      p.path + e, p.E};
}

struct LdsTree {
    static constexpr bool kInLds = true;
    uint8_t* blocks;       // this tree's (S+1) blocks, contiguous in LDS
    uint32_t block_stride;
    uint32_t links_offset;
    int32_t* path;         // this tree's S path words in LDS
    // (a tree in LDS keeps block k at slab k: loc = 2 k)
    __device__ __forceinline__ ChildStats* stats(int loc) const {
        return reinterpret_cast<ChildStats*>(blocks + static_cast<uint32_t>(loc >> 1) * block_stride);
    }
    __device__ __forceinline__ ChildLinks* links(int loc) const {
        return reinterpret_cast<ChildLinks*>(blocks + static_cast<uint32_t>(loc >> 1) * block_stride + links_offset);
    }
    __device__ __forceinline__ void path_store(int level, int packed) const { path[level] = packed; }
    __device__ __forceinline__ int path_load(int level) const { return path[level]; }
};

// ---- self_play.py:381-405 ucb_score, one child ----------------------------------------------------
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

struct Descent {
    int depth;        // number of select steps == tree depth of the leaf
    int parent;       // expanded-node index of the leaf's parent (its hidden-state slab)
    int slot;         // child slot of the leaf inside its parent's block
    int parent_loc;   // block location of the leaf's parent
    int parent_visits;  // its visit count before this simulation
};

// What one select_child call (self_play.py:364-379) decides, known to every lane of the group.
struct LevelPick {
    int slot;     // chosen child slot
    int visits;   // its visit count
    int child;    // its expanded-node index, -1 = not expanded (the descent ends here)
    int loc;      // where its own block lives
    int n_ties;   // size of the tie list the choice was drawn from
};

// select_child (self_play.py:364-379) over the child records a lane group holds: lane j has children c * G + j in
// st[c] / lk[c] (records of slots >= n_children are ignored).  N = the node's visit count.  pbc_table: [2][S+1] in
// LDS.  The group leader owns the RNG cursor (mt_pos / words).
template <int G, int CH>
__device__ __forceinline__ LevelPick pick_child(const ChildStats (&st)[CH], ChildLinks (&lk)[CH], const double* pbc_table,
                                                int S, int N, int n_children, int span, const MinMax& mm, double discount,
                                                bool two_player, uint32_t* mt_key, int32_t& mt_pos, uint32_t& words, int j,
                                                int group_base, int32_t* error_flag) {
    const double pb_log = pbc_table[N];
    const double pb_sqrt = pbc_table[S + 1 + N];

    double score[CH];
    double best = -INFINITY;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int child = c * G + j;
        score[c] = -INFINITY;
        if (child < n_children) {
            score[c] = ucb_score(pb_log, pb_sqrt, st[c], lk[c], discount, two_player, mm.minimum, mm.maximum);
            best = fmax(best, score[c]);
        } else {
            lk[c] = ChildLinks{0.f, 0, -1, 0};
        }
    }
    best = group_max<G>(best, span);

    // tie list in child order (self_play.py:372-378)
    unsigned long long tie_mask[CH];
    int n_ties = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const bool is_max = (c * G + j < n_children) && (score[c] == best);
        const unsigned long long ballot = __ballot(is_max);
        if constexpr (G == 64)
            tie_mask[c] = ballot;
        else
            tie_mask[c] = (ballot >> group_base) & ((1ull << G) - 1ull);
        n_ties += __popcll(tie_mask[c]);
    }
    int pick = 0;
    if (n_ties > 1) {
        uint32_t r = 0;
        if (j == 0) r = mt_below(mt_key, &mt_pos, static_cast<uint32_t>(n_ties), &words);
        pick = static_cast<int>(__shfl(r, 0, G));
    } else if (n_ties == 0) {  // NaN scores: the reference would raise; flag and take slot 0
        if (j == 0) atomicOr(error_flag, 1);
        tie_mask[0] = 1ull;
    }
    int slot = 0, sel_visits = 0, sel_child = -1, sel_loc = 0;
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
                // the chosen lane publishes its child's links to the group with an OR-butterfly of DPP
                // moves (child_node is biased by 1 so that "not expanded" (-1) contributes zero bits)
                int pub_visits = (j == bit) ? lk[c].visits : 0;
                int pub_child = (j == bit) ? lk[c].child_node + 1 : 0;
                int pub_loc = (j == bit) ? lk[c].block_loc : 0;
                if constexpr (G <= 16) {
                    // over the whole group, so that every lane ends up with the values
                    MZ_BUTTERFLY(G, G, (pub_visits |= partner_bits<M>(pub_visits),
                                        pub_child |= partner_bits<M>(pub_child),
                                        pub_loc |= partner_bits<M>(pub_loc)));
                } else {
                    pub_visits = __shfl(lk[c].visits, bit, G);
                    pub_child = __shfl(lk[c].child_node, bit, G) + 1;
                    pub_loc = __shfl(lk[c].block_loc, bit, G);
                }
                sel_visits = pub_visits;
                sel_child = pub_child - 1;
                sel_loc = pub_loc;
                found = true;
            } else if (!found) {
                remaining -= cnt;  // (found / remaining are uniform across the group's lanes)
            }
        }
    }
    return LevelPick{slot, sel_visits, sel_child, sel_loc, n_ties};
}

// One level of the descent: select_child at the node whose block lives at `loc` (N = its visit count, n_children of
// its A slots are children).  Records are loaded for every slot of the block (a masked root has all A slots
// initialised), so that the loads do not wait for n_children.
template <int G, int CH, typename Acc>
__device__ __forceinline__ LevelPick select_level(const Acc& acc, const double* pbc_table, int S, int A, int loc, int N,
                                                  int n_children, int span, const MinMax& mm, double discount,
                                                  bool two_player, uint32_t* mt_key, int32_t& mt_pos, uint32_t& words, int j,
                                                  int group_base, int32_t* error_flag) {
    const auto stats = acc.stats(loc);
    const auto links = acc.links(loc);
    ChildStats st[CH];
    ChildLinks lk[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int child = c * G + j;
        st[c] = ChildStats{0.0, 0.0};
        lk[c] = ChildLinks{0.f, 0, -1, 0};
        if (child < A) {
            st[c] = stats[child];
            lk[c] = links[child];
        }
    }
    return pick_child<G, CH>(st, lk, pbc_table, S, N, n_children, span, mm, discount, two_player, mt_key, mt_pos, words, j,
                             group_base, error_flag);
}

// lanes that can hold a child: pow2 >= min(A, G)
template <int G>
__device__ __forceinline__ int child_span(int A) {
    int span = 1;
    while (span < A && span < G) span <<= 1;
    return span;
}

// The `while node.expanded()` loop (self_play.py:321-335) with select_child (self_play.py:364-379).
template <int G, int CH, typename Acc>
__device__ __forceinline__ Descent descend(const Acc& acc, const double* pbc_table, int S, int A, int sim,
                                           int n_root_children, const MinMax& mm, double discount, bool two_player,
                                           uint32_t* mt_key, int32_t& mt_pos, uint32_t& words, int j, int group_base,
                                           int32_t* path_ties /* this tree's column or null */, int ties_stride,
                                           int32_t* error_flag) {
    const int span = child_span<G>(A);
    int n_children = n_root_children;
    int k = 0;    // expanded-node index of the current parent
    int loc = 0;  // where its block lives
    int N = sim;  // its visit count: the root has been visited once per finished simulation
    int depth = 0;
    int slot = 0;
    for (;;) {
        const LevelPick pick = select_level<G, CH>(acc, pbc_table, S, A, loc, N, n_children, span, mm, discount, two_player,
                                                   mt_key, mt_pos, words, j, group_base, error_flag);
        slot = pick.slot;
        if (j == 0) {
            acc.path_store(depth, (loc << 8) | slot);
            if (path_ties) path_ties[static_cast<size_t>(depth) * ties_stride] = pick.n_ties;
        }
        ++depth;
        if (pick.child < 0) break;  // reached a node that is not expanded yet
        if (depth > sim) {          // cannot happen on a consistent tree (only sim+1 nodes are expanded);
            if (j == 0) atomicOr(error_flag, 2);  // guarantees every wave leaves the loop regardless
            break;
        }
        k = pick.child;
        loc = pick.loc;
        N = pick.visits;
        n_children = A;
    }
    return Descent{depth, k, slot, loc, N};
}

// models.py:656-661: invert the value scaling, fp32, torch's operation order.
__device__ __forceinline__ float inverse_value_transform(float x) {
    const float u = (fabsf(x) + 1.0f) + 0.001f;
    const float w = 0.004f * u;
    const float r = sqrtf(1.0f + w) - 1.0f;
    const float q = r / 0.002f;
    const float y = q * q - 1.0f;
    const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    return sgn * y;
}

// models.py:641-662 support_to_scalar for TWO logit vectors at once (value and reward), fp32, torch's
// operation order per vector; the two reductions are interleaved so their dependent chains overlap.
// The F logits of a vector are spread over the G lanes of the group; with at most 4 per lane the
// exponentials are computed once and kept in registers.  A vector's result does not depend on the other.
template <int G>
__device__ __forceinline__ void support_to_scalar_pair(const float* la, const float* lb, int F, int support, int j,
                                                       float& out_a, float& out_b) {
    const bool small = F <= 4 * G;
    float xa[4], xb[4], ea[4], eb[4];
    float ma = -INFINITY, mb = -INFINITY;
    if (small) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = j + u * G;
            xa[u] = (i < F) ? la[i] : -INFINITY;
            xb[u] = (i < F) ? lb[i] : -INFINITY;
            ma = fmaxf(ma, xa[u]);
            mb = fmaxf(mb, xb[u]);
        }
    } else {
        for (int i = j; i < F; i += G) {
            ma = fmaxf(ma, la[i]);
            mb = fmaxf(mb, lb[i]);
        }
    }
    MZ_BUTTERFLY(G, G, (ma = fmaxf(ma, partner<M>(ma)), mb = fmaxf(mb, partner<M>(mb))));
    float sa = 0.f, sb = 0.f;
    if (small) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = j + u * G;
            ea[u] = (i < F) ? expf(xa[u] - ma) : 0.f;
            eb[u] = (i < F) ? expf(xb[u] - mb) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // same accumulation order as the strided loop: j, j+G, j+2G, ...
            if (j + u * G < F) {
                sa += ea[u];
                sb += eb[u];
            }
        }
    } else {
        for (int i = j; i < F; i += G) {
            sa += expf(la[i] - ma);
            sb += expf(lb[i] - mb);
        }
    }
    MZ_BUTTERFLY(G, G, (sa = sa + partner<M>(sa), sb = sb + partner<M>(sb)));
    const float inva = 1.0f / sa, invb = 1.0f / sb;
    float acca = 0.f, accb = 0.f;
    if (small) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = j + u * G;
            if (i < F) {
                acca += static_cast<float>(i - support) * (ea[u] * inva);
                accb += static_cast<float>(i - support) * (eb[u] * invb);
            }
        }
    } else {
        for (int i = j; i < F; i += G) {
            acca += static_cast<float>(i - support) * (expf(la[i] - ma) * inva);
            accb += static_cast<float>(i - support) * (expf(lb[i] - mb) * invb);
        }
    }
    MZ_BUTTERFLY(G, G, (acca = acca + partner<M>(acca), accb = accb + partner<M>(accb)));
    out_a = inverse_value_transform(acca);
    out_b = inverse_value_transform(accb);
}

template <int G>
__device__ __forceinline__ float support_to_scalar_group(const float* logits, int F, int support, int j) {
    float a, b;
    support_to_scalar_pair<G>(logits, logits, F, support, j, a, b);
    return a;
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

// node.expand over the full action space (self_play.py:346-352, 452-466): children of node k_new.
template <int G, int CH, typename Acc>
__device__ __forceinline__ void write_children(const Acc& acc, int loc_new, int A, const double (&prior)[CH], int j) {
    const auto stats = acc.stats(loc_new);
    const auto links = acc.links(loc_new);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int child = c * G + j;
        if (child < A) {
            stats[child] = ChildStats{0.0, prior[c]};
            links[child] = ChildLinks{0.f, 0, -1, 0};
        }
    }
}

struct StagedNode {
    double value_sum;
    float reward;
    int32_t visits;
};

// one interior node of the search path (self_play.py:412-428); `same` = node.to_play == to_play
__device__ __forceinline__ void backup_step(double& value_sum, int32_t& visits, double r, bool two_player, bool same,
                                            double discount, double& value, MinMax& mm) {
    if (!two_player) {
        value_sum += value;
        visits += 1;
        const double q = value_sum / static_cast<double>(visits);
        const double seen = r + discount * q;
        mm.maximum = fmax(mm.maximum, seen);
        mm.minimum = fmin(mm.minimum, seen);
        value = r + discount * value;
    } else {
        value_sum += same ? value : -value;
        visits += 1;
        const double q = value_sum / static_cast<double>(visits);
        const double seen = r + discount * -q;
        mm.maximum = fmax(mm.maximum, seen);
        mm.minimum = fmin(mm.minimum, seen);
        value = (same ? -r : r) + discount * value;
    }
}

// backpropagate (self_play.py:407-431) along the path recorded by descend(), leaf first; the leaf's
// own record (reward, first visit, link to its new children) is written here too.  mm and
// root_value_sum are the group leader's registers.  `staged`: kStageLevels LDS slots of this tree
// (only used when the tree itself is in HBM: the lanes fetch the path's records with independent
// loads in flight, the leader runs the sequential recursion out of LDS, the lanes write back).
template <int G, typename Acc>
__device__ __forceinline__ void backup(const Acc& acc, int depth, int sim, double value, float reward_f, bool two_player,
                                       double discount, MinMax& mm, double& root_value_sum, double root_reward,
                                       StagedNode* staged, int j, int loc_new) {
    const int k_new = sim + 1;
    const double reward = static_cast<double>(reward_f);
    // leaf (tree depth == depth): first visit, value_sum was 0; it is `to_play`'s own node
    if (j == 0) {
        const int packed = acc.path_load(depth - 1);
        const int slot = packed & 0xff;
        ChildStats* st = acc.stats(packed >> 8) + slot;
        ChildLinks* lk = acc.links(packed >> 8) + slot;
        const double vs = 0.0 + value;
        st->value_sum = vs;
        *lk = ChildLinks{reward_f, 1, k_new, loc_new};
        const double q = vs / 1.0;
        const double seen = two_player ? (reward + discount * -q) : (reward + discount * q);
        mm.maximum = fmax(mm.maximum, seen);
        mm.minimum = fmin(mm.minimum, seen);
        value = (two_player ? -reward : reward) + discount * value;
    }
    if constexpr (Acc::kInLds) {
        // Tree in LDS: one lane per path level.  The lanes fetch their nodes' records together, the leader
        // runs only the value recursion itself (a multiply-add per level on rewards handed over through
        // `staged`), then every lane finishes its own node -- the fp64 division for the node's mean value
        // runs in parallel across the levels -- and the min-max statistics are reduced over the group
        // (max / min are exact under any association).  Same operations on the same operands as the
        // sequential walk, so the results are bit-identical to it.
        constexpr int C = G < kStageLevels ? G : kStageLevels;
        float* r_buf = reinterpret_cast<float*>(staged);        // [C] rewards, leaf side first
        double* v_buf = reinterpret_cast<double*>(staged) + 8;  // [C] value arriving at each node
        for (int hi = depth - 2; hi >= 0; hi -= C) {
            const int count = (hi + 1 < C) ? hi + 1 : C;
            const bool mine = j < count;
            const int level = hi - j;  // this lane's node sits at tree depth level + 1
            ChildStats* st = nullptr;
            ChildLinks* lk = nullptr;
            double vs = 0.0;
            ChildLinks l{0.f, 0, -1, 0};
            if (mine) {
                const int packed = acc.path_load(level);
                const int slot = packed & 0xff;
                st = acc.stats(packed >> 8) + slot;
                lk = acc.links(packed >> 8) + slot;
                vs = st->value_sum;
                l = *lk;
                r_buf[j] = l.reward;
            }
            group_memory_fence();
            if (j == 0) {
                for (int i = 0; i < count; ++i) {
                    v_buf[i] = value;
                    const double r = static_cast<double>(r_buf[i]);
                    const bool same = ((depth - (hi - i + 1)) & 1) == 0;
                    value = (two_player ? (same ? -r : r) : r) + discount * value;
                }
            }
            group_memory_fence();
            double seen_max = -INFINITY, seen_min = INFINITY;
            if (mine) {
                const double v = v_buf[j];
                const double r = static_cast<double>(l.reward);
                const bool same = ((depth - (level + 1)) & 1) == 0;
                vs += (two_player && !same) ? -v : v;
                const int32_t visits = l.visits + 1;
                const double q = vs / static_cast<double>(visits);
                const double seen = r + discount * (two_player ? -q : q);
                st->value_sum = vs;
                lk->visits = visits;
                seen_max = seen;
                seen_min = seen;
            }
            MZ_BUTTERFLY(G, G, (seen_max = fmax(seen_max, partner<M>(seen_max)),
                                seen_min = fmin(seen_min, partner<M>(seen_min))));
            if (j == 0) {
                mm.maximum = fmax(mm.maximum, seen_max);
                mm.minimum = fmin(mm.minimum, seen_min);
            }
            group_memory_fence();
        }
    } else {
        for (int hi = depth - 2; hi >= 0; hi -= kStageLevels) {
            const int count = (hi + 1 < kStageLevels) ? hi + 1 : kStageLevels;  // levels hi, hi-1, ...
            // every lane's path words first, then every record: all of a tree's loads are in flight together (at HBM scale
            // each is a DRAM round trip of microseconds; issued one per loop iteration they would queue up behind each other)
            constexpr int kPer = (kStageLevels + G - 1) / G;
            int packed_u[kPer];
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int i = j + u * G;
                packed_u[u] = (i < count) ? acc.path_load(hi - i) : 0;
            }
            double vs_u[kPer];
            float2 rv_u[kPer];                           // {reward, visits}: the first 8 bytes of the child's ChildLinks
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int i = j + u * G;
                if (i < count) {
                    const int slot = packed_u[u] & 0xff;
                    vs_u[u] = (acc.stats(packed_u[u] >> 8) + slot)->value_sum;
                    rv_u[u] = *reinterpret_cast<const float2*>(acc.links(packed_u[u] >> 8) + slot);
                }
            }
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int i = j + u * G;
                if (i < count) staged[i] = StagedNode{vs_u[u], rv_u[u].x, __builtin_bit_cast(int32_t, rv_u[u].y)};
            }
            group_memory_fence();
            if (j == 0) {
                for (int i = 0; i < count; ++i) {
                    StagedNode n = staged[i];
                    const int level = hi - i;  // node's tree depth is level + 1
                    const bool same = ((depth - (level + 1)) & 1) == 0;
                    backup_step(n.value_sum, n.visits, static_cast<double>(n.reward), two_player, same, discount, value,
                                mm);
                    staged[i] = n;
                }
            }
            group_memory_fence();
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int i = j + u * G;
                if (i < count) {
                    const int slot = packed_u[u] & 0xff;
                    const StagedNode n = staged[i];
                    (acc.stats(packed_u[u] >> 8) + slot)->value_sum = n.value_sum;
                    (acc.links(packed_u[u] >> 8) + slot)->visits = n.visits;
                }
            }
            group_memory_fence();
        }
    }
    // root (tree depth 0)
    if (j == 0) {
        const double n_root = static_cast<double>(sim + 1);
        double seen;
        if (!two_player) {
            root_value_sum += value;
            seen = root_reward + discount * (root_value_sum / n_root);
        } else {
            const bool same = (depth & 1) == 0;
            root_value_sum += same ? value : -value;
            seen = root_reward + discount * -(root_value_sum / n_root);
        }
        mm.maximum = fmax(mm.maximum, seen);
        mm.minimum = fmin(mm.minimum, seen);
    }
}

// root.expand over the legal actions + add_exploration_noise (self_play.py:303-315, 452-477).
// prior[c] arrives soft-maxed (or injected) for child slot c*G+j; noise_row may be null.
template <int G, int CH, typename Acc>
__device__ __forceinline__ void write_root_children(const Acc& acc, int A, int n_children, double (&prior)[CH],
                                                    const double* noise_row, double noise_frac, int j) {
    const auto stats = acc.stats(0);
    const auto links = acc.links(0);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int child = c * G + j;
        if (child < A) {
            double pr = (child < n_children) ? prior[c] : 0.0;
            if (child < n_children && noise_row) {
                // prior * (1 - frac) + n * frac   (self_play.py:477)
                const double keep = pr * (1 - noise_frac);
                const double add = noise_row[child] * noise_frac;
                pr = keep + add;
            }
            stats[child] = ChildStats{0.0, pr};
            links[child] = ChildLinks{0.f, 0, -1, 0};
        }
    }
}

}  // namespace mz
