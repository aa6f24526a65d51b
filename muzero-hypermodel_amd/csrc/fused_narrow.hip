// fused_narrow.hip -- whole-move search kernel for "narrow" fully-connected networks (every layer fits a
// 16-lane DPP row: CartPole's 8/16-wide MLPs) and the matching lock-step inference kernels.
//
// One tree per DPP row, four trees per wavefront, `waves` wavefronts per workgroup sharing the staged
// weights; trees, paths, hidden states and the (N, n) exploration table live in LDS for the whole move,
// activations live in registers.  See narrow_device.h for the device functions and the arithmetic contract,
// mcts_kernels.hip for the generic fused kernel this one specialises.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>

#include "fc_net_device.h"
#include "kernel_common.h"
#include "narrow_device.h"
#include "np_legacy_rng.h"
#include "tree_device.h"
#include "tree_layout.h"

namespace mz {

constexpr int kNarrowMaxThreads = 256;

// lanes of the row that can hold a child (the kernel's SPAN template argument): 2 (exactly two actions and block
// indices that fit a byte: the windowed descent), 4, 8 or 16
static inline int narrow_span(int A, int S) {
    if (A == 2 && S <= 254) return 2;
    int span = 4;
    while (span < A) span <<= 1;
    return span;
}

// Does this (engine, network) pair qualify for the narrow path?
bool narrow_supported(const TreeParams& p, const FcNet& net) {
    auto one_hidden = [](const FcMlp& m) { return m.n_layers == 2 && m.layer[0].out <= kRow; };
    if (p.group != kRow || p.A > kRow || p.chunks != 1) return false;
    if (net.enc + net.A > kRow || net.obs > kRow || net.F > 2 * kRow) return false;
    if (!(net.repr.n_layers == 1 || one_hidden(net.repr))) return false;
    if (!one_hidden(net.dyn) || !one_hidden(net.reward) || !one_hidden(net.value) || !one_hidden(net.policy)) return false;
    if (p.S < 1 || p.S >= 32768) return false;
    return true;
}

// LDS plan; prefers 4 wavefronts (16 trees) per workgroup, i.e. one workgroup per CU at 4096 trees, and the
// (N, n) table when it fits.  First pass: plans that let a CU keep four wavefronts of the kernel (one per
// SIMD); second pass: anything that fits a workgroup's 160 KB.  Returns false when nothing does.
bool plan_narrow_layout(const TreeParams& p, const FcNet& net, size_t lds_limit, NarrowLayout* out) {
    auto align16 = [](size_t v) { return (v + 15) / 16 * 16; };
    auto align64 = [](size_t v) { return (v + 63) / 64 * 64; };
    int rows = 4;
    if (const char* env = std::getenv("MZMCTS_NARROW_ROWS")) rows = std::max(1, std::min(4, std::atoi(env)));
    int max_mode = 2;
    if (const char* env = std::getenv("MZMCTS_NARROW_PBC2")) max_mode = std::max(0, std::min(2, std::atoi(env)));
    const size_t span = static_cast<size_t>(narrow_span(p.A, p.S));
    for (int pass = 0; pass < 2; ++pass)
    for (int mode = max_mode; mode >= 0; --mode) {
        if (mode == 2 && p.S > 63) continue;   // rows of 64 entries: n <= N <= S
        for (int waves = 4; waves >= 1; waves >>= 1) {
            NarrowLayout lay{};
            size_t off = 0;
            lay.off_pbc = 0;
            lay.pbc2_mode = mode;
            if (const char* env = std::getenv("MZMCTS_NARROW_EXACT_DIV")) lay.exact_division = std::atoi(env) != 0;
            off = align16(sizeof(double) * 2 * (static_cast<size_t>(p.S) + 1));
            if (mode == 2) {
                lay.off_pbc2 = static_cast<uint32_t>(off);
                off = align16(off + sizeof(double) * (static_cast<size_t>(p.S) + 1) * 64);
            } else if (mode == 1) {
                lay.off_pbc2 = static_cast<uint32_t>(off);
                off = align16(off + sizeof(double) * (static_cast<size_t>(p.S) + 1) * (static_cast<size_t>(p.S) + 2) / 2);
            } else {
                lay.off_pbc2 = 0xffffffffu;
            }
            lay.off_units = static_cast<uint32_t>(off);
            off += sizeof(float) * 4 * 4 * kRow * kNarrowUnits;
            lay.off_bias = static_cast<uint32_t>(off);
            off = align64(off + sizeof(float) * kRow * kNarrowUnits);
            lay.off_trees = static_cast<uint32_t>(off);   // 64-byte aligned (descend_pair flips address bits)
            size_t t = static_cast<size_t>(p.S + 1) * 32 * span;   // LdsTreeV<SPAN>::kBlockStride
            lay.off_side = static_cast<uint32_t>(t);
            t = align16(t + sizeof(SideStats) * static_cast<size_t>(p.S + 1) * span);
            lay.off_path = static_cast<uint32_t>(t);
            t = align16(t + sizeof(int32_t) * static_cast<size_t>(p.S));
            lay.off_hidden = static_cast<uint32_t>(t);
            t = align16(t + sizeof(float) * static_cast<size_t>(p.S + 1) * net.enc);
            lay.off_misc = static_cast<uint32_t>(t);
            t = align16(t + 4 * 2 * kRow);
            lay.off_desc = static_cast<uint32_t>(t);
            if (span == 2) t += 16 * static_cast<size_t>(p.S + 1);
            t = align64(t);
            lay.tree_bytes = static_cast<uint32_t>(t);
            lay.waves = waves;
            lay.rows = rows;
            const size_t total = off + static_cast<size_t>(waves) * rows * t;
            lay.total_bytes = static_cast<uint32_t>(total);
            // pass 0: a CU keeps 16 trees resident: 16 / (waves * rows) workgroups share its LDS
            if (total <= (pass == 0 ? lds_limit * waves * rows / 16 : lds_limit)) {
                *out = lay;
                return true;
            }
        }
    }
    return false;
}

__device__ __forceinline__ void stage_narrow_tables(const TreeParams& p, const NarrowLayout& lay, uint8_t* smem, int tid,
                                                    int nthreads) {
    double* pbc = reinterpret_cast<double*>(smem + lay.off_pbc);
    for (int i = tid; i <= p.S; i += nthreads) {
        pbc[i] = p.pbc_log[i];
        pbc[p.S + 1 + i] = p.pbc_sqrt[i];
    }
    if (lay.pbc2_mode == 2) {
        // rows of 64: entry (N << 6) + n, n <= N -- the triangular table's values, addressed by shifts
        double* pbc2 = reinterpret_cast<double*>(smem + lay.off_pbc2);
        for (int t = tid; t < (p.S + 1) * 64; t += nthreads) {
            const int N = t >> 6, n = t & 63;
            if (n <= N) pbc2[t] = p.pbc_log[N] * (p.pbc_sqrt[N] / static_cast<double>(n + 1));
        }
    } else if (lay.pbc2_mode == 1) {
        // [N][n], n <= N: (log(..)+init) * (sqrt(N) / (n+1)) -- ucb_score's two operations (self_play.py:385-390)
        double* pbc2 = reinterpret_cast<double*>(smem + lay.off_pbc2);
        const int total = (p.S + 1) * (p.S + 2) / 2;
        int N = 0, row = 0;  // row = N(N+1)/2
        for (int t = tid; t < total; t += nthreads) {
            while (row + N + 1 <= t) {
                row += N + 1;
                ++N;
            }
            const int n = t - row;
            pbc2[t] = p.pbc_log[N] * (p.pbc_sqrt[N] / static_cast<double>(n + 1));
        }
    }
}

// -------------------------------------------------------------------------------------------------
// one launch per move: root inference + expansion + noise, S simulations, publish
// -------------------------------------------------------------------------------------------------
template <int SPAN, int PBC2>
__global__ __launch_bounds__(kNarrowMaxThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void search_fused_narrow_kernel(
    TreeParams p, FcNet net, NarrowLayout lay, const float* __restrict__ weights,
    const float* __restrict__ observations,  // [E][obs]
    MoveCtl ctl, int n_sims, int publish_tree) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    MZ_STAMP_DECL
    const double* pbc = reinterpret_cast<const double*>(smem + lay.off_pbc);
    const double* pbc2 = reinterpret_cast<const double*>(smem + (PBC2 ? lay.off_pbc2 : 0u));
    float4* units = reinterpret_cast<float4*>(smem + lay.off_units);
    float* bias = reinterpret_cast<float*>(smem + lay.off_bias);
    stage_narrow_tables(p, lay, smem, threadIdx.x, blockDim.x);
    stage_narrow_units(net, weights, units, bias, threadIdx.x, blockDim.x);
    __syncthreads();
    MZ_STAMP(0);

    const int j = threadIdx.x % kRow;
    const int group_base = (threadIdx.x & 63) - j;  // lane of the row's first lane inside its wavefront
    const int row = group_base / kRow;
    if (row >= lay.rows) return;                    // this wavefront carries fewer than four trees
    const int tree_in_block = (threadIdx.x / 64) * lay.rows + row;
    const int e = blockIdx.x * (blockDim.x / 64) * lay.rows + tree_in_block;
    if (e >= p.E) return;
    const int n_root = p.root_children[e];
    if (move_stalled(p, ctl, e, j)) return;
    const double* noise = ctl.noise;
    if (j == 0) reset_search_state(p, e, ctl.rng_skip);
    if (n_root == 0) {
        if (j == 0 && ctl.actions) ctl.actions[e] = -1;
        return;
    }
    // the DPP lane mappings the code below relies on (a wrong assumption must not pass silently)
    if (row_shl1_bits(j) != ((j + 1 < kRow) ? j + 1 : 0) || row_ror_bits<1>(j) != ((j + kRow - 1) & (kRow - 1))) {
        if (j == 0) atomicOr(p.error_flag, 8);
        return;
    }

    uint8_t* region = smem + lay.off_trees + static_cast<size_t>(tree_in_block) * lay.tree_bytes;
    const LdsTreeV<SPAN> tree{region, reinterpret_cast<SideStats*>(region + lay.off_side),
                              reinterpret_cast<int32_t*>(region + lay.off_path)};
    float* hidden_lds = reinterpret_cast<float*>(region + lay.off_hidden);
    int32_t* root_action_lds = reinterpret_cast<int32_t*>(region + lay.off_misc);
    float* root_logit_lds = reinterpret_cast<float*>(region + lay.off_misc) + kRow;
    const int enc = net.enc;
    const bool two_player = p.P == 2;
    const bool wide_support = net.F > kRow;

    // ---- root: initial inference, root.expand over the legal actions, exploration noise ----------
    float predicted;
    {
        const float obs = (j < net.obs) ? observations[static_cast<size_t>(e) * net.obs + j] : 0.f;
        const NarrowHeads h = narrow_initial(units, bias, enc, wide_support, net.repr.n_layers == 2, obs, j);
        float unused;
        narrow_support_pair(h.value_a, h.value_b, h.value_a, h.value_b, net.F, net.support, j, predicted, unused);
        // child slot c of the root is action root_action[c]: fetch its logit across the lanes through LDS
        const int my_action = (j < n_root) ? p.root_action[static_cast<size_t>(e) * p.A + j] : 0;
        root_action_lds[j] = my_action;
        root_logit_lds[j] = h.policy;
        if (j < enc) hidden_lds[j] = h.norm;  // slab 0
        group_memory_fence();
        const bool valid = j < n_root;
        double prior[1] = {narrow_softmax(valid ? root_logit_lds[my_action] : 0.f, valid)};
        write_root_children<kRow, 1>(tree, p.A, n_root, prior, noise ? noise + static_cast<size_t>(e) * p.A : nullptr,
                                     p.noise_frac, j);
        if constexpr (SPAN == 2) {  // the prior moves aside; the block holds table[0][0] * prior (see LdsTreeV)
            if (j == 0) *reinterpret_cast<uint4*>(region + lay.off_desc) = uint4{0u, 0u, 0u, 0u};   // (the root, at position 1 of its own table)
            if (j < p.A) {
                const double pr = tree.stats(0)[j].prior;
                tree.side(0)[j].prior = pr;
                tree.stats(0)[j].prior = exploration_factor<PBC2>(pbc, pbc2, p.S, 0, 0) * pr;
            }
        }
        group_memory_fence();
    }
    // prior_score factor of the children of a node expanded by this search: it has one visit when they are first scored
    const double leaf_factor = exploration_factor<PBC2>(pbc, pbc2, p.S, 1, 0);

    MinMax mm{INFINITY, -INFINITY};  // replicated in every lane of the row
    double root_value_sum = 0.0;     // lane 0
    const double root_reward = 0.0;  // log(one_hot(centre)) decodes to exactly 0
    int32_t mt_pos = (j == 0) ? p.mt_pos[e] : 0;
    uint32_t words = 0;
    int max_depth = 0;
    int64_t depth_sum = 0;
    unsigned long long exotic = lay.exact_division ? 1ull : 0ull;   // (per wavefront) a backed-up value left normalized_value's plain range
    uint32_t* mt_key = p.mt_key + static_cast<size_t>(e) * kMtN;
    MZ_STAMP(1);

    const ResidentWeights resident = load_resident_weights(units, bias, j);
    const float transform_reciprocal = inverse_transform_reciprocal();
    int root_entry = 0;
    if constexpr (SPAN == 2) root_entry = window_root_entry(region + lay.off_desc, j);
    // ---- S simulations, entirely inside the CU -----------------------------------------------------
    for (int sim = 0; sim < n_sims; ++sim) {
        Descent d;
        float state;
        if constexpr (SPAN == 2)
            d = descend_window(tree, region + lay.off_desc, root_entry, sim, n_root, mm, exotic, mt_key, mt_pos, words, j, group_base,
                               p.error_flag);
        else
            d = descend_row<SPAN, PBC2>(tree, pbc, pbc2, p.S, p.A, sim, n_root, mm, exotic, mt_key, mt_pos, words, j,
                                        group_base, p.error_flag MZ_DSTAMP_ARGS);
        state = hidden_lds[d.parent * enc + (j < enc ? j : 0)];
        // the backup's operands (leaf-side round) and the path entries the new node is linked under: asked for here and
        // between the layers of the network, needed after it
        BackupRound<SPAN> round = backup_fetch_path(tree, d.depth, ((d.depth - 1) >> 4) << 4, j);
        int link_entry = 0;
        if constexpr (SPAN == 2) link_entry = link_fetch_path(tree, d.depth, j);
        MZ_STAMP(2);
        const int action = (d.depth == 1) ? root_action_lds[d.slot] : d.slot;
        const float x0 = (j < enc) ? state : ((j - enc == action) ? 1.f : 0.f);
        const NarrowHeads h = narrow_recurrent(
            units, bias, enc, wide_support, x0, j, resident, [&]() { backup_fetch_records(tree, round, j); },
            [&]() { backup_fetch_factors<SPAN, PBC2>(round, pbc, pbc2, p.S, sim, j); });
        MZ_STAMP(3);
        float value_f, reward_f;
        narrow_support_pair<true>(h.value_a, h.value_b, h.reward_a, h.reward_b, net.F, net.support, j, value_f, reward_f,
                                  transform_reciprocal);
        double prior[1] = {narrow_softmax<SPAN, true>(h.policy, j < p.A)};
        MZ_STAMP(4);
        const int k_new = sim + 1;
        if constexpr (SPAN == 2) {
            write_pair_children(tree, k_new, p.A, prior[0], leaf_factor, j);
            link_new_node(region + lay.off_desc, link_entry, d.depth, k_new, j);
            root_entry = window_root_entry(region + lay.off_desc, j);
        } else {
            write_children<kRow, 1>(tree, k_new, p.A, prior, j);
        }
        if (j < enc) hidden_lds[k_new * enc + j] = h.norm;
        MZ_STAMP(5);
        backup_row<SPAN, PBC2>(tree, round, d.depth, sim, static_cast<double>(value_f), reward_f, two_player, p.discount, mm,
                               root_value_sum, root_reward, exotic, pbc, pbc2, p.S, j);
        group_memory_fence();
        if (d.depth > max_depth) max_depth = d.depth;
        depth_sum += d.depth;
        MZ_STAMP(6);
    }

    // ---- publish: per-tree statistics, the root's child block, and on request the whole tree -----------
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
            if (ctl.actions) ctl.actions[e] = slot >= 0 ? root_action_lds[slot] : slot;
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
    if (ctl.visits && j < p.A) ctl.visits[static_cast<size_t>(e) * p.A + j] = (j < n_root) ? tree.links(0)[j].visits : 0;
    // (in LDS a block is two member arrays with the value sums apart, in HBM an array of 32-byte child records: 16-byte
    // word 2 c / 2 c + 1 of a record block are the stats / links of child c; a child never visited has value_sum 0)
    const int block_words = 2 * p.A;
    const int n_blocks = publish_tree ? n_sims + 1 : 1;
    for (int t = j; t < n_blocks * block_words; t += kRow) {
        const int k = t / block_words, i = t - k * block_words;
        uint4* dst = reinterpret_cast<uint4*>(p.blocks + (static_cast<size_t>(k) * p.E + e) * p.line_stride);   // own line, half 0
        if (i < p.A) {
            ChildStats st = tree.stats(k)[i];
            const SideStats sd = tree.side(k)[i];
            st.value_sum = tree.links(k)[i].visits > 0 ? sd.value_sum : 0.0;
            if constexpr (SPAN == 2) st.prior = sd.prior;
            *reinterpret_cast<ChildStats*>(dst + 2 * i) = st;
        } else {
            *reinterpret_cast<ChildLinks*>(dst + 2 * (i - p.A) + 1) = tree.links(k)[i - p.A];
        }
    }
    for (int t = j; t < n_blocks * enc; t += kRow) {
        const int k = t / enc, i = t - k * enc;
        p.hidden[(static_cast<size_t>(k) * p.E + e) * enc + i] = hidden_lds[t];
    }
    MZ_STAMP(7);
    MZ_STAMP_FLUSH;
}

// -------------------------------------------------------------------------------------------------
// lock-step form of the same network functions (one launch per inference over the [E, .] batch), so the
// fused kernel can be compared with the lock-step search bit for bit
// -------------------------------------------------------------------------------------------------
template <bool INITIAL>
__global__ __launch_bounds__(kNarrowMaxThreads) void fc_inference_narrow_kernel(
    FcNet net, const float* __restrict__ weights, int E, const float* __restrict__ in, const int64_t* __restrict__ action,
    float* __restrict__ value_logits, float* __restrict__ reward_logits, float* __restrict__ policy_logits,
    float* __restrict__ hidden_out) {
    __shared__ float4 units[kNarrowUnits * 4 * kRow];
    __shared__ float bias[kNarrowUnits * kRow];
    stage_narrow_units(net, weights, units, bias, threadIdx.x, blockDim.x);
    __syncthreads();
    const int e = blockIdx.x * (blockDim.x / kRow) + threadIdx.x / kRow;
    const int j = threadIdx.x % kRow;
    if (e >= E) return;
    const bool wide_support = net.F > kRow;
    NarrowHeads h;
    if (INITIAL) {
        const float obs = (j < net.obs) ? in[static_cast<size_t>(e) * net.obs + j] : 0.f;
        h = narrow_initial(units, bias, net.enc, wide_support, net.repr.n_layers == 2, obs, j);
    } else {
        const int a = static_cast<int>(action[e]);
        const float state = in[static_cast<size_t>(e) * net.enc + (j < net.enc ? j : 0)];
        const float x0 = (j < net.enc) ? state : ((j - net.enc == a) ? 1.f : 0.f);
        h = narrow_recurrent(units, bias, net.enc, wide_support, x0, j, load_resident_weights(units, bias, j));
    }
    if (j < net.enc) hidden_out[static_cast<size_t>(e) * net.enc + j] = h.norm;
    if (j < net.A) policy_logits[static_cast<size_t>(e) * net.A + j] = h.policy;
    const int centre = net.F / 2;
    if (j < net.F) {
        value_logits[static_cast<size_t>(e) * net.F + j] = h.value_a;
        // initial reward = log(one_hot(centre)): -inf everywhere but 0 at the centre (models.py:176-186)
        reward_logits[static_cast<size_t>(e) * net.F + j] = INITIAL ? (j == centre ? 0.f : -INFINITY) : h.reward_a;
    }
    if (kRow + j < net.F) {
        value_logits[static_cast<size_t>(e) * net.F + kRow + j] = h.value_b;
        reward_logits[static_cast<size_t>(e) * net.F + kRow + j] =
            INITIAL ? (kRow + j == centre ? 0.f : -INFINITY) : h.reward_b;
    }
}

#ifdef MZ_STAMPS
hipError_t read_stamp_sums_narrow(unsigned long long* out, bool reset) {
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
template <int SPAN>
static hipError_t launch_narrow_span(const TreeParams& p, const FcNet& net, const NarrowLayout& lay, const float* weights,
                                     const float* observations, const MoveCtl& ctl, int n_sims, int publish_tree,
                                     hipStream_t stream, const LaunchTiming* timing) {
    const int threads = 64 * lay.waves;
    const int trees = lay.waves * lay.rows;
    const int grid = (p.E + trees - 1) / trees;
    auto go = [&](auto kernel) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lay.total_bytes));
        if (err != hipSuccess) return err;
        launch_kernel(kernel, dim3(grid), dim3(threads), lay.total_bytes, stream, timing, p, net, lay, weights, observations,
                      ctl, n_sims, publish_tree);
        return hipGetLastError();
    };
    if (lay.pbc2_mode == 2) return go(search_fused_narrow_kernel<SPAN, 2>);
    if (lay.pbc2_mode == 1) return go(search_fused_narrow_kernel<SPAN, 1>);
    return go(search_fused_narrow_kernel<SPAN, 0>);
}

hipError_t launch_search_fused_narrow(const TreeParams& p, const FcNet& net, const NarrowLayout& lay, const float* weights,
                                      const float* observations, const MoveCtl& ctl, int n_sims, int publish_tree,
                                      hipStream_t stream, const LaunchTiming* timing) {
    if (narrow_span(p.A, p.S) == 2) return launch_narrow_span<2>(p, net, lay, weights, observations, ctl, n_sims, publish_tree, stream, timing);
    if (p.A <= 4) return launch_narrow_span<4>(p, net, lay, weights, observations, ctl, n_sims, publish_tree, stream, timing);
    if (p.A <= 8) return launch_narrow_span<8>(p, net, lay, weights, observations, ctl, n_sims, publish_tree, stream, timing);
    return launch_narrow_span<16>(p, net, lay, weights, observations, ctl, n_sims, publish_tree, stream, timing);
}

hipError_t launch_fc_inference_narrow(const TreeParams& p, const FcNet& net, const float* weights, bool initial,
                                      const float* in, const int64_t* action, float* value_logits, float* reward_logits,
                                      float* policy_logits, float* hidden_out, hipStream_t stream) {
    const int threads = kNarrowMaxThreads;
    const int trees = threads / kRow;
    const int grid = (p.E + trees - 1) / trees;
    if (initial)
        fc_inference_narrow_kernel<true><<<dim3(grid), dim3(threads), 0, stream>>>(net, weights, p.E, in, action, value_logits,
                                                                                 reward_logits, policy_logits, hidden_out);
    else
        fc_inference_narrow_kernel<false><<<dim3(grid), dim3(threads), 0, stream>>>(net, weights, p.E, in, action, value_logits,
                                                                                  reward_logits, policy_logits, hidden_out);
    return hipGetLastError();
}

}  // namespace mz
