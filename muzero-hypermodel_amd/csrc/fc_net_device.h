// fc_net_device.h -- MuZeroFullyConnectedNetwork's inference half (reference models.py:128-195) as
// device functions for one lane group, so a whole simulation (select -> dynamics -> prediction ->
// expand/backup) can run inside one kernel without leaving the CU.
//
// The network is evaluated as a short list of PHASES.  A phase is a set of up to three Linear layers
// that do not depend on each other (e.g. the first layers of the reward, policy and value heads);
// all their output neurons are dealt round-robin to the G lanes of the group.  Each neuron is one
// sequential fp32 FMA chain over its inputs, so its value does not depend on G or on how phases are
// formed.  A lane works on two neurons at a time and issues all their 16-byte LDS reads (inputs and
// weight rows, 4 chunks = 16 inputs per block) before the first FMA: with one wavefront per SIMD there
// is nobody else to hide an LDS round trip, so the number of dependent round trips per inference --
// not the FLOP count -- is what this layout minimises.  Weights are staged once per workgroup into LDS
// with every row padded to 16 bytes; layer inputs / outputs live in a per-tree LDS scratch.
#pragma once
#include <hip/hip_runtime.h>

#include "tree_device.h"

namespace mz {

constexpr int kFcMaxLayers = 4;   // Linear layers per MLP (hidden layers + 1)
constexpr int kFcMaxWidth = 256;  // widest layer input / output

struct FcLayer {
    int32_t in, out;
    int32_t w_off, b_off;    // offsets into the flat fp32 weight buffer (state_dict order, rows of `in` floats)
    int32_t in_pad;          // in rounded up to a multiple of 4
    int32_t w_lds, b_lds;    // offsets into the on-chip copy: rows padded to in_pad floats (16-byte aligned)
};
struct FcMlp {
    int32_t n_layers;
    FcLayer layer[kFcMaxLayers];
};
struct FcJob {
    int32_t in_pad, out, w_lds, b_lds;
    int32_t x_off, y_off;    // scratch offsets (floats) of the input / output vectors
    int32_t elu;             // hidden layer: ELU; last layer of an MLP: identity
};
struct FcPhase {
    int32_t n_jobs, total_out;
    FcJob job[3];
};
// Scratch per tree (floats, every region 16-byte aligned, pads kept at zero):
//   [x_in | raw | norm | reward | value | policy | temps of head 0,1,2 (two each)]
struct FcNet {
    int32_t obs, enc, A, F, support;
    int32_t n_weights;       // flat buffer
    int32_t n_weights_lds;   // padded on-chip copy
    int32_t off_raw, off_norm, off_reward, off_value, off_policy, scratch_floats;
    // initial_inference: representation layers, rescale, then policy/value heads
    int32_t n_init_pre, n_init_post;
    FcPhase init_pre[kFcMaxLayers], init_post[kFcMaxLayers];
    // recurrent_inference: dynamics layers, rescale, then reward/policy/value heads
    int32_t n_rec_pre, n_rec_post;
    FcPhase rec_pre[kFcMaxLayers], rec_post[kFcMaxLayers];
    FcMlp repr, dyn, reward, policy, value;  // for weight staging
};

// LDS plan of the fused whole-move kernel (bytes from the start of dynamic LDS).
struct FusedLayout {
    uint32_t off_weights;
    uint32_t off_table;    // NeuronDesc table
    uint32_t off_rec_table;
    uint32_t off_trees;
    uint32_t tree_bytes;   // per-tree region
    uint32_t off_path;     // within a tree region
    uint32_t off_scratch;
    uint32_t off_hidden;   // 0xffffffff = hidden states are read back from the HBM pool
    uint32_t total_bytes;
};

// Copy one MLP's weights from the flat buffer into the padded on-chip layout (all threads of the
// workgroup; caller synchronises afterwards).
__device__ __forceinline__ void stage_mlp_weights(const FcMlp& m, const float* __restrict__ flat, float* w_lds,
                                                  int tid, int nthreads) {
    for (int l = 0; l < m.n_layers; ++l) {
        const FcLayer L = m.layer[l];
        const int total = L.out * L.in_pad;
        for (int t = tid; t < total; t += nthreads) {
            const int o = t / L.in_pad, i = t - o * L.in_pad;
            w_lds[L.w_lds + t] = (i < L.in) ? flat[L.w_off + o * L.in + i] : 0.f;
        }
        for (int o = tid; o < L.out; o += nthreads) w_lds[L.b_lds + o] = flat[L.b_off + o];
    }
}

__device__ __forceinline__ void stage_fc_weights(const FcNet& net, const float* __restrict__ flat, float* w_lds, int tid,
                                                 int nthreads) {
    stage_mlp_weights(net.repr, flat, w_lds, tid, nthreads);
    stage_mlp_weights(net.dyn, flat, w_lds, tid, nthreads);
    stage_mlp_weights(net.reward, flat, w_lds, tid, nthreads);
    stage_mlp_weights(net.policy, flat, w_lds, tid, nthreads);
    stage_mlp_weights(net.value, flat, w_lds, tid, nthreads);
}

// ---- per-lane neuron descriptors ------------------------------------------------------------------
// Which neurons a lane evaluates never changes during a kernel, so the (phase, neuron) -> (weight row,
// input vector, output slot, bias, width) mapping is resolved ONCE per workgroup into an LDS table;
// inside the simulation loop a lane fetches the four descriptors of a pass with four 16-byte reads
// instead of re-deriving them from the kernel arguments (scalar loads + ~100 VALU ops per phase on the
// dependent chain).  Entry n of a phase's table describes neuron n; a pass covers neurons
// [pass*4G, (pass+1)*4G), lane j taking n = pass*4G + u*G + j for u = 0..3.
struct alignas(16) NeuronDesc {
    uint32_t row_bytes;  // weight row, byte offset from the on-chip weight copy
    uint32_t xy;         // input offset | output offset << 16 (floats from the tree's scratch)
    uint32_t meta;       // chunks (in_pad / 4) | elu << 8;  chunks == 0: no neuron
    float bias;
};

__device__ __forceinline__ int phase_passes(const FcPhase& ph, int G) { return (ph.total_out + 4 * G - 1) / (4 * G); }

__device__ __forceinline__ int phase_list_entries(const FcPhase* list, int n, int G) {
    int total = 0;
    for (int p = 0; p < n; ++p) total += phase_passes(list[p], G) * 4 * G;
    return total;
}

// entries of all four phase lists, in the order init_pre, init_post, rec_pre, rec_post
__device__ __forceinline__ int fc_table_entries(const FcNet& net, int G) {
    return phase_list_entries(net.init_pre, net.n_init_pre, G) + phase_list_entries(net.init_post, net.n_init_post, G) +
           phase_list_entries(net.rec_pre, net.n_rec_pre, G) + phase_list_entries(net.rec_post, net.n_rec_post, G);
}

__device__ __forceinline__ NeuronDesc describe_neuron(const FcPhase& ph, const float* w, int n) {
    NeuronDesc d{0u, 0u, 0u, 0.f};
    if (n >= ph.total_out) return d;
    int jb = 0, o = n;
    if (ph.n_jobs > 1 && o >= ph.job[0].out) {
        o -= ph.job[0].out;
        jb = 1;
        if (ph.n_jobs > 2 && o >= ph.job[1].out) {
            o -= ph.job[1].out;
            jb = 2;
        }
    }
    const FcJob J = ph.job[jb];
    d.row_bytes = static_cast<uint32_t>(J.w_lds + o * J.in_pad) * 4u;
    d.xy = static_cast<uint32_t>(J.x_off) | (static_cast<uint32_t>(J.y_off + o) << 16);
    d.meta = static_cast<uint32_t>(J.in_pad >> 2) | (J.elu ? 0x100u : 0u);
    d.bias = w[J.b_lds + o];
    return d;
}

// Build the table of one phase list (all threads of the workgroup); returns entries written.
__device__ __forceinline__ int build_phase_tables(const FcPhase* list, int n, const float* w, NeuronDesc* table, int G,
                                                  int tid, int nthreads) {
    int base = 0;
    for (int p = 0; p < n; ++p) {
        const int entries = phase_passes(list[p], G) * 4 * G;
        for (int t = tid; t < entries; t += nthreads) table[base + t] = describe_neuron(list[p], w, t);
        base += entries;
    }
    return base;
}

__device__ __forceinline__ void build_fc_tables(const FcNet& net, const float* w, NeuronDesc* table, int G, int tid,
                                                int nthreads) {
    int base = build_phase_tables(net.init_pre, net.n_init_pre, w, table, G, tid, nthreads);
    base += build_phase_tables(net.init_post, net.n_init_post, w, table + base, G, tid, nthreads);
    base += build_phase_tables(net.rec_pre, net.n_rec_pre, w, table + base, G, tid, nthreads);
    build_phase_tables(net.rec_post, net.n_rec_post, w, table + base, G, tid, nthreads);
}

__device__ __forceinline__ float fma4(const float4& wv, const float4& xv, float acc) {
    acc = fmaf(wv.x, xv.x, acc);
    acc = fmaf(wv.y, xv.y, acc);
    acc = fmaf(wv.z, xv.z, acc);
    acc = fmaf(wv.w, xv.w, acc);
    return acc;
}

// ELU(alpha = 1) on the negative side: exp(v) - 1 (absolute error <= 1 ulp of 1, i.e. 6e-8, against the
// reference's expm1 -- three orders below the 1e-5 bar -- at a third of expm1f's instruction count).
__device__ __forceinline__ float elu_negative(float v) { return expf(fminf(v, 0.f)) - 1.0f; }

// One pass of a phase for a lane: NS neurons (n = base + u*G + j, u < NS).  Their descriptors arrive with
// NS 16-byte reads, then the loads of KB 4-input chunks per neuron are issued back to back before the FMAs.
// Loads are unconditional (indices clamped into the row); chunks beyond a neuron's width are dropped with
// a select.  Each neuron is one sequential FMA chain over its inputs.
template <int G, int NS>
__device__ __forceinline__ void run_pass(const NeuronDesc* table, const float* w, float* scratch, int j) {
    constexpr int KB = NS <= 2 ? 4 : 2;  // chunks per neuron whose loads are in flight together
    NeuronDesc d[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) d[u] = table[u * G + j];
    const float4* row[NS];
    const float4* x[NS];
    int chunks[NS], last[NS];
    float acc[NS];
    int max_chunks = 0;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        row[u] = reinterpret_cast<const float4*>(reinterpret_cast<const uint8_t*>(w) + d[u].row_bytes);
        x[u] = reinterpret_cast<const float4*>(scratch + (d[u].xy & 0xffffu));
        chunks[u] = static_cast<int>(d[u].meta & 0xffu);
        last[u] = chunks[u] > 0 ? chunks[u] - 1 : 0;
        max_chunks = chunks[u] > max_chunks ? chunks[u] : max_chunks;
        acc[u] = 0.f;
    }
    for (int c0 = 0; c0 < max_chunks; c0 += KB) {
        float4 wv[NS][KB], xv[NS][KB];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
#pragma unroll
            for (int t = 0; t < KB; ++t) {
                const int i = (c0 + t < last[u]) ? c0 + t : last[u];
                wv[u][t] = row[u][i];
                xv[u][t] = x[u][i];
            }
        }
#pragma unroll
        for (int t = 0; t < KB; ++t) {
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const float next = fma4(wv[u][t], xv[u][t], acc[u]);
                acc[u] = (c0 + t < chunks[u]) ? next : acc[u];
            }
        }
    }
    float v[NS];
    bool any_elu = false;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        v[u] = acc[u] + d[u].bias;
        any_elu = any_elu || ((d[u].meta & 0x100u) != 0u && !(v[u] > 0.f));
    }
    if (__any(any_elu)) {  // hidden layers only: output layers never pay for the exponentials
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const float e = elu_negative(v[u]);
            v[u] = ((d[u].meta & 0x100u) != 0u && !(v[u] > 0.f)) ? e : v[u];
        }
    }
#pragma unroll
    for (int u = 0; u < NS; ++u)
        if (chunks[u]) scratch[d[u].xy >> 16] = v[u];
}

// models.py:626-638, one phase: Linear (+ELU on hidden layers) for every neuron of up to three
// independent layers.  Inputs must be visible to the group on entry; outputs are on exit.
// A pass covers 4G neurons of the phase's table; the last pass only runs as many neuron slots per lane
// as the phase still has (a wave-uniform count), so a 16-neuron layer on a 16-lane group costs one slot.
template <int G>
__device__ __forceinline__ void run_phase(const FcPhase& ph, const NeuronDesc* table, const float* w, float* scratch,
                                          int j) {
    const int passes = phase_passes(ph, G);
    for (int pass = 0; pass < passes; ++pass) {
        const NeuronDesc* t = table + pass * 4 * G;
        const int remaining = ph.total_out - pass * 4 * G;
        const int slots = remaining >= 4 * G ? 4 : (remaining + G - 1) / G;
        switch (slots) {
            case 1: run_pass<G, 1>(t, w, scratch, j); break;
            case 2: run_pass<G, 2>(t, w, scratch, j); break;
            case 3: run_pass<G, 3>(t, w, scratch, j); break;
            default: run_pass<G, 4>(t, w, scratch, j); break;
        }
    }
    group_memory_fence();
}

// run a list of phases whose tables start at `table`; returns the table position after the list
template <int G>
__device__ __forceinline__ const NeuronDesc* run_phase_list(const FcPhase* list, int n, const NeuronDesc* table,
                                                            const float* w, float* scratch, int j) {
    for (int p = 0; p < n; ++p) {
        run_phase<G>(list[p], table, w, scratch, j);
        table += phase_passes(list[p], G) * 4 * G;
    }
    return table;
}

// Row-wise min-max rescale to [0,1] (models.py:137-145, 161-168): raw -> norm.
template <int G>
__device__ __forceinline__ void unit_rescale(const float* raw, float* norm, int n, int j) {
    float mn = INFINITY, mx = -INFINITY;
    for (int i = j; i < n; i += G) {
        mn = fminf(mn, raw[i]);
        mx = fmaxf(mx, raw[i]);
    }
    mn = group_minf<G>(mn);
    mx = group_maxf<G>(mx);
    float scale = mx - mn;
    if (scale < 1e-5f) scale += 1e-5f;
    for (int i = j; i < n; i += G) norm[i] = (raw[i] - mn) / scale;
    group_memory_fence();
}

// Zero the whole scratch once (pads must read as zero; regions only ever receive finite activations).
template <int G>
__device__ __forceinline__ void fc_clear_scratch(const FcNet& net, float* scratch, int j) {
    for (int i = j; i < net.scratch_floats; i += G) scratch[i] = 0.f;
    group_memory_fence();
}

// models.py:172-190 initial_inference (the reward head is the constant log(one_hot): decodes to 0).
// On exit scratch holds norm (the root hidden state), value logits and policy logits.
template <int G>
__device__ __forceinline__ void fc_initial(const FcNet& net, const NeuronDesc* table, const float* w, float* scratch,
                                           const float* obs, int j) {
    float* x = scratch;
    for (int i = j; i < ((net.obs + 3) & ~3); i += G) x[i] = (i < net.obs) ? obs[i] : 0.f;
    group_memory_fence();
    table = run_phase_list<G>(net.init_pre, net.n_init_pre, table, w, scratch, j);
    unit_rescale<G>(scratch + net.off_raw, scratch + net.off_norm, net.enc, j);
    run_phase_list<G>(net.init_post, net.n_init_post, table, w, scratch, j);
}

// models.py:147-170, 192-195 recurrent_inference.  On exit scratch holds norm (next hidden state),
// reward / value / policy logits.  The reward head reads the UN-normalised next state
// (models.py:157-159), so all three heads can run side by side once the rescale is done.
template <int G>
__device__ __forceinline__ void fc_recurrent(const FcNet& net, const NeuronDesc* rec_table, const float* w, float* scratch,
                                             const float* hidden, int action, int j) {
    float* x = scratch;
    for (int i = j; i < net.enc; i += G) x[i] = hidden[i];
    // one-hot action, then zeros up to the padded width
    for (int a = j; net.enc + a < ((net.enc + net.A + 3) & ~3); a += G) x[net.enc + a] = (a == action) ? 1.f : 0.f;
    group_memory_fence();
    rec_table = run_phase_list<G>(net.rec_pre, net.n_rec_pre, rec_table, w, scratch, j);
    unit_rescale<G>(scratch + net.off_raw, scratch + net.off_norm, net.enc, j);
    run_phase_list<G>(net.rec_post, net.n_rec_post, rec_table, w, scratch, j);
}

}  // namespace mz
