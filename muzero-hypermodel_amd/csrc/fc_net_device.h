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

struct NeuronRef {
    const float4* row;  // weight row (padded)
    const float4* x;    // input vector (padded)
    float* y;           // where the activation goes
    float bias;
    int chunks;         // in_pad / 4; 0 = no neuron
    bool elu;
};

// Which job / row neuron n of a phase is.  The phase descriptor is a kernel argument (wave-uniform,
// held in SGPRs); the job is picked with per-lane selects, never with a per-lane index into it (that
// would turn every field access into a memory load from the kernarg segment).
__device__ __forceinline__ NeuronRef locate_neuron(const FcPhase& ph, const float* w, float* scratch, int n) {
    const bool live = n < ph.total_out;
    const FcJob j0 = ph.job[0], j1 = ph.job[1], j2 = ph.job[2];
    const bool past0 = ph.n_jobs > 1 && n >= j0.out;
    const bool past1 = ph.n_jobs > 2 && n >= j0.out + j1.out;
    int o = past1 ? n - j0.out - j1.out : (past0 ? n - j0.out : n);
    const int in_pad = past1 ? j2.in_pad : (past0 ? j1.in_pad : j0.in_pad);
    const int w_lds = past1 ? j2.w_lds : (past0 ? j1.w_lds : j0.w_lds);
    const int b_lds = past1 ? j2.b_lds : (past0 ? j1.b_lds : j0.b_lds);
    const int x_off = past1 ? j2.x_off : (past0 ? j1.x_off : j0.x_off);
    const int y_off = past1 ? j2.y_off : (past0 ? j1.y_off : j0.y_off);
    const int elu = past1 ? j2.elu : (past0 ? j1.elu : j0.elu);
    o = live ? o : 0;  // a lane without a neuron still forms valid addresses (row 0) and never stores
    NeuronRef r;
    r.row = reinterpret_cast<const float4*>(w + w_lds + o * in_pad);
    r.x = reinterpret_cast<const float4*>(scratch + x_off);
    r.y = scratch + y_off + o;
    r.bias = w[b_lds + o];
    r.chunks = live ? (in_pad >> 2) : 0;
    r.elu = elu != 0;
    return r;
}

__device__ __forceinline__ float fma4(const float4& wv, const float4& xv, float acc) {
    acc = fmaf(wv.x, xv.x, acc);
    acc = fmaf(wv.y, xv.y, acc);
    acc = fmaf(wv.z, xv.z, acc);
    acc = fmaf(wv.w, xv.w, acc);
    return acc;
}

__device__ __forceinline__ float elu_or_identity(float v, bool elu) {
    const float e = expm1f(fminf(v, 0.f));  // ELU(alpha = 1) on the negative side
    return (elu && !(v > 0.f)) ? e : v;
}

// models.py:626-638, one phase: Linear (+ELU on hidden layers) for every neuron of up to three
// independent layers.  Inputs must be visible to the group on entry; outputs are on exit.
// Loads are unconditional (indices clamped into the row) so that they stay 16-byte LDS reads issued
// back to back; chunks beyond a neuron's width are discarded with a select, not a branch.
template <int G>
__device__ __forceinline__ void run_phase(const FcPhase& ph, const float* w, float* scratch, int j) {
    constexpr int KB = 4;  // chunks (of 4 inputs) whose loads are issued together
    for (int n0 = j; n0 < ph.total_out; n0 += 2 * G) {
        const NeuronRef a = locate_neuron(ph, w, scratch, n0);
        const NeuronRef b = locate_neuron(ph, w, scratch, n0 + G);
        float acc_a = 0.f, acc_b = 0.f;
        const int chunks = a.chunks > b.chunks ? a.chunks : b.chunks;
        const int last_a = a.chunks > 0 ? a.chunks - 1 : 0, last_b = b.chunks > 0 ? b.chunks - 1 : 0;
        for (int c0 = 0; c0 < chunks; c0 += KB) {
            float4 wa[KB], xa[KB], wb[KB], xb[KB];
#pragma unroll
            for (int u = 0; u < KB; ++u) {
                const int ia = (c0 + u < last_a) ? c0 + u : last_a;
                const int ib = (c0 + u < last_b) ? c0 + u : last_b;
                wa[u] = a.row[ia];
                xa[u] = a.x[ia];
                wb[u] = b.row[ib];
                xb[u] = b.x[ib];
            }
#pragma unroll
            for (int u = 0; u < KB; ++u) {
                const float ta = fma4(wa[u], xa[u], acc_a);
                const float tb = fma4(wb[u], xb[u], acc_b);
                acc_a = (c0 + u < a.chunks) ? ta : acc_a;
                acc_b = (c0 + u < b.chunks) ? tb : acc_b;
            }
        }
        const float va = elu_or_identity(acc_a + a.bias, a.elu);
        const float vb = elu_or_identity(acc_b + b.bias, b.elu);
        if (a.chunks) *a.y = va;
        if (b.chunks) *b.y = vb;
    }
    group_memory_fence();
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
__device__ __forceinline__ void fc_initial(const FcNet& net, const float* w, float* scratch, const float* obs, int j) {
    float* x = scratch;
    for (int i = j; i < ((net.obs + 3) & ~3); i += G) x[i] = (i < net.obs) ? obs[i] : 0.f;
    group_memory_fence();
    for (int ph = 0; ph < net.n_init_pre; ++ph) run_phase<G>(net.init_pre[ph], w, scratch, j);
    unit_rescale<G>(scratch + net.off_raw, scratch + net.off_norm, net.enc, j);
    for (int ph = 0; ph < net.n_init_post; ++ph) run_phase<G>(net.init_post[ph], w, scratch, j);
}

// models.py:147-170, 192-195 recurrent_inference.  On exit scratch holds norm (next hidden state),
// reward / value / policy logits.  The reward head reads the UN-normalised next state
// (models.py:157-159), so all three heads can run side by side once the rescale is done.
template <int G>
__device__ __forceinline__ void fc_recurrent(const FcNet& net, const float* w, float* scratch, const float* hidden,
                                             int action, int j) {
    float* x = scratch;
    for (int i = j; i < net.enc; i += G) x[i] = hidden[i];
    // one-hot action, then zeros up to the padded width
    for (int a = j; net.enc + a < ((net.enc + net.A + 3) & ~3); a += G) x[net.enc + a] = (a == action) ? 1.f : 0.f;
    group_memory_fence();
    for (int ph = 0; ph < net.n_rec_pre; ++ph) run_phase<G>(net.rec_pre[ph], w, scratch, j);
    unit_rescale<G>(scratch + net.off_raw, scratch + net.off_norm, net.enc, j);
    for (int ph = 0; ph < net.n_rec_post; ++ph) run_phase<G>(net.rec_post[ph], w, scratch, j);
}

}  // namespace mz
