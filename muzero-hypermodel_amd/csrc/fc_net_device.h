// fc_net_device.h -- MuZeroFullyConnectedNetwork's inference half (reference models.py:128-195) as
// device functions for one lane group, so a whole simulation (select -> dynamics -> prediction ->
// expand/backup) can run inside one kernel without leaving the CU.
//
// Work split: output neuron o of a layer belongs to lane o mod G and is accumulated by that lane
// alone, sequentially over the inputs with explicit fp32 FMAs, so a neuron's value does not depend
// on G.  Layer inputs / outputs live in a small per-tree LDS scratch; the weights are staged once per
// workgroup into LDS with every row padded to 16 bytes -- the G lanes of every group read the same
// G rows, so the reads broadcast across the groups of the wave.
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
// scratch layout per tree, in floats: [x_in | t0 | t1 | raw | norm | reward | value | policy]; every
// region starts 16-byte aligned and is padded to a multiple of 4 floats (pads hold zeros).
struct FcNet {
    int32_t obs, enc, A, F, support;
    int32_t n_weights;       // flat buffer
    int32_t n_weights_lds;   // padded on-chip copy
    int32_t off_t0, off_t1, off_raw, off_norm, off_reward, off_value, off_policy, scratch_floats;
    FcMlp repr, dyn, reward, policy, value;
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

// models.py:626-638 mlp(): Linear (+ELU between layers, identity at the end).
// `w` is the padded on-chip weight copy; `x` (padded with zeros to a multiple of 4) must be visible to
// the whole group on entry; `y` is visible to the whole group on exit, its pad zeroed.
// Each output neuron is one sequential fp32 FMA chain over its inputs (so its value is independent of
// G); a lane runs up to four neurons at a time for instruction-level parallelism, with 16-byte LDS
// reads of the shared input vector and of each weight row.
template <int G>
__device__ __forceinline__ void mlp_forward(const FcMlp& m, const float* w, const float* x, float* y, float* t0,
                                            float* t1, int j) {
    const float* cur = x;
    for (int l = 0; l < m.n_layers; ++l) {
        const FcLayer L = m.layer[l];
        const bool last = l == m.n_layers - 1;
        float* dst = last ? y : ((l & 1) ? t1 : t0);
        const int chunks = L.in_pad >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(cur);
        for (int o0 = j; o0 < L.out; o0 += 4 * G) {
            const float4* row[4];
            bool live[4];
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int o = o0 + u * G;
                live[u] = o < L.out;
                row[u] = reinterpret_cast<const float4*>(w + L.w_lds + (live[u] ? o : o0) * L.in_pad);
            }
            for (int c = 0; c < chunks; ++c) {
                const float4 xv = x4[c];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (live[u]) {
                        const float4 wv = row[u][c];
                        acc[u] = fmaf(wv.x, xv.x, acc[u]);
                        acc[u] = fmaf(wv.y, xv.y, acc[u]);
                        acc[u] = fmaf(wv.z, xv.z, acc[u]);
                        acc[u] = fmaf(wv.w, xv.w, acc[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (live[u]) {
                    const int o = o0 + u * G;
                    float v = acc[u] + w[L.b_lds + o];
                    if (!last) v = v > 0.f ? v : expm1f(v);  // ELU(alpha = 1)
                    dst[o] = v;
                }
            }
        }
        for (int o = L.out + j; o < ((L.out + 3) & ~3); o += G) dst[o] = 0.f;  // keep the pad at zero
        group_memory_fence();
        cur = dst;
    }
}

// Row-wise min-max rescale to [0,1] (models.py:137-145, 161-168): raw -> norm (pad zeroed).
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
    for (int i = n + j; i < ((n + 3) & ~3); i += G) norm[i] = 0.f;
    group_memory_fence();
}

// models.py:172-190 initial_inference (the reward head is the constant log(one_hot): decodes to 0).
// On exit scratch holds norm (the root hidden state), value logits and policy logits.
template <int G>
__device__ __forceinline__ void fc_initial(const FcNet& net, const float* w, float* scratch, const float* obs, int j) {
    float* x = scratch;
    for (int i = j; i < ((net.obs + 3) & ~3); i += G) x[i] = (i < net.obs) ? obs[i] : 0.f;
    group_memory_fence();
    mlp_forward<G>(net.repr, w, x, scratch + net.off_raw, scratch + net.off_t0, scratch + net.off_t1, j);
    unit_rescale<G>(scratch + net.off_raw, scratch + net.off_norm, net.enc, j);
    mlp_forward<G>(net.policy, w, scratch + net.off_norm, scratch + net.off_policy, scratch + net.off_t0,
                   scratch + net.off_t1, j);
    mlp_forward<G>(net.value, w, scratch + net.off_norm, scratch + net.off_value, scratch + net.off_t0,
                   scratch + net.off_t1, j);
}

// models.py:147-170, 192-195 recurrent_inference.  On exit scratch holds norm (next hidden state),
// reward / value / policy logits.
template <int G>
__device__ __forceinline__ void fc_recurrent(const FcNet& net, const float* w, float* scratch, const float* hidden,
                                             int action, int j) {
    float* x = scratch;
    for (int i = j; i < net.enc; i += G) x[i] = hidden[i];
    // one-hot action, then zeros up to the padded width
    for (int a = j; net.enc + a < ((net.enc + net.A + 3) & ~3); a += G) x[net.enc + a] = (a == action) ? 1.f : 0.f;
    group_memory_fence();
    mlp_forward<G>(net.dyn, w, x, scratch + net.off_raw, scratch + net.off_t0, scratch + net.off_t1, j);
    // reward head reads the UN-normalised next state (models.py:157-159)
    mlp_forward<G>(net.reward, w, scratch + net.off_raw, scratch + net.off_reward, scratch + net.off_t0,
                   scratch + net.off_t1, j);
    unit_rescale<G>(scratch + net.off_raw, scratch + net.off_norm, net.enc, j);
    mlp_forward<G>(net.policy, w, scratch + net.off_norm, scratch + net.off_policy, scratch + net.off_t0,
                   scratch + net.off_t1, j);
    mlp_forward<G>(net.value, w, scratch + net.off_norm, scratch + net.off_value, scratch + net.off_t0,
                   scratch + net.off_t1, j);
}

}  // namespace mz
