// trainer_kernels.hip -- the loss of one training step over all unrolled positions in one launch (include/mztrain.h).
//
// One 64-lane workgroup per SAMPLE walks its K1 unrolled steps in order, so the sums over the steps are accumulated
// exactly as the reference's Python loop does (trainer.py:176-198).  Per step and head: log-softmax of the logits
// (max, exp, sum, log: fp32), the cross-entropy against the target -- two-hot for value and reward
// (models.scalar_to_support, models.py:665-685), dense for the policy -- and its gradient softmax * sum(target) -
// target, scaled by what flows into that term: PER weight, value-loss weight, 1 / gradient scale for k > 0.
// Launch-latency-sized work (B x K1 x (2 F + A) floats); what it buys is ~25 launches less per unrolled step.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/mzmcts.h"
#include "../../include/mztrain.h"

namespace {

constexpr int kLanes = 64;

__device__ __forceinline__ float wave_max(float v) {
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, kLanes));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m, kLanes);
    return v;
}

// log-softmax statistics of a row: returns max and log(sum(exp(x - max))) to every lane
__device__ __forceinline__ void row_stats(const float* x, int n, int lane, float& mx, float& log_sum) {
    float m = -INFINITY;
    for (int i = lane; i < n; i += kLanes) m = fmaxf(m, x[i]);
    mx = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < n; i += kLanes) s += expf(x[i] - mx);
    log_sum = logf(wave_sum(s));
}

// models.scalar_to_support for one scalar: indices and weights of the (at most) two non-zero entries
__device__ __forceinline__ void two_hot(float x, int support, int& lo, float& w_lo, int& hi, float& w_hi) {
    const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    float t = sgn * (sqrtf(fabsf(x) + 1.f) - 1.f) + 0.001f * x;
    t = fminf(fmaxf(t, static_cast<float>(-support)), static_cast<float>(support));
    const float low = floorf(t);
    const float frac = t - low;
    lo = static_cast<int>(low) + support;
    w_lo = 1.f - frac;
    hi = lo + 1;
    w_hi = frac;
    if (hi > 2 * support) {  // the reference scatters 0.0 into entry 0 then: no second entry
        hi = -1;
        w_hi = 0.f;
    }
}

// cross-entropy of a row against a two-hot target + gradient row (scaled); result in every lane
__device__ __forceinline__ float ce_two_hot(const float* x, float* g, int n, float target, int support, float scale,
                                            int lane) {
    float mx, log_sum;
    row_stats(x, n, lane, mx, log_sum);
    int lo, hi;
    float w_lo, w_hi;
    two_hot(target, support, lo, w_lo, hi, w_hi);
    const float total = (hi >= 0) ? (w_lo + w_hi) : w_lo;   // sum of the target row, in index order
    for (int i = lane; i < n; i += kLanes) {
        const float t = (i == lo) ? w_lo : ((i == hi) ? w_hi : 0.f);
        const float soft = expf((x[i] - mx) - log_sum);
        g[i] = scale * (soft * total - t);
    }
    float loss = -w_lo * ((x[lo] - mx) - log_sum);
    if (hi >= 0) loss = loss + -w_hi * ((x[hi] - mx) - log_sum);
    return loss;
}

__device__ __forceinline__ float ce_dense(const float* x, const float* target, float* g, int n, float scale, int lane) {
    float mx, log_sum;
    row_stats(x, n, lane, mx, log_sum);
    float total = 0.f;
    for (int i = 0; i < n; ++i) total += target[i];          // (n <= a few dozen actions: every lane, index order)
    float loss = 0.f;
    for (int i = 0; i < n; ++i) loss += -target[i] * ((x[i] - mx) - log_sum);
    for (int i = lane; i < n; i += kLanes) g[i] = scale * (expf((x[i] - mx) - log_sum) * total - target[i]);
    return loss;
}

// models.support_to_scalar of one row (fp32, softmax then expectation then the inverse value transform)
__device__ __forceinline__ float decode_row(const float* x, int n, int support, int lane) {
    float mx, log_sum;
    row_stats(x, n, lane, mx, log_sum);
    const float inv = 1.f / expf(log_sum);
    float acc = 0.f;
    for (int i = lane; i < n; i += kLanes) acc += static_cast<float>(i - support) * (expf(x[i] - mx) * inv);
    const float v = wave_sum(acc);
    const float sgn = (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f);
    const float r = (sqrtf(1.f + 4.f * 0.001f * (fabsf(v) + 1.f + 0.001f)) - 1.f) / (2.f * 0.001f);
    return sgn * (r * r - 1.f);
}

__global__ __launch_bounds__(kLanes) void unroll_loss_kernel(mztrain_loss_args a) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int F = 2 * a.support_size + 1, A = a.actions, B = a.batch, K1 = a.steps;
    const float w = a.weight ? a.weight[b] : 1.f;
    float sum_v = 0.f, sum_r = 0.f, sum_p = 0.f;
    for (int k = 0; k < K1; ++k) {
        const size_t row = static_cast<size_t>(k) * B + b;
        const size_t tgt = static_cast<size_t>(b) * K1 + k;
        const float step = (k > 0) ? 1.f / a.gradient_scale[tgt] : 1.f;   // share of the gradient this step passes on
        const float lv = ce_two_hot(a.value_logits + row * F, a.grad_value + row * F, F, a.target_value[tgt],
                                    a.support_size, (w * a.value_loss_weight) * step, lane);
        sum_v = (k == 0) ? lv : sum_v + lv;
        if (k > 0) {
            const float lr = ce_two_hot(a.reward_logits + row * F, a.grad_reward + row * F, F, a.target_reward[tgt],
                                        a.support_size, w * step, lane);
            sum_r = (k == 1) ? lr : sum_r + lr;
        } else {
            for (int i = lane; i < F; i += kLanes) a.grad_reward[row * F + i] = 0.f;
        }
        const float lp = ce_dense(a.policy_logits + row * A, a.target_policy + tgt * A, a.grad_policy + row * A, A,
                                  w * step, lane);
        sum_p = (k == 0) ? lp : sum_p + lp;
        const float predicted = decode_row(a.value_logits + row * F, F, a.support_size, lane);
        if (lane == 0) a.priorities[tgt] = powf(fabsf(predicted - a.target_value[tgt]), a.per_alpha);
    }
    if (lane == 0) {
        a.head_sums[b] = sum_v;
        a.head_sums[B + b] = sum_r;
        a.head_sums[2 * static_cast<size_t>(B) + b] = sum_p;
        const float loss = (sum_v * a.value_loss_weight + sum_r) + sum_p;
        a.sample_loss[b] = a.weight ? loss * w : loss;
    }
}

}  // namespace

extern "C" int mztrain_unroll_loss(const mztrain_loss_args* args, void* stream) {
    if (!args || !args->value_logits || !args->reward_logits || !args->policy_logits || !args->target_value ||
        !args->target_reward || !args->target_policy || !args->gradient_scale || !args->sample_loss || !args->head_sums ||
        !args->priorities || !args->grad_value || !args->grad_reward || !args->grad_policy)
        return MZMCTS_ERR_INVALID;
    if (args->batch <= 0 || args->steps <= 0 || args->support_size <= 0 || args->actions <= 0) return MZMCTS_ERR_INVALID;
    unroll_loss_kernel<<<dim3(args->batch), dim3(kLanes), 0, static_cast<hipStream_t>(stream)>>>(*args);
    return hipGetLastError() == hipSuccess ? MZMCTS_OK : MZMCTS_ERR_HIP;
}
