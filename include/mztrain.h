/* mztrain.h -- C ABI of the trainer-side HIP kernels (libmzmcts.so), SURVEY.md section 8(f) row 4.
 *
 * One entry: the loss of Trainer.update_weights (reference trainer.py:176-215 and loss_function, trainer.py:271-291)
 * over ALL unrolled steps of a batch in one launch -- the two-hot value / reward targets (models.scalar_to_support,
 * models.py:665-685), the three cross-entropies per step, their sums over the steps in the reference's order, the
 * per-sample total with the value-loss weight and the PER importance weight, the new PER priorities
 * (|support_to_scalar(value logits) - target value| ** PER_alpha, trainer.py:199-209) and the gradient of the total
 * with respect to every logit (each unrolled step's share divided by its gradient scale, trainer.py:176-198).
 * The network forward / backward and the optimizer stay with PyTorch-ROCm; this replaces the ~25 element-wise
 * launches per unrolled step between them.  Raw device pointers, fp32, contiguous; no torch types.
 */
#ifndef MZTRAIN_H
#define MZTRAIN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mztrain_loss_args {
    const float *value_logits;   /* [K1][B][F]  step-major stack of the unroll's value logits, F = 2 support + 1 */
    const float *reward_logits;  /* [K1][B][F]  (step 0's row is ignored: no reward is predicted for the root)  */
    const float *policy_logits;  /* [K1][B][A] */
    const float *target_value;   /* [B][K1]     scalar targets (ReplayBuffer.make_target)                       */
    const float *target_reward;  /* [B][K1] */
    const float *target_policy;  /* [B][K1][A] */
    const float *gradient_scale; /* [B][K1] */
    const float *weight;         /* [B] PER importance-sampling weights, or NULL (config.PER off)               */
    int32_t batch;               /* B  */
    int32_t steps;               /* K1 = num_unroll_steps + 1 */
    int32_t support_size;
    int32_t actions;             /* A  */
    float value_loss_weight;
    float per_alpha;
    float *sample_loss;   /* out [B]      (value sum * value_loss_weight + reward sum + policy sum) * weight         */
    float *head_sums;     /* out [3][B]   per-sample sums over the steps of the value / reward / policy losses      */
    float *priorities;    /* out [B][K1] */
    float *grad_value;    /* out [K1][B][F]  d(sample_loss[b]) / d(value_logits[k][b][:]), gradient scale applied    */
    float *grad_reward;   /* out [K1][B][F]  (zeros for step 0) */
    float *grad_policy;   /* out [K1][B][A] */
} mztrain_loss_args;

/* Queues the launch on `stream` (hipStream_t); 0 = ok, < 0 = error (MZMCTS_ERR_*). */
int mztrain_unroll_loss(const mztrain_loss_args *args, void *stream);

#ifdef __cplusplus
}
#endif
#endif
