/*
 * mzreplay.h -- C ABI of the device-resident replay store (part of libmzmcts.so).
 *
 * SURVEY.md section 8(f) row 2, the direct consumer of the search's output: finished games
 * (GameHistory, reference self_play.py:480-548) are kept on the GPU as packed arrays and turned into
 * training targets there:
 *
 *   mzreplay_add_games      ReplayBuffer.save_game: initial priorities |root_value - target|^alpha and the
 *                           game priority (replay_buffer.py:33-50)
 *   mzreplay_game_observations / mzreplay_set_reanalysed
 *                           the two ends of Reanalyse's per-game step (replay_buffer.py:335-356): the stacked
 *                           observations of every position of a game as one inference batch, and the fresh
 *                           root values (float32) that compute_target_value bootstraps from afterwards
 *   mzreplay_make_batch     ReplayBuffer.make_target / compute_target_value and
 *                           GameHistory.get_stacked_observations for a batch of (game, position) pairs
 *                           (replay_buffer.py:222-295, self_play.py:514-548)
 *
 * Which games and positions go into a batch, and the random actions of absorbing states, are the
 * caller's draws (the reference takes them from numpy's legacy RandomState in a fixed order:
 * replay_buffer.py:67-195); this library only evaluates them.  fp64 sums run in the reference's order with
 * `discount ** i` taken from a table the caller fills with its own libm (Python floats), so values and
 * policies are bit-identical to the reference's; priorities go through the device's pow() and agree to
 * float32 rounding.
 *
 * Conventions as in mzmcts.h: raw host / device pointers, hipStream_t as void*, 0 = ok, < 0 = error.
 */
#ifndef MZREPLAY_H
#define MZREPLAY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mzreplay mzreplay;

typedef struct mzreplay_config {
    int32_t capacity;             /* game slots (config.replay_buffer_size) */
    int32_t max_moves;            /* longest game, in moves */
    int32_t num_actions;
    int32_t obs_channels, obs_height, obs_width;
    int32_t stacked_observations;
    int32_t td_steps, num_unroll_steps;
    int32_t device;
    double per_alpha;
    const double *discount_powers; /* host f64[td_steps + 1]: config.discount ** i, i = 0..td_steps */
} mzreplay_config;

int mzreplay_create(const mzreplay_config *config, mzreplay **out);
void mzreplay_destroy(mzreplay *store);
const char *mzreplay_last_error(const mzreplay *store);

/* Store n games in the given slots (overwriting what was there).  Host arrays, game-major, padded to
 * max_moves: lengths i32[n] (moves), observations f32[n][max_moves+1][C*H*W], actions i32[n][max_moves+1],
 * rewards f64[n][max_moves+1], to_play i32[n][max_moves+1], child_visits f64[n][max_moves][A],
 * root_values f64[n][max_moves].  Outputs (host, may be NULL): priorities f32[n][max_moves],
 * game_priority f32[n].  Blocking. */
int mzreplay_add_games(mzreplay *store, int32_t n, const int32_t *slots, const int32_t *lengths,
                       const float *observations, const int32_t *actions, const double *rewards,
                       const int32_t *to_play, const double *child_visits, const double *root_values,
                       float *priorities, float *game_priority, void *stream);

/* Targets of a batch.  Host inputs: slots i32[B], positions i32[B], absorbing_actions i32[B][U+1] (entry u is
 * used where position + u lies past the end of the game).  Device outputs: observations
 * f32[B][C'][H][W] with C' = C + stacked * (C + 1), actions i64[B][U+1], values / rewards /
 * gradient_scale f64[B][U+1], policies f64[B][U+1][A].  Asynchronous on `stream`. */
int mzreplay_make_batch(mzreplay *store, int32_t batch, const int32_t *slots, const int32_t *positions,
                        const int32_t *absorbing_actions, float *observations, int64_t *actions, double *values,
                        double *rewards, double *policies, double *gradient_scale, void *stream);

/* Stacked observations of positions 0..length-1 of the game in `slot`: observations dev f32[length][C'][H][W].
 * Asynchronous. */
int mzreplay_game_observations(mzreplay *store, int32_t slot, int32_t length, float *observations, void *stream);
/* Reanalyse's values for that game: f32[length], host or device memory.  From now on the game's targets
 * bootstrap from them, accumulating in float32 as the reference does once the values are a numpy float32 array
 * (NumPy >= 2 promotion; recorded in fixture G13).  mzreplay_add_games into the slot forgets them.  Asynchronous. */
int mzreplay_set_reanalysed(mzreplay *store, int32_t slot, const float *values, int32_t length, void *stream);

/* Bytes of device memory the store occupies. */
int64_t mzreplay_device_bytes(const mzreplay *store);

#ifdef __cplusplus
}
#endif
#endif /* MZREPLAY_H */
