/*
 * mzenv.h -- C ABI of the device-resident batched game environments (part of libmzmcts.so).
 *
 * once the search runs at 10^7..10^8 simulations/s, stepping 4096
 * Python `Game` objects on the host (one `game.step` per env and move) is the bottleneck.  These
 * kernels keep E games of one kind resident on the GPU behind the same plugin semantics as the
 * reference's game files:
 *
 *   MZENV_TICTACTOE  games/tictactoe.py:242-305 (rules), :132-145 (reward x20), observation planes
 *                    [own stones, opponent stones, player to move] as int-valued floats
 *   MZENV_CONNECT4   games/connect4.py:219-304 (rules), :132-143 (reward x10)
 *   MZENV_CARTPOLE   games/cartpole.py wraps gym's CartPole-v1; gym is not vendored by the reference, so
 *                    this restates the published classic-control equations (Euler, tau = 0.02) exactly as
 *                    muzero-hypermodel_amd/games/cartpole.py does on the host -- parity UNPINNED against gym
 *
 * Conventions as in mzmcts.h: raw device / host pointers, hipStream_t as void*, 0 = ok, < 0 = error,
 * mzenv_last_error() for the text.  All launch functions are asynchronous and allocation-free.
 */
#ifndef MZENV_H
#define MZENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZENV_CARTPOLE 0
#define MZENV_TICTACTOE 1
#define MZENV_CONNECT4 2

typedef struct mzenv mzenv;

/* seeds: host u32[E]; env e behaves like `Game(seeds[e])` of the host plugin.  Blocking. */
int mzenv_create(int32_t game, int32_t num_envs, int32_t device, const uint32_t *seeds, mzenv **out);
void mzenv_destroy(mzenv *env);
const char *mzenv_last_error(const mzenv *env);

/* static shape of the game: actions A, players, observation (C,H,W) */
int mzenv_shape(const mzenv *env, int32_t *num_actions, int32_t *num_players, int32_t *obs_shape3);

/* Game.reset() for every env whose mask byte is non-zero (mask dev u8[E], NULL = all envs). */
int mzenv_reset(mzenv *env, const uint8_t *mask, void *stream);

/* Game.step(action) for every env (envs with action < 0 are left untouched; their reward and done read 0):
 *   actions dev i32[E];  reward_out dev f32[E];  done_out dev u8[E] */
int mzenv_step(mzenv *env, const int32_t *actions, float *reward_out, uint8_t *done_out, void *stream);

/* What the search needs before a move: the observation batch, Game.legal_actions() in the plugin's
 * order, Game.to_play().
 *   obs_out dev f32[E,C,H,W];  legal_out dev i32[E,A] (first num_legal[e] entries valid);
 *   num_legal_out dev i32[E];  to_play_out dev i32[E] */
int mzenv_observe(mzenv *env, float *obs_out, int32_t *legal_out, int32_t *num_legal_out, int32_t *to_play_out,
                  void *stream);

/* One self-play move of every env in a single call (one launch on `stream`): Game.step(actions), the
 * observation after the move (terminal observations included) -> obs_after_out, Game.reset() of the envs that
 * just finished, and the observation the next search sees -> obs_next_out (legal / num_legal / to_play outputs
 * describe that next state). */
int mzenv_advance(mzenv *env, const int32_t *actions, float *reward_out, uint8_t *done_out, float *obs_after_out,
                  float *obs_next_out, int32_t *legal_out, int32_t *num_legal_out, int32_t *to_play_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MZENV_H */
