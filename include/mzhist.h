/*
 * mzhist.h -- C ABI of the host-side game-history filer (part of libmzmcts.so).
 *
 * SelfPlay.play_game's bookkeeping (reference self_play.py:116-121, 176-182, 497-512) for E environments and a
 * whole batch of moves at a time: the results of a move batch (mzmcts_moves_collect: actions, root visit
 * counts, root value sums -- passed with byte strides, so the search's pinned download ring can be handed over
 * as it is) and of the environment kernels (rewards, done flags, observations) are appended to one packed
 * history row per env; games that ended inside the batch come back as packed arrays in GameHistory's layout
 *     observations[i, 0..n]  actions[i, 0..n] (index 0 = the reference's dummy action 0)  rewards[i, 0..n]
 *     to_play[i, 0..n]       child_visits[i, 0..n-1][A] (visit counts / num_simulations, by action)
 *     root_values[i, 0..n-1] (root value sum / num_simulations)
 * Pure host code (memory movement over E x M small records), spread over the library's worker pool.
 */
#ifndef MZHIST_H
#define MZHIST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mzhist mzhist;

int mzhist_create(int32_t num_envs, int32_t max_moves, int32_t obs_floats, int32_t num_actions, mzhist **out);
void mzhist_destroy(mzhist *hist);
const char *mzhist_last_error(const mzhist *hist);

/* Every env starts a game: reset observations f32[E][obs], to_play i32[E] (NULL = player 0). */
int mzhist_begin(mzhist *hist, const float *first_observations, const int32_t *first_to_play);

typedef struct mzhist_moves {
    int32_t n_moves;              /* M */
    int32_t num_simulations;
    const int32_t *moves_done;    /* [E]: env e played the first moves_done[e] moves of the batch */
    const void *actions;          /* move m: i32[E] at actions + m * actions_stride (bytes) */
    int64_t actions_stride;
    const void *visits;           /* move m: i32[E][A], root children by child slot */
    int64_t visits_stride;
    const void *root_value_sum;   /* move m: f64[E] */
    int64_t root_value_sum_stride;
    const int32_t *legal;         /* [E][A] child slot -> action (the batch's legal action sets) */
    const int32_t *num_legal;     /* [E] */
    const float *rewards;         /* [M][E] */
    const uint8_t *done;          /* [M][E] game over after this move */
    const float *obs_after;       /* [M][E][obs] observation after the move (terminal one included) */
    const float *obs_next;        /* [M][E][obs] observation the next search sees (reset where done) */
    const int32_t *to_play_after; /* [M][E] or NULL (single player) */
    const int32_t *to_play_next;  /* [M][E] or NULL */
    int64_t legal_stride;         /* bytes from one move's legal sets to the next move's; 0 = the batch has one set */
    int64_t num_legal_stride;     /* (board games: mzmcts_moves_inputs hands out [M][E][A] / [M][E]) */
} mzhist_moves;

/* File the batch; *n_finished = games that ended in it. */
int mzhist_file(mzhist *hist, const mzhist_moves *moves, int32_t *n_finished);

/* The games finished by the last mzhist_file, packed with `row_moves` = the longest one's length:
 * env_index i32[n], length i32[n], observations f32[n][row_moves+1][obs], actions i32[n][row_moves+1],
 * rewards f32[n][row_moves+1], to_play i32[n][row_moves+1], child_visits f64[n][row_moves][A],
 * root_values f64[n][row_moves].  Pointers stay valid until the next mzhist_file. */
int mzhist_finished(mzhist *hist, const int32_t **env_index, const int32_t **length, const float **observations,
                    const int32_t **actions, const float **rewards, const int32_t **to_play,
                    const double **child_visits, const double **root_values, int32_t *row_moves);

/* Exchange the rows of the RUNNING games with the caller (load != 0: take them over, load == 0: hand them back):
 * observations f32[E][max_moves+1][obs], actions i32[E][max_moves+1], rewards f32[E][max_moves+1],
 * to_play i32[E][max_moves+1], child_visits f64[E][max_moves][A], root_values f64[E][max_moves], lengths i32[E]. */
int mzhist_rows(mzhist *hist, float *observations, int32_t *actions, float *rewards, int32_t *to_play,
                double *child_visits, double *root_values, int32_t *lengths, int32_t load);

/* Moves filed so far in env e's running game (i32[E]). */
const int32_t *mzhist_lengths(const mzhist *hist);

#ifdef __cplusplus
}
#endif
#endif /* MZHIST_H */
