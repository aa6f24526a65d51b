/*
 * mzmcts.h -- C ABI of the MI355X batched-MCTS engine (libmzmcts.so).
 *
 * This is the drop-in boundary for the self-play / MCTS hot path of
 * jiawei415/muzero-hypermodel (a snapshot of muzero-general).  The reference has no native code:
 * its "FFI" for this path is the Python surface of self_play.py.  Each entry point below names the
 * reference code it replaces (file:line relative to the reference root).  The Python host in
 * muzero-hypermodel_amd/ binds these symbols with ctypes (muzero-hypermodel_amd/_native.py) and
 * re-creates the reference classes (SelfPlay / MCTS / Node / GameHistory / MinMaxStats) on top.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  "dev" pointers are HIP device pointers whose
 *     lifetime the caller guarantees for the duration of the call (torch tensors cross as
 *     tensor.data_ptr()); "host" pointers are ordinary host memory.
 *   - every function that launches work takes the hipStream_t to launch on as `void *stream`
 *     (0 = the null stream).  Kernels are asynchronous; functions documented as "blocking" synchronise
 *     that stream before returning.  Launch functions perform no allocation and no synchronisation,
 *     so they may be captured into a hipGraph.
 *   - one caller thread and one stream per engine; one engine (process) per GPU
 *     (reference: one single-threaded Ray actor per worker, self_play.py:11-29).
 *   - return value: 0 = ok, < 0 = error; mzmcts_last_error() gives the message.  Plugin-contract
 *     violations use the reference's assertion texts (self_play.py:297-302).
 *   - E trees are searched in lock step.  Tree e uses RNG stream e, a clone of numpy's legacy
 *     RandomState seeded like reference worker `config.seed + e` (muzero.py:175, self_play.py:22).
 *   - child statistics are indexed by child SLOT: slot i of the root is the i-th entry of the
 *     legal-action list handed to mzmcts_begin_search (the reference's dict insertion order,
 *     self_play.py:303-309, 464-466); below the root slot == action (self_play.py:346-352).
 */
#ifndef MZMCTS_H
#define MZMCTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZMCTS_ABI_VERSION 2

/* error codes */
#define MZMCTS_OK 0
#define MZMCTS_ERR_INVALID (-1)      /* bad argument / bad call order */
#define MZMCTS_ERR_HIP (-2)          /* HIP runtime failure (no device, OOM, launch error) */
#define MZMCTS_ERR_EMPTY_LEGAL (-3)  /* "Legal actions should not be an empty array." */
#define MZMCTS_ERR_LEGAL_RANGE (-4)  /* "Legal actions should be a subset of the action space." */
#define MZMCTS_ERR_PLAYERS (-5)      /* "More than two player mode not implemented." */

typedef struct mzmcts_engine mzmcts_engine;

/* Fields of MuZeroConfig the tree kernels read (games/cartpole.py:21-46 etc.). */
typedef struct mzmcts_config {
    int32_t num_envs;        /* E: trees searched in lock step on this GPU                    */
    int32_t num_actions;     /* A = len(config.action_space)                                  */
    int32_t num_simulations; /* S = config.num_simulations                                    */
    int32_t num_players;     /* len(config.players): 1 or 2 (self_play.py:412-431)            */
    int32_t support_size;    /* config.support_size; F = 2*support_size+1 logits              */
    int32_t hidden_floats;   /* H: fp32 values per hidden state (0 = engine keeps no states)  */
    int32_t device;          /* HIP device ordinal                                            */
    int32_t group_width;     /* lanes of a wavefront per tree: 0 = auto (pow2 >= min(A,64)); or a power of two
                                in [auto, 64] to give each tree more lanes (fused FC search); or auto / 2
                                (4 or 8, when A > auto / 2): two children per lane, half the wavefronts of
                                the tree kernels; or 1 when A == 2                                     */
    double discount;                  /* config.discount                                      */
    double pb_c_base;                 /* config.pb_c_base                                     */
    double pb_c_init;                 /* config.pb_c_init                                     */
    double root_dirichlet_alpha;      /* config.root_dirichlet_alpha                          */
    double root_exploration_fraction; /* config.root_exploration_fraction                     */
    void *hidden_pool; /* optional caller-owned dev buffer f32[(S+1), E, H] (e.g. a torch tensor the
                          network writes next states into); NULL = the engine allocates it       */
} mzmcts_config;

/* Root statistics of every tree after a search, host arrays filled by mzmcts_readout().
 * Any pointer may be NULL.  What the reference reads off `root` after MCTS.run
 * (self_play.py:230-233, 500-510; diagnose_model.py:36-72). */
typedef struct mzmcts_root_stats {
    int32_t *visits;          /* [E,A] child.visit_count per slot                              */
    double *child_value_sum;  /* [E,A] child.value_sum                                         */
    double *child_prior;      /* [E,A] child.prior (after exploration noise)                   */
    double *child_reward;     /* [E,A] child.reward                                            */
    int32_t *child_expanded;  /* [E,A] 1 if the child was expanded (has a hidden state)        */
    double *root_value_sum;   /* [E]   root.value_sum                                          */
    int32_t *root_visits;     /* [E]   root.visit_count                                        */
    int32_t *max_tree_depth;  /* [E]   mcts_info["max_tree_depth"]                             */
    double *root_predicted_value; /* [E] mcts_info["root_predicted_value"]                     */
    double *min_max;          /* [E,2] MinMaxStats.minimum, .maximum                           */
    int64_t *depth_sum;       /* [E]   sum over simulations of the select depth (for d-bar)    */
    uint32_t *tie_break_words;/* [E]   32-bit RNG words the tie-breaks of this search drew     */
} mzmcts_root_stats;

/* HIP-event timings accumulated while profiling is on (bench.py roofline leg). */
typedef struct mzmcts_profile {
    double select_ms, expand_backup_ms, root_ms, fused_ms; /* summed kernel durations      */
    int64_t select_launches, expand_backup_launches, root_launches, fused_launches;
    int64_t select_depth_sum;  /* sum over launches and trees of the select depth (d-bar); move batches add
                                * to it only while profiling is on (mzmcts_set_profiling) */
    int64_t simulations;       /* tree-simulations executed (launches x active trees)     */
    double step_ms;            /* mzmcts_expand_backup_select launches (expand_backup + next select in one)  */
    int64_t step_launches;
} mzmcts_profile;

/* ---- lifetime -------------------------------------------------------------------------------
 * Replaces SelfPlay.__init__'s per-worker state (self_play.py:17-29) and the per-move
 * `MCTS(config)` / `Node(0)` / `MinMaxStats()` allocations (self_play.py:145, 280, 317): the
 * engine owns every device pool for the actor's lifetime and recycles them per move. */
int mzmcts_abi_version(void);
int mzmcts_create(const mzmcts_config *config, mzmcts_engine **out);
void mzmcts_destroy(mzmcts_engine *engine);
const char *mzmcts_last_error(const mzmcts_engine *engine); /* engine may be NULL (create errors) */

/* ---- RNG streams (numpy.random.seed / get_state / set_state; self_play.py:22) ---------------
 * seeds: host u32[E]; stream e behaves like `numpy.random.seed(seeds[e])`.  Blocking. */
int mzmcts_seed(mzmcts_engine *engine, const uint32_t *seeds, void *stream);
/* Exchange one stream with numpy's global generator state (key[624], pos, has_gauss, gauss) so a
 * single-env facade interleaves exactly with other users of numpy.random.  Blocking. */
int mzmcts_rng_set_state(mzmcts_engine *engine, int32_t env, const uint32_t *key, int32_t pos,
                         int32_t has_gauss, double cached_gaussian, void *stream);
int mzmcts_rng_get_state(mzmcts_engine *engine, int32_t env, uint32_t *key, int32_t *pos,
                         int32_t *has_gauss, double *cached_gaussian, void *stream);

/* ---- one search = MCTS.run (self_play.py:261-362) -------------------------------------------
 * begin_search: host side of the root set-up (self_play.py:297-315).
 *   legal_actions host i32[E,A]: row e holds num_legal[e] actions in the plugin's order
 *   num_legal     host i32[E];   0 marks env e inactive for this search (no RNG use, no work)
 *   to_play       host i32[E]:   Game.to_play() per env
 *   Draws the Dirichlet noise for every active env from its RNG stream (numpy legacy gamma
 *   sampler, host libm) when add_exploration_noise != 0 and queues the uploads on `stream`.
 *   noise_out (optional host f64[E,A]) receives the noise vectors. */
int mzmcts_begin_search(mzmcts_engine *engine, const int32_t *legal_actions, const int32_t *num_legal,
                        const int32_t *to_play, int32_t add_exploration_noise, double *noise_out,
                        void *stream);

/* expand_roots: device side of root.expand + add_exploration_noise (self_play.py:293-315,
 * 452-477) from initial_inference outputs (models.py:172-190):
 *   value_logits  dev f32[E,F]  -> root_predicted_value via support_to_scalar (models.py:641-662)
 *   reward_logits dev f32[E,F] or NULL (NULL = reward 0, what log(one_hot) decodes to)
 *   policy_logits dev f32[E,A]  -> fp32 softmax over the legal logits only
 *   root_hidden   dev f32[E,H] or NULL when the caller wrote pool slab 0 in place */
int mzmcts_expand_roots(mzmcts_engine *engine, const float *value_logits, const float *reward_logits,
                        const float *policy_logits, const float *root_hidden, void *stream);
/* Injected-mode root (parity tests): reward dev f64[E], pre-noise priors dev f64[E,A] per slot. */
int mzmcts_expand_roots_injected(mzmcts_engine *engine, const double *root_reward,
                                 const double *root_priors, void *stream);

/* select: the `while node.expanded()` descent with select_child / ucb_score for all E trees
 * (self_play.py:321-335, 364-405) plus the gather of the parent hidden states into one contiguous
 * batch for recurrent_inference (self_play.py:339-343):
 *   parent_hidden_out dev f32[E,H] or NULL (skip the gather), action_out dev i64[E] or NULL */
int mzmcts_select(mzmcts_engine *engine, float *parent_hidden_out, int64_t *action_out, void *stream);
/* The same, with the gather laid out as the residual networks' dynamics input (models.py:553-568): row e of
 * planes_out = [parent hidden state, hidden_floats = channels x plane | one plane of action / action_space]:
 *   planes_out dev f32[E, channels + 1, plane], action_out dev i64[E] (also written, as by mzmcts_select). */
int mzmcts_select_planes(mzmcts_engine *engine, float *planes_out, int64_t *action_out, int32_t plane,
                         int32_t action_space, void *stream);

/* expand_backup: support_to_scalar on value/reward (self_play.py:344-345), node.expand over the
 * full action space (self_play.py:346-352, 452-466), backpropagate with MinMaxStats
 * (self_play.py:354, 407-431, 560-562), max_tree_depth (self_play.py:356).
 *   value_logits/reward_logits dev f32[E,F], policy_logits dev f32[E,A],
 *   next_hidden dev f32[E,H] or NULL when the caller wrote slab mzmcts_next_slab() in place. */
int mzmcts_expand_backup(mzmcts_engine *engine, const float *value_logits, const float *reward_logits,
                         const float *policy_logits, const float *next_hidden, void *stream);
/* Injected mode: already-decoded scalars (value, reward dev f64[E]; priors dev f64[E,A]). */
int mzmcts_expand_backup_injected(mzmcts_engine *engine, const double *value, const double *reward,
                                  const double *priors, void *stream);

/* One step of the lock-step loop in ONE launch: expand_backup of the simulation in flight followed by select of the
 * next one (self_play.py:344-356, 407-431, then 321-343 of the next iteration) -- most of the nodes a descent visits
 * were just rewritten by the backup and are still in the XCD's L2, and a launch boundary goes.  Arguments as
 * mzmcts_expand_backup + mzmcts_select (/ mzmcts_select_planes / the injected form); not for the last simulation of a
 * search (use mzmcts_expand_backup there).  Results are bit-identical to the two separate calls. */
int mzmcts_expand_backup_select(mzmcts_engine *engine, const float *value_logits, const float *reward_logits,
                                const float *policy_logits, const float *next_hidden, float *parent_hidden_out,
                                int64_t *action_out, void *stream);
int mzmcts_expand_backup_select_planes(mzmcts_engine *engine, const float *value_logits, const float *reward_logits,
                                       const float *policy_logits, const float *next_hidden, float *planes_out,
                                       int64_t *action_out, int32_t plane, int32_t action_space, void *stream);
int mzmcts_expand_backup_select_injected(mzmcts_engine *engine, const double *value, const double *reward,
                                         const double *priors, float *parent_hidden_out, int64_t *action_out, void *stream);

/* Hidden-state pool: slab k (dev f32[E,H]) holds the state of the node expanded by simulation k-1
 * (slab 0 = roots).  mzmcts_next_slab() is the slab the coming expand_backup will own. */
float *mzmcts_hidden_slab(mzmcts_engine *engine, int32_t slab);
int32_t mzmcts_next_slab(const mzmcts_engine *engine);
int32_t mzmcts_simulations_done(const mzmcts_engine *engine);
/* Set the host-side simulation counter without touching device state: 0 before a captured
 * hipGraph of S simulations is replayed for a new move, S after the replay (the launches the graph
 * replays are not seen by the host-side counter). */
int mzmcts_set_simulations_done(mzmcts_engine *engine, int32_t n);

/* ---- results ---------------------------------------------------------------------------------
 * readout: copy the root statistics of all trees to the host (blocking) and advance the host RNG
 * mirrors by the tie-break words each tree consumed. */
int mzmcts_readout(mzmcts_engine *engine, const mzmcts_root_stats *out, void *stream);
/* Asynchronous form: queue the device->host copies behind the search on `stream` and return at once;
 * the next mzmcts_readout() then only waits for them.  Lets a host that drives several engines (env
 * groups) overlap one group's host work with the other groups' kernels. */
int mzmcts_readout_begin(mzmcts_engine *engine, void *stream);
/* SelfPlay.select_action (self_play.py:223-246) for every env on its own RNG stream, using the
 * visit counts of the last readout.  temperature host f64[E] (0 = argmax, +inf = uniform);
 * action_out host i32[E] (action ids), slot_out optional host i32[E].  Inactive envs give -1. */
int mzmcts_sample_actions(mzmcts_engine *engine, const double *temperature, int32_t *action_out,
                          int32_t *slot_out);
/* GameHistory.store_search_statistics (self_play.py:497-512): child_visits host f64[E,A] over the
 * full action space, root_values host f64[E]. */
int mzmcts_search_statistics(mzmcts_engine *engine, double *child_visits, double *root_values);

/* Path of the most recent select per tree (tests, diagnose-style callers): depth host i32[E],
 * actions host i32[E,S] (-1 padded), tie_counts host i32[E,S].  Blocking. */
int mzmcts_last_paths(mzmcts_engine *engine, int32_t *depth, int32_t *actions, int32_t *tie_counts,
                      void *stream);
/* Make select also record the size of each tie list (needed for tie_counts above; off by default
 * because it adds a 4-byte store per level).  Allocates: call outside graph capture. */
int mzmcts_set_debug_ties(mzmcts_engine *engine, int32_t enabled);
/* Whole tree of one env (Node facade; diagnose_model.py walks root.children recursively):
 * host arrays over (S+1)*A child records, record (k*A + slot) = child `slot` of expanded node k;
 * child_node[k*A+slot] = index of that child's own expanded node or -1.  Blocking. */
int mzmcts_export_tree(mzmcts_engine *engine, int32_t env, int32_t *visits, double *value_sum,
                       double *prior, double *reward, int32_t *child_node, void *stream);

/* ---- fully-connected networks in-kernel --------------------------------------------------------
 * MuZeroFullyConnectedNetwork's inference half (models.py:128-195) as HIP device code, so that a whole
 * move -- initial_inference, root expansion and all S x (select, recurrent_inference, expand, backup)
 * -- runs in ONE launch with every tree, its hidden states, the activations and the weights resident
 * in LDS (trees are independent, so nothing synchronises between them).  Residual networks keep the
 * lock-step path above with any external inference engine. */
typedef struct mzmcts_fc_desc {
    int32_t observation_floats; /* flattened (stacked) observation size                              */
    int32_t encoding_size;      /* config.encoding_size == hidden_floats                             */
    int32_t n_hidden[5];        /* hidden-layer counts of: representation, dynamics, reward, policy, value */
    int32_t hidden[5][3];       /* their widths (fc_*_layers), at most 3 hidden layers each          */
} mzmcts_fc_desc;
/* weights: dev f32[n_weights], every Linear's weight then bias in state_dict order (representation,
 * dynamics_encoded_state, dynamics_reward, prediction_policy, prediction_value).  The pointer is
 * retained (e.g. the flat buffer an RCCL weight broadcast lands in): refreshing it refreshes the net. */
int mzmcts_fc_configure(mzmcts_engine *engine, const mzmcts_fc_desc *desc, const float *weights,
                        int64_t n_weights);
/* Lock-step form of the same device code (one launch per [E,.] batch), outputs as the torch modules':
 *   initial:   observations dev f32[E,obs]        recurrent: hidden dev f32[E,enc], action dev i64[E]
 *   value_logits / reward_logits dev f32[E,F], policy_logits dev f32[E,A], hidden_out dev f32[E,enc] */
int mzmcts_fc_initial_inference(mzmcts_engine *engine, const float *observations, float *value_logits,
                                float *reward_logits, float *policy_logits, float *hidden_out, void *stream);
int mzmcts_fc_recurrent_inference(mzmcts_engine *engine, const float *hidden, const int64_t *action,
                                  float *value_logits, float *reward_logits, float *policy_logits,
                                  float *hidden_out, void *stream);
/* Whole search in one launch (after mzmcts_begin_search): observations dev f32[E,obs].
 * hidden_in_lds: 1 = keep hidden states in LDS too when they fit, 0 = read them back from the HBM pool.
 * Leaves the engine in the same state as expand_roots + S x (select, expand_backup): readout,
 * sample_actions, search_statistics and export_tree work unchanged. */
int mzmcts_search_fused_fc(mzmcts_engine *engine, const float *observations, int32_t hidden_in_lds,
                           void *stream);
/* Dynamic LDS bytes per workgroup the fused kernel would use (0 = not configured / does not fit). */
int64_t mzmcts_fused_lds_bytes(mzmcts_engine *engine, int32_t hidden_in_lds);
/* Two whole-move kernels exist.  GENERIC handles any MuZeroFullyConnectedNetwork (activations and weights
 * in LDS).  NARROW handles networks whose every layer fits one 16-lane row (group_width 16, one hidden
 * layer of <= 16 units per MLP -- none or one for the representation --, encoding_size + actions <= 16,
 * observation <= 16 floats, support <= 32 logits: the reference's cartpole.py:61-71) with activations in
 * registers; AUTO (default) picks NARROW when the network qualifies.  The lock-step mzmcts_fc_*_inference
 * calls follow the same choice, so either fused kernel can be compared with the lock-step search bit for bit.
 * publish_tree: 1 (default) = the fused kernel copies the whole tree to the HBM pools (mzmcts_export_tree,
 * mzmcts_hidden_slab); 0 = only what MCTS.run's callers consume, the root's children and statistics. */
#define MZMCTS_FUSED_AUTO 0
#define MZMCTS_FUSED_GENERIC 1
#define MZMCTS_FUSED_NARROW 2
int mzmcts_set_fused_options(mzmcts_engine *engine, int32_t variant, int32_t publish_tree);
/* The variant mzmcts_search_fused_fc would launch now: MZMCTS_FUSED_GENERIC / _NARROW, 0 = none fits. */
int32_t mzmcts_fused_variant(mzmcts_engine *engine);

/* ---- batches of moves without host round trips ------------------------------------------------
 * SelfPlay.play_game's per-move loop (self_play.py:129-182) for all envs, n_moves at a time, as kernels
 * queued back to back on one stream: whole-move search (which also samples the move's action with
 * SelfPlay.select_action on the tree's RNG stream), then whatever the caller queues on the same stream to
 * turn those actions into the next observations (mzenv_step / mzenv_observe, include/mzenv.h), then the
 * next search.  The host draws every move's Dirichlet noise up front and is not involved again until
 * mzmcts_moves_collect.
 *
 * The pre-drawn noise assumes what the reference's RNG order implies for the common case -- search m spends
 * one word on UCB tie-breaks (the first simulation's, when every root child still scores 0; tie-break words
 * come ahead of the next Dirichlet draw in the stream) and select_action consumes 0 (T = 0) or 2 (T = 1/k) words.  Where that fails the env simply is not searched from the next move on (moves_done[e] <
 * n_moves, its actions read -1: mzenv_step leaves such an env untouched) and the stream mirror is put back;
 * the caller plays the missing moves in the next batch.  Played moves are bit-identical to the one-at-a-time
 * path.  Legal action sets are those of a game whose action set does not change between moves (the Dirichlet
 * dimension must be known in advance); temperature per env must be 0, +inf (one move per batch) or 1/k with
 * k = 1..4, the values for which visit_count ** (1 / T) is exact integer arithmetic.
 *   prepare   blocking host work (noise rows of the whole batch) + asynchronous uploads
 *   predraw_next / submit_next
 *             the same in two halves, so that the host draws batch b+1 WHILE batch b runs: predraw_next (any
 *             time between b's last enqueue and its collect) continues the RNG mirror past b under the same
 *             assumptions; collect(b) redraws the rows of envs that ended differently; submit_next uploads
 *   enqueue   one search of the prepared batch, asynchronous; observations dev f32[E, obs]
 *   actions   device pointer, i32[E], of move `move`'s sampled actions (valid until the next prepare)
 *   collect   blocking: moves_done i32[E]; per move m < enqueued: actions i32[M,E], visits i32[M,E,A] (root
 *             children by child slot, i.e. in legal-action order), root_value_sum f64[M,E],
 *             root_predicted f32[M,E], max_depth i32[M,E]   (any output may be NULL) */
int mzmcts_moves_prepare(mzmcts_engine *engine, int32_t n_moves, const int32_t *legal_actions,
                         const int32_t *num_legal, const int32_t *to_play, int32_t add_exploration_noise,
                         const double *temperature, void *stream);
int mzmcts_moves_predraw_next(mzmcts_engine *engine, int32_t n_moves, const int32_t *legal_actions,
                              const int32_t *num_legal, const int32_t *to_play, int32_t add_exploration_noise,
                              const double *temperature);
int mzmcts_moves_submit_next(mzmcts_engine *engine, void *stream);
/* Drop a pre-drawn batch that will not be run: the RNG mirror goes back to where the device copy stands. */
int mzmcts_moves_discard_next(mzmcts_engine *engine);
/* A batch whose inputs live on the DEVICE, for games whose legal action sets change from move to move (board games):
 * legal_actions i32[E][A], num_legal i32[E], to_play i32[E] are device arrays the caller's environment kernels
 * (include/mzenv.h mzenv_advance) rewrite between the moves of the batch, on the same stream; every move's search reads
 * them as they are when it runs.  The exploration noise is drawn on the device (numpy.random.dirichlet on each tree's own
 * stream, 0 < root_dirichlet_alpha <= 1), because the length of a move's noise row -- its legal count -- is not known
 * to the host before the moves before it have been played; nothing is pre-drawn, nothing can stall.  temperature: host
 * f64[E] (0, inf or 1/k).  Then mzmcts_moves_enqueue per move and mzmcts_moves_collect as for a host-input batch (the
 * mirrors of the RNG streams step over every word the batch consumed); mzmcts_moves_inputs afterwards returns what each
 * move was searched with: num_legal i32[M][E], legal i32[M][E][A] (child slot -> action), to_play i32[M][E] (any NULL). */
int mzmcts_moves_prepare_device(mzmcts_engine *engine, int32_t n_moves, const int32_t *legal_actions,
                                const int32_t *num_legal, const int32_t *to_play, int32_t add_exploration_noise,
                                const double *temperature, void *stream);
int mzmcts_moves_inputs(mzmcts_engine *engine, int32_t *num_legal, int32_t *legal_actions, int32_t *to_play);
int mzmcts_moves_enqueue(mzmcts_engine *engine, const float *observations, void *stream);
/* A move of a device-input batch searched LOCK-STEP -- any network, the caller runs it between the tree launches -- with
 * no host round trip (self_play.py:129-182 for every env, queued on one stream):
 *   mzmcts_moves_begin_lockstep   records the move's inputs for the host, copies them into the engine's own root inputs,
 *                                 steps over the mirrors' pending RNG words, draws the exploration noise on the device;
 *                                 the engine is then where mzmcts_begin_search leaves it
 *   ... mzmcts_expand_roots, S x (mzmcts_select* / the network / mzmcts_expand_backup) -- or the replay of a hipGraph that
 *       holds them ...
 *   mzmcts_moves_end_lockstep     SelfPlay.select_action (self_play.py:223-246) on each tree's own stream from the root's
 *                                 visit counts, the move's slot of the output ring (mzmcts_moves_actions(move) is what
 *                                 the environment kernels step with), and the batch moves on to its next move
 * mzmcts_moves_collect afterwards as for any batch.  Needs no fully-connected network (mzmcts_moves_enqueue does). */
int mzmcts_moves_begin_lockstep(mzmcts_engine *engine, void *stream);
int mzmcts_moves_end_lockstep(mzmcts_engine *engine, void *stream);
/* play_game's temperature rule inside a device-input batch (self_play.py:152-158: the given temperature only while
 * len(game_history.action_history) < temperature_threshold, the best action afterwards).  Call right after
 * mzmcts_moves_prepare_device: game_moves host i32[E] = moves already played in each env's current game; the kernels keep
 * the counters from there (every searched move adds one).  threshold 0 = no rule.
 * mzmcts_moves_finished: dev u8[E], the environment kernels' `done` flags of the move just played (mzenv_advance's
 * done_out) -- envs flagged there start a new game, their counter restarts at the next move of the batch.  One-shot:
 * consumed by the next mzmcts_moves_enqueue / mzmcts_moves_begin_lockstep. */
int mzmcts_moves_temperature_threshold(mzmcts_engine *engine, int32_t threshold, const int32_t *game_moves, void *stream);
int mzmcts_moves_finished(mzmcts_engine *engine, const uint8_t *finished);
const int32_t *mzmcts_moves_actions(mzmcts_engine *engine, int32_t move);
int mzmcts_moves_collect(mzmcts_engine *engine, int32_t *moves_done, int32_t *actions, int32_t *visits,
                         double *root_value_sum, float *root_predicted, int32_t *max_depth, void *stream);
/* Zero-copy access to what collect downloads: the pinned host ring holds one block per move, `move_stride`
 * bytes apart, with actions i32[E], visits i32[E][A], root_value_sum f64[E], root_predicted f32[E], max_depth
 * i32[E] at offsets[0..4]; the ring holds capacity_moves blocks.  Valid from a collect until the next one; entries of env e in moves >=
 * moves_done[e] are undefined (its action reads -1 in the first such move). */
int mzmcts_moves_ring(mzmcts_engine *engine, void **host_base, int64_t *move_stride, int64_t *offsets,
                      int32_t *capacity_moves);
/* The same for what mzmcts_moves_inputs unpacks (device-input batches): one block per move with num_legal i32[E],
 * to_play i32[E], legal i32[E][A] at offsets[0..2].
 * Device-input batches alternate between TWO rings of each kind (mzmcts_moves_prepare_device switches): a lock-step
 * move's blocks are downloaded as soon as the move has run (mzmcts_moves_end_lockstep, on a copy stream of the engine),
 * so both calls return the ring of the batch prepared last, and views of a collected batch stay valid while the NEXT
 * batch runs and is collected -- until the mzmcts_moves_prepare_device after that. */
int mzmcts_moves_inputs_ring(mzmcts_engine *engine, void **host_base, int64_t *move_stride, int64_t *offsets);

/* ---- residual-network epilogue (no engine: any device tensor of the current device) ---------
 * What follows every convolution of the reference's residual networks in eval() -- BatchNorm2d with its
 * running statistics, the residual add, ReLU (models.py:215-237 ResidualBlock.forward; 318-335, 467-480) --
 * in one launch:   out = act(x * scale[c] + shift[c] (+ residual)),   c = (index / plane) % channels
 * over a contiguous NCHW tensor of `count` floats (plane = H*W; scale = gamma / sqrt(var + eps),
 * shift = beta - mean * scale, both dev f32[channels]).  residual may be NULL; relu: 0 = identity.
 * x / residual / out must be 16-byte aligned; out must not alias x.  No allocation, no synchronisation. */
int mzmcts_affine_act(const float *x, const float *scale, const float *shift, const float *residual, float *out,
                      int64_t count, int32_t channels, int32_t plane, int32_t relu, void *stream);

/* The CNN down-sampler of the representation network in inference mode (models.py:278-297 DownsampleCNN, called from
 * models.py:318-327): Conv2d(channels, mid, kernel1, stride 4, padding 2) -> ReLU -> MaxPool2d(3, 2) -> Conv2d(mid, cout, 5,
 * padding 2) -> ReLU -> MaxPool2d(3, 2) -> AdaptiveAvgPool2d((out_h, out_w)), one launch, both convolutions on the fp32 matrix
 * cores (exact fp32 products and sums; the order of the sums differs from the convolution library's).
 *   x dev f32[batch, channels, height, width] (16-byte aligned), w1 dev f32[mid, channels, kernel1, kernel1] (16-byte aligned),
 *   b1 dev f32[mid], w2 dev f32[cout, mid, 5, 5], b2 dev f32[cout], out dev f32[batch, cout, out_h, out_w].
 * Covered: 4 x 84 x 84 frames, kernel1 = 12, mid <= 16, cout <= 16, out_h, out_w <= 8 (BASELINE config #5).
 * MZMCTS_ERR_INVALID when the shape is not covered: the caller keeps its convolution library.  No allocation, no
 * synchronisation. */
int mzmcts_downsample_cnn(const float *x, int64_t batch, int32_t channels, int32_t height, int32_t width, const float *w1,
                          const float *b1, int32_t mid, int32_t kernel1, const float *w2, const float *b2, int32_t cout,
                          int32_t out_h, int32_t out_w, float *out, void *stream);

/* The dynamics network's input (models.py:553-568): out[b] = state[b]'s `channels` planes followed by one plane
 * filled with action[b] / action_space.  state dev f32[batch, channels, plane], action dev i64[batch],
 * out dev f32[batch, channels + 1, plane]; batch <= 65535.  One launch for torch's cast, division and cat;
 * the same fp32 division, so bit-identical. */
int mzmcts_state_action_planes(const float *state, const int64_t *action, float *out, int64_t batch,
                               int32_t channels, int32_t plane, int32_t action_space, void *stream);

/* The hidden-state rescale of the residual networks (models.py:525-549, 586-602): every row of `row_len`
 * floats (one (sample, channel) board plane) becomes (x - min) / span, span = max - min (+ 1e-5 when below
 * 1e-5).  x, out: dev f32[rows, row_len] contiguous, out may be x.  row_len <= 128.  One launch instead of
 * torch's seven; same fp32 operations, bit-identical results.  No allocation, no synchronisation. */
int mzmcts_unit_rescale(const float *x, float *out, int64_t rows, int32_t row_len, void *stream);

/* One reward / value / policy head of the residual networks (models.py:467-480, 500-522): 1x1 convolution with
 * bias over the board, flatten, Linear, ELU, Linear -> logits, in one launch.  Pointers are the torch
 * parameters themselves (dev f32, read at every launch, so an in-place weight refresh is seen):
 *   conv_w [reduced, channels], conv_b [reduced], fc1_w [hidden, reduced*plane], fc1_b [hidden],
 *   fc2_w [outputs, hidden], fc2_b [outputs];   x dev f32[batch, channels, plane] (16-byte aligned),
 *   out dev f32[batch, outputs].  MZMCTS_ERR_INVALID when the head does not fit in LDS (the caller keeps the
 * torch modules for that).  fp32, sums in index order: equal to the torch modules to fp32 rounding. */
typedef struct mzmcts_head_desc {
    const float *conv_w, *conv_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
    int32_t channels, plane, reduced, hidden, outputs;
} mzmcts_head_desc;
int mzmcts_conv_head(const float *x, const mzmcts_head_desc *head, float *out, int64_t batch, void *stream);
/* n_heads (1 or 2) heads reading the same x in one launch (value and policy, models.py:500-522): heads[h] writes
 * outs[h] dev f32[batch, heads[h].outputs]; all heads share channels and plane. */
int mzmcts_conv_heads(const float *x, const mzmcts_head_desc *heads, int32_t n_heads, float *const *outs,
                      int64_t batch, void *stream);
/* n_heads (1..3) heads in one launch, head h reading its OWN tensor xs[h] (the reward head reads the dynamics
 * network's raw output, value and policy the prediction network's: models.py:467-480, 500-522); all heads share
 * channels and plane. */
int mzmcts_conv_heads_multi(const float *const *xs, const mzmcts_head_desc *heads, int32_t n_heads, float *const *outs,
                            int64_t batch, void *stream);

/* ---- 3x3 board convolution on the matrix cores, epilogue fused -------------------------------
 * out = act( conv3x3(x, weight; padding 1, stride 1, no bias) * scale[c] + shift[c] (+ residual) ): Conv2d ->
 * BatchNorm2d in eval() -> (+ skip) -> ReLU of the reference's residual networks (models.py:213-229 conv3x3 and
 * ResidualBlock.forward, 318-330, 399-420) in ONE launch.  Exact fp32 on v_mfma_f32_16x16x4_f32: every output is a
 * k-ordered fmaf chain over (tap, input channel), i.e. the arithmetic of a scalar fp32 loop.
 *   x dev f32[batch, cin, height, width] NCHW; out dev f32[batch, cout, height, width] (not x); residual: as out or NULL;
 *   scale / shift dev f32[cout] (gamma / sqrt(var + eps), beta - mean * scale); relu: 0 = identity;
 *   packed: the weight [cout, cin, 3, 3] rearranged k-major by mzmcts_board_conv_pack into
 *   mzmcts_board_conv_packed_floats(cin, cout) floats (repack after every weight change; same buffer, so a
 *   captured hipGraph follows).  Shapes: mzmcts_board_conv_supported(cin, cout, height, width) != 0
 *   (boards 3x3, 6x6, 6x7; cout 16 or 64).  No allocation, no synchronisation. */
int64_t mzmcts_board_conv_packed_floats(int32_t cin, int32_t cout);
int mzmcts_board_conv_pack(const float *weight, float *packed, int32_t cin, int32_t cout, void *stream);
int mzmcts_board_conv_supported(int32_t cin, int32_t cout, int32_t height, int32_t width);
int mzmcts_board_conv3x3(const float *x, const float *packed, const float *scale, const float *shift,
                         const float *residual, float *out, int64_t batch, int32_t cin, int32_t cout, int32_t height,
                         int32_t width, int32_t relu, void *stream);

/* A whole tower of such layers in ONE launch, activations resident in LDS from layer to layer (csrc/board_conv.hip):
 * layer 0 reads x dev f32[batch, cin0, height, width]; every layer produces `channels` planes; `skip` adds the input of
 * the layer BEFORE (the residual block's input, models.py:226-229) ahead of the ReLU.  export_raw / export_unit (either
 * may be NULL) receive the layer's output as NCHW dev f32[batch, channels, height, width], export_unit after the
 * per-plane min-max rescale of models.py:525-549 (which then also replaces the output for the following layers):
 *   dynamics    conv(C+1 -> C), N residual blocks; last layer: export_raw -> reward head, export_unit -> next state
 *   prediction  N residual blocks; last layer: export_raw -> value / policy heads
 * (models.py:399-420, 500-522, 586-602).  n_layers <= 16; shapes as mzmcts_board_conv_supported; MZMCTS_ERR_INVALID
 * when two activation buffers of the workgroup do not fit in LDS (the caller keeps the per-layer path). */
typedef struct mzmcts_tower_layer {
    const void *packed;        /* mzmcts_board_conv_pack (fp32 tower) or mzmcts_board_conv_pack_split (split tower) */
    const float *scale, *shift; /* dev f32[channels] each, 16-byte aligned (the towers read four channels at a time) */
    const float *const_table;  /* split tower, layer 0 with a constant last input plane: its table; else NULL */
    float *export_raw, *export_unit;
    int32_t cin, relu, skip, reserved;
    /* layer 0 only (NULL elsewhere, and NULL = off): overflow hand-over between the two forms of a 64-channel tower,
     * dev i32[mzmcts_board_tower_blocks(batch, channels, height, width) + 1].  The split tower WRITES it: gate[i] = 1 if a
     * value of block i's samples left the fp16 range (|value| * 8 >= 65504) or was not finite, else 0; gate[blocks] counts
     * flagged blocks since the caller last cleared it.  The exact-fp32 tower READS it: of its own workgroups only those
     * holding a sample of a block with gate[i] != 0 run (its workgroups may be larger than the split launch's blocks).
     * Launched back to back on one stream (split, then fp32 with the same layers' fp32 weights and exports) the pair
     * re-computes overflowed samples at full range with no host in between -- also inside a captured hipGraph. */
    int32_t *gate;
} mzmcts_tower_layer;
/* Blocks (workgroups of samples) a tower launch of `batch` samples has; for a 64-channel tower: of the split launch,
 * the unit of the overflow hand-over. */
int64_t mzmcts_board_tower_blocks(int64_t batch, int32_t channels, int32_t height, int32_t width);
int mzmcts_board_tower(const float *x, int64_t batch, int32_t cin0, int32_t channels, int32_t height, int32_t width,
                       const mzmcts_tower_layer *layers, int32_t n_layers, void *stream);
/* A tower launch that also computes reward / value / policy heads (models.py:467-480, 500-522) from the activations while
 * they are in LDS: head h reads the output of layer `layer` (before that layer's export_unit rescale) and writes dev
 * f32[batch, head.outputs] logits -- conv_head_mfma_kernel's arithmetic, the same bits as mzmcts_conv_heads_multi on the
 * exported tensor, without the export, the re-read and the second launch.  Exactly one of x / gather is given.  Covered:
 * 16 channels on 3 x 3 boards (the board-column kernel), reduced <= 16, hidden <= 16, outputs <= 32, at most two heads
 * per layer, heads on the last layer or on an export_unit layer.  MZMCTS_ERR_INVALID when the shape is not covered: the
 * caller launches mzmcts_board_tower[_gathered] and mzmcts_conv_heads_multi instead. */
typedef struct mzmcts_tower_head {
    mzmcts_head_desc head;
    float *out;
    int32_t layer, reserved;
} mzmcts_tower_head;
struct mzmcts_tower_gather;
int mzmcts_board_tower_heads(const float *x, const struct mzmcts_tower_gather *gather, int64_t batch, int32_t cin0,
                             int32_t channels, int32_t height, int32_t width, const mzmcts_tower_layer *layers,
                             int32_t n_layers, const mzmcts_tower_head *heads, int32_t n_heads, void *stream);
/* The same tower (channels == 64) on the 16-bit matrix path at fp32 accuracy: every operand is carried as two fp16
 * halves (22 significant bits), a product as three fp16 MFMAs with fp32 accumulation -- representation error below an
 * fp32 fmaf chain's rounding, 5.3 x fewer matrix-pipe cycles (csrc/board_conv.hip).  Weights come from
 * mzmcts_board_conv_pack_split (mzmcts_board_conv_split_halfs(cin_conv, cout) 16-bit words; cin_conv = cin - 1 when
 * const_plane).  const_plane != 0: the LAST input plane of x is one constant per sample (the dynamics input's action
 * plane, models.py:553-568); it is not convolved, its contribution comes from `const_table` (dev f32[cout, height *
 * width], written by the pack call).  |activation| must stay below 8188 (larger values turn into inf / NaN): with
 * layers[0].gate set the launch reports the blocks where that happened, for the exact-fp32 tower to re-run. */
int64_t mzmcts_board_conv_split_halfs(int32_t cin_conv, int32_t cout);
int mzmcts_board_conv_pack_split(const float *weight, void *packed, float *const_table, int32_t cin, int32_t cout,
                                 int32_t const_plane, int32_t height, int32_t width, void *stream);
int mzmcts_board_tower_split(const float *x, int64_t batch, int32_t cin0, int32_t const_plane, int32_t channels,
                             int32_t height, int32_t width, const mzmcts_tower_layer *layers, int32_t n_layers, void *stream);

/* The dynamics + prediction towers of a simulation step reading their input straight from the search's pools instead of
 * from an [E, channels + 1, height, width] tensor (models.py:553-568 builds it: hidden state of the leaf's parent + one
 * plane of action / action_space; mzmcts_select_planes writes it): sample b takes pool[(parent[b] * envs + b) * hidden_floats
 * ...] for its first `channels` planes and action[b] / action_space for the last.  mzmcts_tower_gather_args fills the
 * descriptor from an engine after mzmcts_select (action = the batch the select wrote, dev i64[E]); split != 0 runs the
 * two-fp16-halves tower (channels == 64), else the exact-fp32 one.  Same results as the tensor form, one launch and one
 * round trip through HBM of the dynamics input less per simulation. */
typedef struct mzmcts_tower_gather {
    const float *pool;         /* hidden-state pool f32[(S+1)][envs][hidden_floats] */
    const int32_t *parent;     /* [envs] slab of the leaf's parent */
    const int64_t *action;     /* [envs] */
    int64_t envs;
    int32_t hidden_floats;     /* channels * height * width */
    float action_space;        /* len(config.action_space) */
} mzmcts_tower_gather;
int mzmcts_tower_gather_args(mzmcts_engine *engine, const int64_t *action, int32_t action_space, mzmcts_tower_gather *out);
int mzmcts_board_tower_gathered(const mzmcts_tower_gather *gather, int64_t batch, int32_t cin0, int32_t split, int32_t channels,
                                int32_t height, int32_t width, const mzmcts_tower_layer *layers, int32_t n_layers, void *stream);

/* ---- measurement ----------------------------------------------------------------------------- */
int mzmcts_set_profiling(mzmcts_engine *engine, int32_t enabled);
/* Trees each wavefront of mzmcts_select works through (self_play.py:321-335 is one descent; the trees are independent,
 * so the order they are descended in changes no result).  0 or 1 = one descent per lane group (default);
 * n > 64 / lanes-per-tree = a wavefront-local queue of n trees whose lane groups pick up the next tree when their
 * descent ends instead of idling behind the deepest one (measured slower at E = 2^20: the kernel is bound by the
 * memory system's request rate). */
int mzmcts_set_select_queue(mzmcts_engine *engine, int32_t trees_per_wavefront);
int mzmcts_get_profile(mzmcts_engine *engine, mzmcts_profile *out, int32_t reset); /* blocking */
/* Bytes of device memory the engine's pools occupy (node blocks, hidden pool, RNG, paths). */
int64_t mzmcts_device_bytes(const mzmcts_engine *engine);

/* ---- exploration noise on the device (numpy.random.dirichlet, self_play.py:468-477) -------------------
 * The legacy gamma sampler goes through libm's log and pow; the library carries glibc's algorithms and tables
 * (csrc/glibc_libm.h) so that device-drawn noise is the reference's to the last bit.  Two blocking self-checks
 * (host arrays in and out):
 *   mzmcts_device_libm       log_out[i] = log(x[i]), pow_out[i] = pow(x[i], y[i]) computed on the GPU, for comparison
 *                            with the host's libm (x >= 0 finite, y > 0);
 *   mzmcts_device_dirichlet  stream s = numpy.random.seed(seeds[s]) followed by `draws` x dirichlet([alpha] * k) on the
 *                            GPU, 0 < alpha <= 1: out f64[n_streams][draws][k], words_out[s] = 32-bit words consumed. */
/* Who draws add_exploration_noise's Dirichlet row of a search: 0 (default) = the host mirror of each stream, inside
 * mzmcts_begin_search (rows returned in noise_out and uploaded); 1 = the GPU, on the device copy of each stream, queued
 * by mzmcts_begin_search right behind its upload (no host work per env; 0 < root_dirichlet_alpha <= 1).  Either way
 * the rows, the search and the stream afterwards are the same to the last bit.  With 1, noise_out is zeros and the
 * rows reach the host with mzmcts_readout: mzmcts_get_noise copies them (f64[E][A], by root child slot). */
int mzmcts_set_device_noise(mzmcts_engine *engine, int32_t enabled);
int mzmcts_get_noise(mzmcts_engine *engine, double *noise_out);
int mzmcts_device_libm(const double *x, const double *y, int64_t n, double *log_out, double *pow_out);
int mzmcts_device_dirichlet(const uint32_t *seeds, int32_t n_streams, double alpha, int32_t k, int32_t draws,
                            double *out, uint32_t *words_out);

/* ---- stand-alone host RNG stream (numpy legacy RandomState clone) ----------------------------
 * The same generator the engine uses per env, exposed for host logic that has no engine
 * (SelfPlay.select_opponent_action's numpy.random.choice, self_play.py:217) and for CPU tests. */
typedef struct mzmcts_rng mzmcts_rng;
mzmcts_rng *mzmcts_rng_create(uint32_t seed);
void mzmcts_rng_destroy(mzmcts_rng *rng);
void mzmcts_rng_reseed(mzmcts_rng *rng, uint32_t seed);
uint32_t mzmcts_rng_next_u32(mzmcts_rng *rng);
double mzmcts_rng_random_sample(mzmcts_rng *rng);
uint32_t mzmcts_rng_choice(mzmcts_rng *rng, uint32_t n);            /* numpy.random.choice(range(n)) */
int32_t mzmcts_rng_choice_p(mzmcts_rng *rng, const double *p, int32_t n); /* choice(n, p=p) */
/* choice(n, size=count, p=p): `count` draws in a row (replay_buffer.py:159 sample_n_games) */
void mzmcts_rng_choice_p_many(mzmcts_rng *rng, const double *p, int32_t n, int32_t count, int32_t *out);
/* ReplayBuffer.sample_position (replay_buffer.py:178-181): probs = priorities / sum(priorities) in float32,
 * then choice(n, p=probs); *prob_out = probs[result] */
int32_t mzmcts_rng_choice_priorities(mzmcts_rng *rng, const float *priorities, int32_t n, float *prob_out);
void mzmcts_rng_dirichlet(mzmcts_rng *rng, double alpha, int32_t k, double *out);
void mzmcts_rng_export(const mzmcts_rng *rng, uint32_t *key, int32_t *pos, int32_t *has_gauss,
                       double *cached_gaussian);
void mzmcts_rng_import(mzmcts_rng *rng, const uint32_t *key, int32_t pos, int32_t has_gauss,
                       double cached_gaussian);
/* select_action on a stand-alone stream: visits host i32[n]; returns the chosen slot. */
int32_t mzmcts_rng_select_action(mzmcts_rng *rng, const int32_t *visits, int32_t n, double temperature);

#ifdef __cplusplus
}
#endif
#endif /* MZMCTS_H */
