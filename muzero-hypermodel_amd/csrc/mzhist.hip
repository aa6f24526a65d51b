// mzhist.hip -- game-history filer (include/mzhist.h): files whole move batches into per-env GameHistory rows on the
// library's worker pool.
#include "engine_host.h"

// ---- game-history filer (include/mzhist.h) ---------------------------------------------------------------------
struct mzhist {
    int E = 0, L = 0, obs = 0, A = 0;
    std::string error;
    // running games, one row per env
    std::vector<float> observations;   // [E][L+1][obs]
    std::vector<int32_t> actions;      // [E][L+1]
    std::vector<float> rewards;        // [E][L+1]
    std::vector<int32_t> to_play;      // [E][L+1]
    std::vector<double> child_visits;  // [E][L][A]
    std::vector<double> root_values;   // [E][L]
    std::vector<int32_t> length;       // [E]
    // games finished by the last mzhist_file
    std::vector<int32_t> fin_env, fin_length, fin_actions, fin_to_play, fin_count, fin_offset;
    std::vector<float> fin_observations, fin_rewards;
    std::vector<double> fin_child_visits, fin_root_values;
    int fin_n = 0, fin_row = 0;
};

extern "C" {

const char* mzhist_last_error(const mzhist* h) { return h ? h->error.c_str() : "mzhist: null handle"; }

int mzhist_create(int32_t num_envs, int32_t max_moves, int32_t obs_floats, int32_t num_actions, mzhist** out) {
    if (!out || num_envs <= 0 || max_moves <= 0 || obs_floats <= 0 || num_actions <= 0) return -1;
    mzhist* h = new mzhist();
    h->E = num_envs;
    h->L = max_moves;
    h->obs = obs_floats;
    h->A = num_actions;
    const size_t E = num_envs, L = max_moves;
    h->observations.assign(E * (L + 1) * obs_floats, 0.f);
    h->actions.assign(E * (L + 1), 0);
    h->rewards.assign(E * (L + 1), 0.f);
    h->to_play.assign(E * (L + 1), 0);
    h->child_visits.assign(E * L * num_actions, 0.0);
    h->root_values.assign(E * L, 0.0);
    h->length.assign(E, 0);
    h->fin_count.assign(E, 0);
    h->fin_offset.assign(E + 1, 0);
    *out = h;
    return 0;
}

void mzhist_destroy(mzhist* h) { delete h; }

const int32_t* mzhist_lengths(const mzhist* h) { return h ? h->length.data() : nullptr; }

int mzhist_begin(mzhist* h, const float* first_observations, const int32_t* first_to_play) {
    if (!h || !first_observations) return -1;
    const size_t L1 = static_cast<size_t>(h->L) + 1;
    for (int e = 0; e < h->E; ++e) {
        std::memcpy(h->observations.data() + static_cast<size_t>(e) * L1 * h->obs,
                    first_observations + static_cast<size_t>(e) * h->obs, sizeof(float) * h->obs);
        h->actions[e * L1] = 0;
        h->rewards[e * L1] = 0.f;
        h->to_play[e * L1] = first_to_play ? first_to_play[e] : 0;
        h->length[e] = 0;
    }
    return 0;
}

int mzhist_rows(mzhist* h, float* observations, int32_t* actions, float* rewards, int32_t* to_play, double* child_visits,
                double* root_values, int32_t* lengths, int32_t load) {
    if (!h || !observations || !actions || !rewards || !to_play || !child_visits || !root_values || !lengths) return -1;
    auto move = [&](auto& mine, auto* theirs) {
        if (load)
            std::memcpy(mine.data(), theirs, sizeof(mine[0]) * mine.size());
        else
            std::memcpy(theirs, mine.data(), sizeof(mine[0]) * mine.size());
    };
    move(h->observations, observations);
    move(h->actions, actions);
    move(h->rewards, rewards);
    move(h->to_play, to_play);
    move(h->child_visits, child_visits);
    move(h->root_values, root_values);
    move(h->length, lengths);
    return 0;
}

int mzhist_file(mzhist* h, const mzhist_moves* mv, int32_t* n_finished) {
    if (!h || !mv || !mv->moves_done || !mv->actions || !mv->visits || !mv->root_value_sum || !mv->legal || !mv->num_legal ||
        !mv->rewards || !mv->done || !mv->obs_after || !mv->obs_next) {
        if (h) h->error = "mzhist_file: null argument";
        return -1;
    }
    const int E = h->E, A = h->A, M = mv->n_moves, obs = h->obs;
    const size_t L = h->L, L1 = L + 1;
    const double S = static_cast<double>(mv->num_simulations);
    auto at = [](const void* base, int64_t stride, int m) { return static_cast<const uint8_t*>(base) + stride * m; };
    WorkerPool& pool = shared_pool();
    auto for_envs = [&](const std::function<void(int, int)>& body) {
        if (pool.size() > 0 && E >= 512)
            pool.run(E, body);
        else
            body(0, E);
    };
    // pass 1: how many games end per env, and how long the longest of them is
    std::atomic<int> longest{0};
    std::atomic<bool> overflow{false}, bad_legal{false};
    for_envs([&](int lo, int hi) {
        int local_longest = 0;
        for (int e = lo; e < hi; ++e) {
            int len = h->length[e], count = 0;
            const int k = std::min(mv->moves_done[e], M);
            for (int m = 0; m < k; ++m) {
                // the legal sets come back from the device: nothing of them is used as an index before it was checked
                const int32_t* legal = reinterpret_cast<const int32_t*>(at(mv->legal, mv->legal_stride, m)) + static_cast<size_t>(e) * A;
                const int n_legal = reinterpret_cast<const int32_t*>(at(mv->num_legal, mv->num_legal_stride, m))[e];
                if (n_legal < 0 || n_legal > A) {
                    bad_legal.store(true);
                } else {
                    for (int i = 0; i < n_legal; ++i)
                        if (legal[i] < 0 || legal[i] >= A) bad_legal.store(true);
                }
                ++len;
                if (len > h->L) overflow.store(true);
                if (mv->done[static_cast<size_t>(m) * E + e]) {
                    ++count;
                    local_longest = std::max(local_longest, len);
                    len = 0;
                }
            }
            h->fin_count[e] = count;
        }
        int seen = longest.load();
        while (local_longest > seen && !longest.compare_exchange_weak(seen, local_longest)) {
        }
    });
    if (bad_legal.load()) {
        h->error = "mzhist_file: a legal-action count outside [0, A] or a legal action outside [0, A)";
        return -1;
    }
    if (overflow.load()) {
        h->error = "mzhist_file: a game outgrew max_moves";
        return -1;
    }
    h->fin_offset[0] = 0;
    for (int e = 0; e < E; ++e) h->fin_offset[e + 1] = h->fin_offset[e] + h->fin_count[e];
    const int n = h->fin_offset[E];
    const size_t W = static_cast<size_t>(longest.load()), W1 = W + 1;
    h->fin_n = n;
    h->fin_row = static_cast<int>(W);
    h->fin_env.resize(n);
    h->fin_length.resize(n);
    h->fin_observations.resize(static_cast<size_t>(n) * W1 * obs);
    h->fin_actions.resize(static_cast<size_t>(n) * W1);
    h->fin_rewards.resize(static_cast<size_t>(n) * W1);
    h->fin_to_play.resize(static_cast<size_t>(n) * W1);
    h->fin_child_visits.resize(static_cast<size_t>(n) * W * A);
    h->fin_root_values.resize(static_cast<size_t>(n) * W);
    // pass 2: append the moves; copy a row out when its game ends and start the next game in place
    for_envs([&](int lo, int hi) {
        for (int e = lo; e < hi; ++e) {
            float* row_obs = h->observations.data() + static_cast<size_t>(e) * L1 * obs;
            int32_t* row_act = h->actions.data() + static_cast<size_t>(e) * L1;
            float* row_rew = h->rewards.data() + static_cast<size_t>(e) * L1;
            int32_t* row_tp = h->to_play.data() + static_cast<size_t>(e) * L1;
            double* row_cv = h->child_visits.data() + static_cast<size_t>(e) * L * A;
            double* row_rv = h->root_values.data() + static_cast<size_t>(e) * L;
            int len = h->length[e];
            int out_slot = h->fin_offset[e];
            const int k = std::min(mv->moves_done[e], M);
            for (int m = 0; m < k; ++m) {
                const size_t me = static_cast<size_t>(m) * E + e;
                const int32_t* visits = reinterpret_cast<const int32_t*>(at(mv->visits, mv->visits_stride, m)) + static_cast<size_t>(e) * A;
                const int32_t* legal = reinterpret_cast<const int32_t*>(at(mv->legal, mv->legal_stride, m)) + static_cast<size_t>(e) * A;
                const int n_legal = reinterpret_cast<const int32_t*>(at(mv->num_legal, mv->num_legal_stride, m))[e];
                double* cv = row_cv + static_cast<size_t>(len) * A;
                for (int a = 0; a < A; ++a) cv[a] = 0.0;
                for (int i = 0; i < n_legal; ++i) cv[legal[i]] = static_cast<double>(visits[i]) / S;
                row_rv[len] = reinterpret_cast<const double*>(at(mv->root_value_sum, mv->root_value_sum_stride, m))[e] / S;
                row_act[len + 1] = reinterpret_cast<const int32_t*>(at(mv->actions, mv->actions_stride, m))[e];
                row_rew[len + 1] = mv->rewards[me];
                std::memcpy(row_obs + static_cast<size_t>(len + 1) * obs, mv->obs_after + me * obs, sizeof(float) * obs);
                row_tp[len + 1] = mv->to_play_after ? mv->to_play_after[me] : 0;
                ++len;
                if (mv->done[me]) {
                    const size_t o = static_cast<size_t>(out_slot);
                    h->fin_env[o] = e;
                    h->fin_length[o] = len;
                    std::memcpy(h->fin_observations.data() + o * W1 * obs, row_obs, sizeof(float) * (len + 1) * obs);
                    std::memcpy(h->fin_actions.data() + o * W1, row_act, sizeof(int32_t) * (len + 1));
                    std::memcpy(h->fin_rewards.data() + o * W1, row_rew, sizeof(float) * (len + 1));
                    std::memcpy(h->fin_to_play.data() + o * W1, row_tp, sizeof(int32_t) * (len + 1));
                    std::memcpy(h->fin_child_visits.data() + o * W * A, row_cv, sizeof(double) * len * A);
                    std::memcpy(h->fin_root_values.data() + o * W, row_rv, sizeof(double) * len);
                    ++out_slot;
                    len = 0;
                    std::memcpy(row_obs, mv->obs_next + me * obs, sizeof(float) * obs);  // the reset observation
                    row_act[0] = 0;
                    row_rew[0] = 0.f;
                    row_tp[0] = mv->to_play_next ? mv->to_play_next[me] : 0;
                }
            }
            h->length[e] = len;
        }
    });
    if (n_finished) *n_finished = n;
    return 0;
}

int mzhist_finished(mzhist* h, const int32_t** env_index, const int32_t** length, const float** observations,
                    const int32_t** actions, const float** rewards, const int32_t** to_play, const double** child_visits,
                    const double** root_values, int32_t* row_moves) {
    if (!h) return -1;
    if (env_index) *env_index = h->fin_env.data();
    if (length) *length = h->fin_length.data();
    if (observations) *observations = h->fin_observations.data();
    if (actions) *actions = h->fin_actions.data();
    if (rewards) *rewards = h->fin_rewards.data();
    if (to_play) *to_play = h->fin_to_play.data();
    if (child_visits) *child_visits = h->fin_child_visits.data();
    if (root_values) *root_values = h->fin_root_values.data();
    if (row_moves) *row_moves = h->fin_row;
    return h->fin_n;
}

}  // extern "C"
