// mzmcts_capi.hip -- host side of libmzmcts.so: the engine part of the C ABI declared in include/mzmcts.h.
//
// Owns the device pools, the per-tree RNG mirrors and the staging buffers; launches the kernels of
// mcts_kernels.hip.  Launch functions do no allocation and no synchronisation (hipGraph-capturable);
// everything blocking says so in the header.  Move batches live in mzmcts_moves.hip, the stand-alone RNG ABI in
// mzmcts_rng.hip, the history filer in mzhist.hip; the shared engine state in engine_host.h.
#include "engine_host.h"

extern "C" {

int mzmcts_abi_version(void) { return MZMCTS_ABI_VERSION; }

const char* mzmcts_last_error(const mzmcts_engine* engine) {
    return engine ? engine->error.c_str() : g_create_error.c_str();
}

int mzmcts_create(const mzmcts_config* c, mzmcts_engine** out) {
    if (!c || !out) return fail(nullptr, MZMCTS_ERR_INVALID, "mzmcts_create: null argument");
    *out = nullptr;
    if (c->num_envs <= 0 || c->num_actions <= 0 || c->num_simulations <= 0 || c->support_size < 0 ||
        c->hidden_floats < 0)
        return fail(nullptr, MZMCTS_ERR_INVALID, "mzmcts_create: sizes must be positive");
    if (c->num_players < 1 || c->num_players > 2)
        return fail(nullptr, MZMCTS_ERR_PLAYERS, "More than two player mode not implemented.");
    if (c->num_actions > 256)
        return fail(nullptr, MZMCTS_ERR_INVALID, "mzmcts_create: at most 256 actions are supported");
    if (c->num_simulations > 32767)
        return fail(nullptr, MZMCTS_ERR_INVALID, "mzmcts_create: at most 32767 simulations are supported");

    int n_dev = 0;
    hipError_t err = hipGetDeviceCount(&n_dev);
    if (err != hipSuccess || n_dev <= 0)
        return fail(nullptr, MZMCTS_ERR_HIP,
                    std::string("mzmcts_create: no HIP device available (") +
                        (err != hipSuccess ? hipGetErrorString(err) : "device count 0") +
                        "); the MCTS engine has no CPU fallback");
    if (c->device < 0 || c->device >= n_dev) return fail(nullptr, MZMCTS_ERR_INVALID, "mzmcts_create: bad device ordinal");
    err = hipSetDevice(c->device);
    if (err != hipSuccess) return hip_fail(nullptr, err, "hipSetDevice");

    auto* eng = new mzmcts_engine();
    eng->cfg = *c;
    const int E = c->num_envs, A = c->num_actions, S = c->num_simulations, H = c->hidden_floats;
    mz::TreeParams& p = eng->p;
    p.E = E;
    p.A = A;
    p.S = S;
    p.P = c->num_players;
    p.support = c->support_size;
    p.F = 2 * c->support_size + 1;
    p.H = H;
    p.chunks = A > 64 ? (A + 63) / 64 : 1;
    p.group = mz::default_group_width(A);
    if (c->group_width == 1 && A == 2) {
        // one lane per tree, the lane loops over the two children: twice the trees (memory requests in flight) per
        // wavefront of the lock-step kernels, whose descents are chains of dependent loads
        p.group = 1;
        p.chunks = 2;
    } else if (c->group_width != 0) {
        const int g = c->group_width;
        // half of pow2(A) lanes (4 or 8) with two children per lane: for action counts just above a power of two (9 of 16
        // lanes busy) the tree kernels run half the wavefronts
        const bool halved = (g == 4 || g == 8) && 2 * g == p.group && A > g;
        if (!halved && (g < p.group || g > 64 || (g & (g - 1)) != 0)) {
            delete eng;
            return fail(nullptr, MZMCTS_ERR_INVALID, "mzmcts_create: group_width must be a power of two in [pow2(A), 64] "
                                                     "(or pow2(A) / 2 = 4 or 8: two children per lane)");
        }
        p.group = g;
        if (halved) p.chunks = 2;
    }
    p.links_offset = 16u * static_cast<uint32_t>(A);
    p.block_stride = mz::round_up(32u * static_cast<uint32_t>(A), 64u);
    // 64-byte blocks (A <= 2) come two to a 128-byte line: a node's first-expanded child shares the node's line
    p.line_stride = p.block_stride == 64u ? 128u : p.block_stride;
    p.discount = c->discount;
    p.noise_frac = c->root_exploration_fraction;
    p.noise_alpha = c->root_dirichlet_alpha;

    int rc = 0;
    auto cleanup_on = [&](int code) {
        if (code != 0) {
            g_create_error = eng->error;
            mzmcts_destroy(eng);
        }
        return code;
    };
    const size_t K = static_cast<size_t>(S) + 1;
    if ((rc = dev_alloc(eng, &p.blocks, K * E * p.line_stride))) return cleanup_on(rc);
    if (c->hidden_pool) {
        p.hidden = static_cast<float*>(c->hidden_pool);
    } else {
        if ((rc = dev_alloc(eng, &p.hidden, K * E * static_cast<size_t>(H), false))) return cleanup_on(rc);
        eng->owns_hidden = true;
    }
    if ((rc = dev_alloc(eng, &p.path, static_cast<size_t>(S) * E))) return cleanup_on(rc);
    p.path_ties = nullptr;
    if ((rc = dev_alloc(eng, &p.path_len, E))) return cleanup_on(rc);
    if ((rc = dev_alloc(eng, &p.leaf_parent, E))) return cleanup_on(rc);
    if ((rc = dev_alloc(eng, &p.leaf_loc, E))) return cleanup_on(rc);
    if ((rc = dev_alloc(eng, &p.root_reward, E))) return cleanup_on(rc);
    {
        // per-move upload block: [legal i32 E*A | num_legal i32 E | to_play i32 E | rng_skip u32 E | noise f64 E*A]
        auto align8 = [](size_t v) { return (v + 7) / 8 * 8; };
        const size_t o_legal = 0;
        const size_t o_nlegal = o_legal + sizeof(int32_t) * static_cast<size_t>(E) * A;
        const size_t o_to_play = o_nlegal + sizeof(int32_t) * E;
        const size_t o_skip = o_to_play + sizeof(int32_t) * E;
        const size_t o_noise = align8(o_skip + sizeof(uint32_t) * E);
        eng->upload_bytes_no_noise = o_noise;
        eng->upload_bytes = o_noise + sizeof(double) * static_cast<size_t>(E) * A;
        if ((rc = dev_alloc(eng, &eng->d_upload, eng->upload_bytes))) return cleanup_on(rc);
        if ((rc = pinned_alloc(eng, &eng->h_upload, eng->upload_bytes))) return cleanup_on(rc);
        p.root_action = reinterpret_cast<int32_t*>(eng->d_upload + o_legal);
        p.root_children = reinterpret_cast<int32_t*>(eng->d_upload + o_nlegal);
        p.root_to_play = reinterpret_cast<int32_t*>(eng->d_upload + o_to_play);
        eng->d_skip = reinterpret_cast<uint32_t*>(eng->d_upload + o_skip);
        eng->d_noise = reinterpret_cast<double*>(eng->d_upload + o_noise);
        p.noise_rows = eng->d_noise;
        eng->h_legal = reinterpret_cast<int32_t*>(eng->h_upload + o_legal);
        eng->h_nlegal = reinterpret_cast<int32_t*>(eng->h_upload + o_nlegal);
        eng->h_to_play = reinterpret_cast<int32_t*>(eng->h_upload + o_to_play);
        eng->h_skip = reinterpret_cast<uint32_t*>(eng->h_upload + o_skip);
        eng->h_noise = reinterpret_cast<double*>(eng->h_upload + o_noise);
        // per-tree download block: [root_value_sum f64 E | min_max 2xf64 E | depth_sum i64 E |
        //                           root_predicted f32 E | max_depth i32 E | tie_words u32 E | error_flag i32 x4]
        const size_t o_rvs = 0;
        const size_t o_mm = o_rvs + sizeof(double) * E;
        const size_t o_ds = o_mm + sizeof(mz::MinMax) * E;
        const size_t o_pred = o_ds + sizeof(int64_t) * E;
        const size_t o_md = o_pred + sizeof(float) * E;
        const size_t o_tw = o_md + sizeof(int32_t) * E;
        const size_t o_nw = o_tw + sizeof(uint32_t) * E;
        const size_t o_err = o_nw + sizeof(uint32_t) * E;
        eng->download_bytes = o_err + sizeof(int32_t) * 4;
        if ((rc = dev_alloc(eng, &eng->d_download, eng->download_bytes))) return cleanup_on(rc);
        if ((rc = pinned_alloc(eng, &eng->h_download, eng->download_bytes))) return cleanup_on(rc);
        p.root_value_sum = reinterpret_cast<double*>(eng->d_download + o_rvs);
        p.min_max = reinterpret_cast<mz::MinMax*>(eng->d_download + o_mm);
        p.depth_sum = reinterpret_cast<int64_t*>(eng->d_download + o_ds);
        p.root_predicted = reinterpret_cast<float*>(eng->d_download + o_pred);
        p.max_depth = reinterpret_cast<int32_t*>(eng->d_download + o_md);
        p.tie_words = reinterpret_cast<uint32_t*>(eng->d_download + o_tw);
        p.noise_words = reinterpret_cast<uint32_t*>(eng->d_download + o_nw);
        p.error_flag = reinterpret_cast<int32_t*>(eng->d_download + o_err);
        eng->h_root_value_sum = reinterpret_cast<double*>(eng->h_download + o_rvs);
        eng->h_min_max = reinterpret_cast<mz::MinMax*>(eng->h_download + o_mm);
        eng->h_depth_sum = reinterpret_cast<int64_t*>(eng->h_download + o_ds);
        eng->h_root_predicted = reinterpret_cast<float*>(eng->h_download + o_pred);
        eng->h_max_depth = reinterpret_cast<int32_t*>(eng->h_download + o_md);
        eng->h_tie_words = reinterpret_cast<uint32_t*>(eng->h_download + o_tw);
        eng->h_noise_words = reinterpret_cast<uint32_t*>(eng->h_download + o_nw);
        eng->h_error_flag = reinterpret_cast<int32_t*>(eng->h_download + o_err);
    }
    if ((rc = dev_alloc(eng, &p.mt_key, static_cast<size_t>(E) * mz::kMtN))) return cleanup_on(rc);
    if ((rc = dev_alloc(eng, &p.mt_pos, E))) return cleanup_on(rc);
    if ((rc = dev_alloc(eng, &eng->d_seeds, E))) return cleanup_on(rc);

    // pb_c tables: host libm, the same log/sqrt Python's math module calls (self_play.py:385-391)
    {
        std::vector<double> tab(2 * K);
        for (size_t n = 0; n < K; ++n) {
            tab[n] = std::log((static_cast<double>(n) + c->pb_c_base + 1) / c->pb_c_base) + c->pb_c_init;
            tab[K + n] = std::sqrt(static_cast<double>(n));
        }
        double* d_tab = nullptr;
        if ((rc = dev_alloc(eng, &d_tab, 2 * K))) return cleanup_on(rc);
        err = hipMemcpy(d_tab, tab.data(), sizeof(double) * 2 * K, hipMemcpyHostToDevice);
        if (err != hipSuccess) return cleanup_on(hip_fail(eng, err, "hipMemcpy(pb_c table)"));
        p.pbc_log = d_tab;
        p.pbc_sqrt = d_tab + K;
    }

    if ((rc = pinned_alloc(eng, &eng->h_slab0, static_cast<size_t>(E) * p.line_stride))) return cleanup_on(rc);

    eng->streams.resize(E);
    eng->behind.assign(E, 0);
    eng->lag.assign(E, 0u);
    eng->last_visits.assign(static_cast<size_t>(E) * A, 0);
    eng->last_root_value_sum.assign(E, 0.0);
    eng->last_root_visits.assign(E, 0);
    // default seeding: stream e == numpy.random.seed(e)
    std::vector<uint32_t> seeds(E);
    for (int e = 0; e < E; ++e) seeds[e] = static_cast<uint32_t>(e);
    *out = eng;
    rc = mzmcts_seed(eng, seeds.data(), nullptr);
    if (rc != 0) {
        *out = nullptr;
        return cleanup_on(rc);
    }
    return MZMCTS_OK;
}

void mzmcts_destroy(mzmcts_engine* eng) {
    if (!eng) return;
    (void)hipSetDevice(eng->cfg.device);
    (void)hipDeviceSynchronize();
    for (auto& ev : eng->events) {
        (void)hipEventDestroy(ev.begin);
        (void)hipEventDestroy(ev.end);
    }
    if (eng->readout_event) (void)hipEventDestroy(eng->readout_event);
    if (eng->batch.done) (void)hipEventDestroy(eng->batch.done);
    if (eng->batch.move_done) (void)hipEventDestroy(eng->batch.move_done);
    if (eng->batch.copy_stream) {
        (void)hipStreamSynchronize(eng->batch.copy_stream);
        (void)hipStreamDestroy(eng->batch.copy_stream);
    }
    for (void* ptr : eng->device_allocs) (void)hipFree(ptr);
    for (void* ptr : eng->pinned_allocs) (void)hipHostFree(ptr);
    delete eng;
}

int mzmcts_seed(mzmcts_engine* eng, const uint32_t* seeds, void* stream_) {
    if (!eng || !seeds) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_seed: null argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int E = eng->p.E;
    eng->for_each_env([&](int lo, int hi) {
        for (int e = lo; e < hi; ++e) {
            eng->streams[e].seed(seeds[e]);
            eng->behind[e] = 0;
        }
    });
    std::fill(eng->lag.begin(), eng->lag.end(), 0u);
    MZ_HIP(eng, hipMemcpyAsync(eng->d_seeds, seeds, sizeof(uint32_t) * E, hipMemcpyHostToDevice, stream));
    MZ_HIP(eng, mz::launch_seed_streams(eng->p.mt_key, eng->p.mt_pos, eng->d_seeds, E, stream));
    MZ_HIP(eng, hipStreamSynchronize(stream));
    return MZMCTS_OK;
}

int mzmcts_rng_set_state(mzmcts_engine* eng, int32_t env, const uint32_t* key, int32_t pos, int32_t has_gauss,
                         double cached, void* stream_) {
    if (!eng || !key || env < 0 || env >= eng->p.E || pos < 0 || pos > mz::kMtN)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_rng_set_state: bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    eng->behind[env] = 0;                              // (the state replaces whatever the mirror still owed)
    mz::HostStream& s = eng->streams[env];
    std::memcpy(s.key, key, sizeof(s.key));
    s.pos = pos;
    s.has_gauss = has_gauss;
    s.gauss = cached;
    eng->lag[env] = 0;
    MZ_HIP(eng, hipMemcpyAsync(eng->p.mt_key + static_cast<size_t>(env) * mz::kMtN, s.key, sizeof(s.key),
                               hipMemcpyHostToDevice, stream));
    MZ_HIP(eng, hipMemcpyAsync(eng->p.mt_pos + env, &s.pos, sizeof(int32_t), hipMemcpyHostToDevice, stream));
    MZ_HIP(eng, hipStreamSynchronize(stream));
    return MZMCTS_OK;
}

int mzmcts_rng_get_state(mzmcts_engine* eng, int32_t env, uint32_t* key, int32_t* pos, int32_t* has_gauss,
                         double* cached, void* stream_) {
    if (!eng || !key || !pos || env < 0 || env >= eng->p.E)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_rng_get_state: bad argument");
    (void)stream_;
    const mz::HostStream& s = eng->mirror(env);  // the host mirror is authoritative between calls
    std::memcpy(key, s.key, sizeof(s.key));
    *pos = s.pos;
    if (has_gauss) *has_gauss = s.has_gauss;
    if (cached) *cached = s.gauss;
    return MZMCTS_OK;
}

int mzmcts_begin_search(mzmcts_engine* eng, const int32_t* legal, const int32_t* num_legal, const int32_t* to_play,
                        int32_t add_noise, double* noise_out, void* stream_) {
    if (!eng || !legal || !num_legal || !to_play)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_begin_search: null argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int E = eng->p.E, A = eng->p.A;
    // plugin contract (self_play.py:297-302)
    for (int e = 0; e < E; ++e) {
        const int n = num_legal[e];
        if (n < 0 || n > A)
            return fail(eng, MZMCTS_ERR_LEGAL_RANGE, "Legal actions should be a subset of the action space.");
        for (int i = 0; i < n; ++i) {
            const int a = legal[static_cast<size_t>(e) * A + i];
            if (a < 0 || a >= A)
                return fail(eng, MZMCTS_ERR_LEGAL_RANGE, "Legal actions should be a subset of the action space.");
        }
    }
    const bool trace = std::getenv("MZMCTS_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = trace ? now() : 0.0;
    std::memcpy(eng->h_legal, legal, sizeof(int32_t) * static_cast<size_t>(E) * A);
    std::memcpy(eng->h_nlegal, num_legal, sizeof(int32_t) * E);
    std::memcpy(eng->h_to_play, to_play, sizeof(int32_t) * E);
    const double alpha = eng->cfg.root_dirichlet_alpha;
    const bool on_device = add_noise && eng->device_noise;   // root_noise_kernel draws the rows, below
    eng->for_each_env([&](int lo, int hi) {
        for (int e = lo; e < hi; ++e) {
            const int n = eng->h_nlegal[e];
            double* row = eng->h_noise + static_cast<size_t>(e) * A;
            for (int i = 0; i < A; ++i) row[i] = 0.0;
            if (n == 0) {
                eng->h_skip[e] = 0;  // inactive: the stream is left alone, pending lag is kept
                continue;
            }
            if (add_noise && !on_device) {
                mz::HostStream& s = eng->mirror(e);
                const uint64_t before = s.words;
                s.dirichlet(alpha, n, row);
                eng->lag[e] += static_cast<uint32_t>(s.words - before);
            }
            eng->h_skip[e] = eng->lag[e];
            eng->lag[e] = 0;
        }
    });
    const double t_dirichlet = trace ? now() : 0.0;
    if (noise_out) std::memcpy(noise_out, eng->h_noise, sizeof(double) * static_cast<size_t>(E) * A);
    eng->noise_this_search = add_noise != 0;
    MZ_HIP(eng, hipMemcpyAsync(eng->d_upload, eng->h_upload,
                               (add_noise && !on_device) ? eng->upload_bytes : eng->upload_bytes_no_noise,
                               hipMemcpyHostToDevice, stream));
    if (on_device) MZ_HIP(eng, mz::launch_root_noise(eng->p, eng->d_skip, stream));
    eng->noise_on_device = on_device;
    if (trace)
        std::fprintf(stderr, "[mzmcts] begin_search E=%d: host prep+dirichlet %.1f us, memcpy+enqueue %.1f us\n", E,
                     t_dirichlet - t_start, now() - t_dirichlet);
    eng->search_begun = true;
    eng->skip_applied = false;
    eng->tie_words_applied = false;
    eng->roots_ready = false;
    eng->have_readout = false;
    eng->sim = 0;
    return MZMCTS_OK;
}

static int expand_roots_common(mzmcts_engine* eng, const float* value_logits, const float* reward_logits,
                               const float* policy_logits, const float* root_hidden, const double* inj_reward,
                               const double* inj_priors, bool injected, hipStream_t stream) {
    if (!eng->search_begun) return fail(eng, MZMCTS_ERR_INVALID, "expand_roots called before begin_search");
    {
        ProfScope scope(eng, stream, kProfRoot);
        MZ_HIP(eng, mz::launch_expand_roots(eng->p, value_logits, reward_logits, policy_logits, root_hidden, inj_reward,
                                            inj_priors, eng->noise_this_search ? eng->d_noise : nullptr,
                                            eng->skip_applied ? nullptr : eng->d_skip, injected, stream, scope.get()));
    }
    eng->roots_ready = true;
    eng->tree_published = true;
    eng->sim = 0;
    return MZMCTS_OK;
}

int mzmcts_expand_roots(mzmcts_engine* eng, const float* value_logits, const float* reward_logits,
                        const float* policy_logits, const float* root_hidden, void* stream) {
    if (!eng || !value_logits || !policy_logits)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_expand_roots: null argument");
    return expand_roots_common(eng, value_logits, reward_logits, policy_logits, root_hidden, nullptr, nullptr, false,
                               static_cast<hipStream_t>(stream));
}

int mzmcts_expand_roots_injected(mzmcts_engine* eng, const double* root_reward, const double* root_priors, void* stream) {
    if (!eng || !root_reward || !root_priors)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_expand_roots_injected: null argument");
    return expand_roots_common(eng, nullptr, nullptr, nullptr, nullptr, root_reward, root_priors, true,
                               static_cast<hipStream_t>(stream));
}

int mzmcts_select(mzmcts_engine* eng, float* parent_hidden_out, int64_t* action_out, void* stream_) {
    if (!eng) return MZMCTS_ERR_INVALID;
    if (!eng->roots_ready) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_select called before expand_roots");
    if (eng->sim >= eng->p.S) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_select: all simulations already ran");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ProfScope scope(eng, stream, kProfSelect);
    MZ_HIP(eng, mz::launch_select(eng->p, eng->sim, eng->p.H > 0 ? parent_hidden_out : nullptr, action_out,
                                  eng->select_queue_trees, stream, scope.get()));
    return MZMCTS_OK;
}

int mzmcts_tower_gather_args(mzmcts_engine* eng, const int64_t* action, int32_t action_space, mzmcts_tower_gather* out) {
    if (!eng || !action || !out || action_space <= 0) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_tower_gather_args: bad argument");
    out->pool = eng->p.hidden;
    out->parent = eng->p.leaf_parent;
    out->action = action;
    out->envs = eng->p.E;
    out->hidden_floats = eng->p.H;
    out->action_space = static_cast<float>(action_space);
    return MZMCTS_OK;
}

int mzmcts_select_planes(mzmcts_engine* eng, float* planes_out, int64_t* action_out, int32_t plane, int32_t action_space,
                         void* stream_) {
    if (!eng || !planes_out || !action_out) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_select_planes: null argument");
    if (plane <= 0 || action_space <= 0 || eng->p.H <= 0 || eng->p.H % plane != 0)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_select_planes: hidden_floats must be channels x plane");
    int rc = mzmcts_select(eng, nullptr, action_out, stream_);
    if (rc) return rc;
    MZ_HIP(eng, mz::launch_gather_dynamics_input(eng->p, action_out, planes_out, plane, action_space,
                                                 static_cast<hipStream_t>(stream_)));
    return MZMCTS_OK;
}

int mzmcts_expand_backup(mzmcts_engine* eng, const float* value_logits, const float* reward_logits,
                         const float* policy_logits, const float* next_hidden, void* stream_) {
    if (!eng || !value_logits || !reward_logits || !policy_logits)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_expand_backup: null argument");
    if (!eng->roots_ready || eng->sim >= eng->p.S)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_expand_backup: no simulation in flight");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (next_hidden && eng->p.H > 0) {
        float* slab = eng->p.hidden + (static_cast<size_t>(eng->sim) + 1) * eng->p.E * eng->p.H;
        MZ_HIP(eng, mz::launch_copy_slab(next_hidden, slab, static_cast<size_t>(eng->p.E) * eng->p.H, stream));
    }
    {
        ProfScope scope(eng, stream, kProfBackup);
        MZ_HIP(eng, mz::launch_expand_backup(eng->p, eng->sim, value_logits, reward_logits, policy_logits, nullptr,
                                             nullptr, nullptr, false, stream, scope.get()));
    }
    eng->sim += 1;
    return MZMCTS_OK;
}

int mzmcts_expand_backup_injected(mzmcts_engine* eng, const double* value, const double* reward, const double* priors,
                                  void* stream_) {
    if (!eng || !value || !reward || !priors)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_expand_backup_injected: null argument");
    if (!eng->roots_ready || eng->sim >= eng->p.S)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_expand_backup_injected: no simulation in flight");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    {
        ProfScope scope(eng, stream, kProfBackup);
        MZ_HIP(eng, mz::launch_expand_backup(eng->p, eng->sim, nullptr, nullptr, nullptr, value, reward, priors, true,
                                             stream, scope.get()));
    }
    eng->sim += 1;
    return MZMCTS_OK;
}

// expand_backup of the simulation in flight + select of the next one, one launch (mcts_kernels.hip)
static int step_common(mzmcts_engine* eng, const float* value_logits, const float* reward_logits, const float* policy_logits,
                       const float* next_hidden, const double* inj_value, const double* inj_reward, const double* inj_priors,
                       bool injected, float* parent_hidden_out, int64_t* action_out, hipStream_t stream, const char* who) {
    if (!eng->roots_ready || eng->sim >= eng->p.S) return fail(eng, MZMCTS_ERR_INVALID, std::string(who) + ": no simulation in flight");
    if (eng->sim + 1 >= eng->p.S)
        return fail(eng, MZMCTS_ERR_INVALID, std::string(who) + ": this is the search's last simulation (use mzmcts_expand_backup)");
    if (next_hidden && eng->p.H > 0) {
        float* slab = eng->p.hidden + (static_cast<size_t>(eng->sim) + 1) * eng->p.E * eng->p.H;
        MZ_HIP(eng, mz::launch_copy_slab(next_hidden, slab, static_cast<size_t>(eng->p.E) * eng->p.H, stream));
    }
    {
        ProfScope scope(eng, stream, kProfStep);
        MZ_HIP(eng, mz::launch_expand_backup_select(eng->p, eng->sim, value_logits, reward_logits, policy_logits, inj_value,
                                                    inj_reward, inj_priors, injected,
                                                    eng->p.H > 0 ? parent_hidden_out : nullptr, action_out, stream, scope.get()));
    }
    eng->sim += 1;
    return MZMCTS_OK;
}

int mzmcts_expand_backup_select(mzmcts_engine* eng, const float* value_logits, const float* reward_logits,
                                const float* policy_logits, const float* next_hidden, float* parent_hidden_out,
                                int64_t* action_out, void* stream_) {
    if (!eng || !value_logits || !reward_logits || !policy_logits)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_expand_backup_select: null argument");
    return step_common(eng, value_logits, reward_logits, policy_logits, next_hidden, nullptr, nullptr, nullptr, false,
                       parent_hidden_out, action_out, static_cast<hipStream_t>(stream_), "mzmcts_expand_backup_select");
}

int mzmcts_expand_backup_select_planes(mzmcts_engine* eng, const float* value_logits, const float* reward_logits,
                                       const float* policy_logits, const float* next_hidden, float* planes_out,
                                       int64_t* action_out, int32_t plane, int32_t action_space, void* stream_) {
    if (!eng || !value_logits || !reward_logits || !policy_logits || !planes_out || !action_out)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_expand_backup_select_planes: null argument");
    if (plane <= 0 || action_space <= 0 || eng->p.H <= 0 || eng->p.H % plane != 0)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_expand_backup_select_planes: hidden_floats must be channels x plane");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc = step_common(eng, value_logits, reward_logits, policy_logits, next_hidden, nullptr, nullptr, nullptr, false, nullptr,
                         action_out, stream, "mzmcts_expand_backup_select_planes");
    if (rc) return rc;
    MZ_HIP(eng, mz::launch_gather_dynamics_input(eng->p, action_out, planes_out, plane, action_space, stream));
    return MZMCTS_OK;
}

int mzmcts_expand_backup_select_injected(mzmcts_engine* eng, const double* value, const double* reward, const double* priors,
                                         float* parent_hidden_out, int64_t* action_out, void* stream_) {
    if (!eng || !value || !reward || !priors)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_expand_backup_select_injected: null argument");
    return step_common(eng, nullptr, nullptr, nullptr, nullptr, value, reward, priors, true, parent_hidden_out, action_out,
                       static_cast<hipStream_t>(stream_), "mzmcts_expand_backup_select_injected");
}

float* mzmcts_hidden_slab(mzmcts_engine* eng, int32_t slab) {
    if (!eng || slab < 0 || slab > eng->p.S || eng->p.H == 0) return nullptr;
    return eng->p.hidden + static_cast<size_t>(slab) * eng->p.E * eng->p.H;
}

int32_t mzmcts_next_slab(const mzmcts_engine* eng) { return eng ? eng->sim + 1 : -1; }
int32_t mzmcts_simulations_done(const mzmcts_engine* eng) { return eng ? eng->sim : -1; }

int mzmcts_set_simulations_done(mzmcts_engine* eng, int32_t n) {
    if (!eng || n < 0 || n > eng->p.S) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_set_simulations_done: out of range");
    eng->sim = n;
    return MZMCTS_OK;
}

static int enqueue_readout_copies(mzmcts_engine* eng, hipStream_t stream) {
    const mz::TreeParams& p = eng->p;
    MZ_HIP(eng, hipMemcpyAsync(eng->h_slab0, p.blocks, static_cast<size_t>(p.E) * p.line_stride, hipMemcpyDeviceToHost, stream));
    MZ_HIP(eng, hipMemcpyAsync(eng->h_download, eng->d_download, eng->download_bytes, hipMemcpyDeviceToHost, stream));
    if (eng->noise_on_device)   // the rows the device drew (what mzmcts_begin_search's noise_out holds for a host draw)
        MZ_HIP(eng, hipMemcpyAsync(eng->h_noise, eng->d_noise, sizeof(double) * static_cast<size_t>(p.E) * p.A,
                                   hipMemcpyDeviceToHost, stream));
    return MZMCTS_OK;
}

int mzmcts_readout_begin(mzmcts_engine* eng, void* stream_) {
    if (!eng) return MZMCTS_ERR_INVALID;
    if (!eng->roots_ready) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_readout_begin called before expand_roots");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!eng->readout_event) MZ_HIP(eng, hipEventCreateWithFlags(&eng->readout_event, hipEventDisableTiming));
    int rc = enqueue_readout_copies(eng, stream);
    if (rc) return rc;
    MZ_HIP(eng, hipEventRecord(eng->readout_event, stream));
    eng->readout_pending = true;
    return MZMCTS_OK;
}

int mzmcts_readout(mzmcts_engine* eng, const mzmcts_root_stats* out, void* stream_) {
    if (!eng) return MZMCTS_ERR_INVALID;
    if (!eng->roots_ready) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_readout called before expand_roots");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const mz::TreeParams& p = eng->p;
    const int E = p.E, A = p.A;
    if (eng->readout_pending) {
        MZ_HIP(eng, hipEventSynchronize(eng->readout_event));
        eng->readout_pending = false;
    } else {
        int rc = enqueue_readout_copies(eng, stream);
        if (rc) return rc;
        MZ_HIP(eng, hipStreamSynchronize(stream));
    }
    if (eng->h_error_flag[0] != 0)
        return fail(eng, MZMCTS_ERR_INVALID,
                    (eng->h_error_flag[0] & 2) ? "device error flag set: tree links are inconsistent (descent ran past the "
                                                 "number of expanded nodes)"
                                               : "device error flag set: a UCB score was NaN (no maximum to select)");

    const int sims = eng->sim;
    const bool first = !eng->tie_words_applied;
    std::atomic<int64_t> depth_total{0}, active{0};
    eng->for_each_env([&](int lo, int hi) {
        int64_t local_depth = 0, local_active = 0;
        for (int e = lo; e < hi; ++e) {
            const uint8_t* blk = eng->h_slab0 + static_cast<size_t>(e) * p.line_stride;   // the root's block: half 0 of line (0, e)
            const mz::HbmChild* rec = reinterpret_cast<const mz::HbmChild*>(blk);
            const int n = eng->h_nlegal[e];
            const bool is_active = n > 0;
            for (int i = 0; i < A; ++i) {
                const size_t o = static_cast<size_t>(e) * A + i;
                const bool live = i < n;
                eng->last_visits[o] = live ? rec[i].links.visits : 0;
                if (out) {
                    if (out->visits) out->visits[o] = live ? rec[i].links.visits : 0;
                    if (out->child_value_sum) out->child_value_sum[o] = live ? rec[i].stats.value_sum : 0.0;
                    if (out->child_prior) out->child_prior[o] = live ? rec[i].stats.prior : 0.0;
                    if (out->child_reward) out->child_reward[o] = live ? static_cast<double>(rec[i].links.reward) : 0.0;
                    if (out->child_expanded) out->child_expanded[o] = (live && rec[i].links.child_node >= 0) ? 1 : 0;
                }
            }
            eng->last_root_value_sum[e] = is_active ? eng->h_root_value_sum[e] : 0.0;
            eng->last_root_visits[e] = is_active ? sims : 0;
            if (is_active) {
                // the tie-breaks ran on the device copy of the stream: bring the host mirror level (once)
                // (and, when the device drew this search's exploration noise, the Dirichlet draw before them)
                if (first) eng->mirror(e).skip((eng->noise_on_device ? eng->h_noise_words[e] : 0u) + eng->h_tie_words[e]);
                local_depth += eng->h_depth_sum[e];
                ++local_active;
            }
            if (out) {
                if (out->root_value_sum) out->root_value_sum[e] = eng->last_root_value_sum[e];
                if (out->root_visits) out->root_visits[e] = eng->last_root_visits[e];
                if (out->max_tree_depth) out->max_tree_depth[e] = is_active ? eng->h_max_depth[e] : 0;
                if (out->root_predicted_value) out->root_predicted_value[e] = static_cast<double>(eng->h_root_predicted[e]);
                if (out->min_max) {
                    out->min_max[2 * e] = eng->h_min_max[e].minimum;
                    out->min_max[2 * e + 1] = eng->h_min_max[e].maximum;
                }
                if (out->depth_sum) out->depth_sum[e] = is_active ? eng->h_depth_sum[e] : 0;
                if (out->tie_break_words) out->tie_break_words[e] = is_active ? eng->h_tie_words[e] : 0u;
            }
        }
        depth_total.fetch_add(local_depth, std::memory_order_relaxed);
        active.fetch_add(local_active, std::memory_order_relaxed);
    });
    if (first) {
        eng->prof.select_depth_sum += depth_total.load();
        eng->prof.simulations += active.load() * sims;
    }
    eng->tie_words_applied = true;
    eng->have_readout = true;
    return MZMCTS_OK;
}

int mzmcts_sample_actions(mzmcts_engine* eng, const double* temperature, int32_t* action_out, int32_t* slot_out) {
    if (!eng || !temperature || !action_out)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_sample_actions: null argument");
    if (!eng->have_readout) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_sample_actions called before readout");
    const int E = eng->p.E, A = eng->p.A;
    eng->for_each_env([&](int lo, int hi) {
        for (int e = lo; e < hi; ++e) {
            const int n = eng->h_nlegal[e];
            if (n == 0) {
                action_out[e] = -1;
                if (slot_out) slot_out[e] = -1;
                continue;
            }
            mz::HostStream& s = eng->mirror(e);
            const uint64_t before = s.words;
            const int slot = s.select_action(eng->last_visits.data() + static_cast<size_t>(e) * A, n, temperature[e]);
            eng->lag[e] += static_cast<uint32_t>(s.words - before);
            action_out[e] = eng->h_legal[static_cast<size_t>(e) * A + slot];
            if (slot_out) slot_out[e] = slot;
        }
    });
    return MZMCTS_OK;
}

int mzmcts_search_statistics(mzmcts_engine* eng, double* child_visits, double* root_values) {
    if (!eng || !child_visits || !root_values)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_search_statistics: null argument");
    if (!eng->have_readout) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_search_statistics called before readout");
    const int E = eng->p.E, A = eng->p.A;
    for (int e = 0; e < E; ++e) {
        double* row = child_visits + static_cast<size_t>(e) * A;
        for (int a = 0; a < A; ++a) row[a] = 0.0;
        const int n = eng->h_nlegal[e];
        int total = 0;
        for (int i = 0; i < n; ++i) total += eng->last_visits[static_cast<size_t>(e) * A + i];
        for (int i = 0; i < n; ++i)
            row[eng->h_legal[static_cast<size_t>(e) * A + i]] =
                static_cast<double>(eng->last_visits[static_cast<size_t>(e) * A + i]) / total;
        const int rv = eng->last_root_visits[e];
        root_values[e] = rv == 0 ? 0.0 : eng->last_root_value_sum[e] / rv;
    }
    return MZMCTS_OK;
}

int mzmcts_set_debug_ties(mzmcts_engine* eng, int32_t enabled) {
    if (!eng) return MZMCTS_ERR_INVALID;
    if (enabled && !eng->p.path_ties) {
        int rc = dev_alloc(eng, &eng->p.path_ties, static_cast<size_t>(eng->p.S) * eng->p.E);
        if (rc) return rc;
    }
    return MZMCTS_OK;
}

int mzmcts_last_paths(mzmcts_engine* eng, int32_t* depth, int32_t* actions, int32_t* tie_counts, void* stream_) {
    if (!eng || !depth) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_last_paths: null argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const mz::TreeParams& p = eng->p;
    const int E = p.E, S = p.S, A = p.A;
    std::vector<int32_t> path(static_cast<size_t>(S) * E), ties;
    MZ_HIP(eng, hipMemcpyAsync(depth, p.path_len, sizeof(int32_t) * E, hipMemcpyDeviceToHost, stream));
    MZ_HIP(eng, hipMemcpyAsync(path.data(), p.path, sizeof(int32_t) * path.size(), hipMemcpyDeviceToHost, stream));
    if (tie_counts && p.path_ties) {
        ties.resize(path.size());
        MZ_HIP(eng, hipMemcpyAsync(ties.data(), p.path_ties, sizeof(int32_t) * ties.size(), hipMemcpyDeviceToHost, stream));
    }
    MZ_HIP(eng, hipStreamSynchronize(stream));
    for (int e = 0; e < E; ++e) {
        for (int d = 0; d < S; ++d) {
            const size_t o = static_cast<size_t>(e) * S + d;
            if (d < depth[e]) {
                const int slot = path[static_cast<size_t>(d) * E + e] & 0xff;
                if (actions) actions[o] = (d == 0) ? eng->h_legal[static_cast<size_t>(e) * A + slot] : slot;
                if (tie_counts) tie_counts[o] = ties.empty() ? -1 : ties[static_cast<size_t>(d) * E + e];
            } else {
                if (actions) actions[o] = -1;
                if (tie_counts) tie_counts[o] = 0;
            }
        }
    }
    return MZMCTS_OK;
}

int mzmcts_export_tree(mzmcts_engine* eng, int32_t env, int32_t* visits, double* value_sum, double* prior,
                       double* reward, int32_t* child_node, void* stream_) {
    if (!eng || env < 0 || env >= eng->p.E) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_export_tree: bad env");
    if (!eng->tree_published)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_export_tree: the last fused search published the root only "
                                             "(mzmcts_set_fused_options(engine, variant, publish_tree = 1) keeps the tree)");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const mz::TreeParams& p = eng->p;
    const int K = p.S + 1, A = p.A;
    // the tree's lines (one per slab); node k's block sits where its parent's link says (block_loc = 2 * slab + half)
    std::vector<uint8_t> buf(static_cast<size_t>(K) * p.line_stride);
    MZ_HIP(eng, hipMemcpy2DAsync(buf.data(), p.line_stride, p.blocks + static_cast<size_t>(env) * p.line_stride,
                                 static_cast<size_t>(p.E) * p.line_stride, p.line_stride, K, hipMemcpyDeviceToHost,
                                 stream));
    MZ_HIP(eng, hipStreamSynchronize(stream));
    const int n_root = eng->h_nlegal[env];
    std::vector<int32_t> loc_of(static_cast<size_t>(K), -1);
    loc_of[0] = 0;
    for (int k = 0; k < K; ++k) {
        const int loc = loc_of[k];                       // (parents have smaller expansion indices than their children)
        const bool placed = loc >= 0 && (loc >> 1) < K;
        const uint8_t* blk = buf.data() + static_cast<size_t>(placed ? (loc >> 1) : 0) * p.line_stride + (placed ? (loc & 1) * 64u : 0u);
        const mz::HbmChild* rec = reinterpret_cast<const mz::HbmChild*>(blk);
        if (placed && k <= eng->sim)
            for (int i = 0; i < (k == 0 ? n_root : A); ++i)
                if (rec[i].links.child_node > k && rec[i].links.child_node < K) loc_of[rec[i].links.child_node] = rec[i].links.block_loc;
        const bool written = placed && k <= eng->sim;
        for (int i = 0; i < A; ++i) {
            const size_t o = static_cast<size_t>(k) * A + i;
            const bool live = written && (k > 0 || i < n_root);
            if (visits) visits[o] = live ? rec[i].links.visits : 0;
            if (value_sum) value_sum[o] = live ? rec[i].stats.value_sum : 0.0;
            if (prior) prior[o] = live ? rec[i].stats.prior : 0.0;
            if (reward) reward[o] = live ? static_cast<double>(rec[i].links.reward) : 0.0;
            if (child_node) child_node[o] = live ? rec[i].links.child_node : -1;
        }
    }
    return MZMCTS_OK;
}

// ---- fully-connected network in-kernel ----------------------------------------------------------------
static int bind_mlp(mz::FcMlp* m, int in, const int32_t* hidden, int n_hidden, int out, int* cursor, int* lds_cursor,
                    int* max_hidden) {
    if (n_hidden < 0 || n_hidden > mz::kFcMaxLayers - 1) return -1;
    int widths[mz::kFcMaxLayers + 1];
    widths[0] = in;
    for (int i = 0; i < n_hidden; ++i) widths[i + 1] = hidden[i];
    widths[n_hidden + 1] = out;
    m->n_layers = n_hidden + 1;
    for (int l = 0; l < m->n_layers; ++l) {
        if (widths[l] <= 0 || widths[l + 1] <= 0 || widths[l] > mz::kFcMaxWidth || widths[l + 1] > mz::kFcMaxWidth) return -1;
        m->layer[l].in = widths[l];
        m->layer[l].out = widths[l + 1];
        m->layer[l].w_off = *cursor;
        *cursor += widths[l] * widths[l + 1];
        m->layer[l].b_off = *cursor;
        *cursor += widths[l + 1];
        m->layer[l].in_pad = (widths[l] + 3) / 4 * 4;
        m->layer[l].w_lds = *lds_cursor;
        *lds_cursor += m->layer[l].in_pad * widths[l + 1];
        m->layer[l].b_lds = *lds_cursor;
        *lds_cursor += (widths[l + 1] + 3) / 4 * 4;
        if (l < m->n_layers - 1 && widths[l + 1] > *max_hidden) *max_hidden = widths[l + 1];
    }
    return 0;
}

int mzmcts_fc_configure(mzmcts_engine* eng, const mzmcts_fc_desc* d, const float* weights, int64_t n_weights) {
    if (!eng || !d || !weights) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_fc_configure: null argument");
    if (d->encoding_size != eng->p.H)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_fc_configure: encoding_size must equal the engine's hidden_floats");
    mz::FcNet net{};
    net.obs = d->observation_floats;
    net.enc = d->encoding_size;
    net.A = eng->p.A;
    net.F = eng->p.F;
    net.support = eng->p.support;
    int cursor = 0, lds_cursor = 0, max_hidden = 1;
    int rc = 0;
    rc |= bind_mlp(&net.repr, net.obs, d->hidden[0], d->n_hidden[0], net.enc, &cursor, &lds_cursor, &max_hidden);
    rc |= bind_mlp(&net.dyn, net.enc + net.A, d->hidden[1], d->n_hidden[1], net.enc, &cursor, &lds_cursor, &max_hidden);
    rc |= bind_mlp(&net.reward, net.enc, d->hidden[2], d->n_hidden[2], net.F, &cursor, &lds_cursor, &max_hidden);
    rc |= bind_mlp(&net.policy, net.enc, d->hidden[3], d->n_hidden[3], net.A, &cursor, &lds_cursor, &max_hidden);
    rc |= bind_mlp(&net.value, net.enc, d->hidden[4], d->n_hidden[4], net.F, &cursor, &lds_cursor, &max_hidden);
    if (rc != 0 || net.obs <= 0 || net.obs > mz::kFcMaxWidth)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_fc_configure: layer sizes outside the supported range "
                                             "(<= 3 hidden layers per MLP, widths <= 256)");
    if (cursor != n_weights)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_fc_configure: n_weights does not match the layer description");
    net.n_weights = cursor;
    net.n_weights_lds = lds_cursor;
    // scratch regions (floats, 16-byte aligned): x_in | raw | norm | reward | value | policy | 3 heads x 2 temps
    auto pad4 = [](int v) { return (v + 3) / 4 * 4; };
    int off = pad4(std::max(net.obs, net.enc + net.A));
    net.off_raw = off;
    off += pad4(net.enc);
    net.off_norm = off;
    off += pad4(net.enc);
    net.off_reward = off;
    off += pad4(net.F);
    net.off_value = off;
    off += pad4(net.F);
    net.off_policy = off;
    off += pad4(net.A);
    int temp[3][2];
    for (int h = 0; h < 3; ++h)
        for (int t = 0; t < 2; ++t) {
            temp[h][t] = off;
            off += pad4(max_hidden);
        }
    net.scratch_floats = std::max(off, 64);  // >= 256 B: the fused kernel's backup borrows it (tree_device.h)
    auto job_of = [&](const mz::FcMlp& m, int l, int x_first, int y_last, int head) {
        mz::FcJob jb{};
        const mz::FcLayer& L = m.layer[l];
        jb.in_pad = L.in_pad;
        jb.out = L.out;
        jb.w_lds = L.w_lds;
        jb.b_lds = L.b_lds;
        jb.x_off = (l == 0) ? x_first : temp[head][(l - 1) & 1];
        jb.y_off = (l == m.n_layers - 1) ? y_last : temp[head][l & 1];
        jb.elu = (l == m.n_layers - 1) ? 0 : 1;
        return jb;
    };
    auto chain = [&](const mz::FcMlp& m, int x_first, int y_last, mz::FcPhase* phases, int32_t* count) {
        *count = m.n_layers;
        for (int l = 0; l < m.n_layers; ++l) {
            phases[l].n_jobs = 1;
            phases[l].job[0] = job_of(m, l, x_first, y_last, 0);
            phases[l].total_out = phases[l].job[0].out;
        }
    };
    struct Head {
        const mz::FcMlp* mlp;
        int x_first, y_last;
    };
    auto heads = [&](std::initializer_list<Head> hs, mz::FcPhase* phases, int32_t* count) {
        int depth = 0;
        for (const Head& h : hs) depth = std::max(depth, static_cast<int>(h.mlp->n_layers));
        *count = depth;
        for (int l = 0; l < depth; ++l) {
            phases[l].n_jobs = 0;
            phases[l].total_out = 0;
            int head_index = 0;
            for (const Head& h : hs) {
                if (l < h.mlp->n_layers) {
                    phases[l].job[phases[l].n_jobs] = job_of(*h.mlp, l, h.x_first, h.y_last, head_index);
                    phases[l].total_out += h.mlp->layer[l].out;
                    phases[l].n_jobs += 1;
                }
                ++head_index;
            }
        }
    };
    chain(net.repr, 0, net.off_raw, net.init_pre, &net.n_init_pre);
    heads({{&net.policy, net.off_norm, net.off_policy}, {&net.value, net.off_norm, net.off_value}}, net.init_post,
          &net.n_init_post);
    chain(net.dyn, 0, net.off_raw, net.rec_pre, &net.n_rec_pre);
    heads({{&net.reward, net.off_raw, net.off_reward},
           {&net.policy, net.off_norm, net.off_policy},
           {&net.value, net.off_norm, net.off_value}},
          net.rec_post, &net.n_rec_post);
    eng->fc = net;
    eng->fc_weights = weights;
    eng->fc_ready = true;
    return MZMCTS_OK;
}

// The narrow kernels (fused_narrow.hip) run when the network qualifies and the caller did not ask for the
// generic ones; asking for them on a network that does not qualify is reported by mzmcts_set_fused_options.
}  // extern "C"

bool mzhost_use_narrow(const mzmcts_engine* eng) {
    if (!eng->fc_ready || eng->fused_variant == MZMCTS_FUSED_GENERIC) return false;
    mz::NarrowLayout lay{};
    return mz::narrow_supported(eng->p, eng->fc) && mz::plan_narrow_layout(eng->p, eng->fc, kLdsPerWorkgroup, &lay);
}

extern "C" {

int mzmcts_set_fused_options(mzmcts_engine* eng, int32_t variant, int32_t publish_tree) {
    if (!eng) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_set_fused_options: null engine");
    if (variant != MZMCTS_FUSED_AUTO && variant != MZMCTS_FUSED_GENERIC && variant != MZMCTS_FUSED_NARROW)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_set_fused_options: unknown variant");
    if (variant == MZMCTS_FUSED_NARROW) {
        if (!eng->fc_ready) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_set_fused_options: call mzmcts_fc_configure first");
        mz::NarrowLayout lay{};
        if (!mz::narrow_supported(eng->p, eng->fc) || !mz::plan_narrow_layout(eng->p, eng->fc, kLdsPerWorkgroup, &lay))
            return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_set_fused_options: the narrow kernel needs group_width 16, one hidden "
                                                 "layer of <= 16 units per MLP, encoding + actions <= 16, support <= 32 logits");
    }
    eng->fused_variant = variant;
    eng->publish_tree = publish_tree != 0;
    return MZMCTS_OK;
}

int32_t mzmcts_fused_variant(mzmcts_engine* eng) {
    if (!eng || !eng->fc_ready) return 0;
    if (mzhost_use_narrow(eng)) return MZMCTS_FUSED_NARROW;
    mz::FusedLayout lay{};
    return mz::plan_fused_layout(eng->p, eng->fc, true, kLdsPerWorkgroup, &lay) ? MZMCTS_FUSED_GENERIC : 0;
}

int mzmcts_fc_initial_inference(mzmcts_engine* eng, const float* observations, float* value_logits, float* reward_logits,
                                float* policy_logits, float* hidden_out, void* stream) {
    if (!eng || !observations || !value_logits || !reward_logits || !policy_logits || !hidden_out)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_fc_initial_inference: null argument");
    if (!eng->fc_ready) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_fc_initial_inference: call mzmcts_fc_configure first");
    if (mzhost_use_narrow(eng))
        MZ_HIP(eng, mz::launch_fc_inference_narrow(eng->p, eng->fc, eng->fc_weights, true, observations, nullptr,
                                                   value_logits, reward_logits, policy_logits, hidden_out,
                                                   static_cast<hipStream_t>(stream)));
    else
        MZ_HIP(eng, mz::launch_fc_inference(eng->p, eng->fc, eng->fc_weights, true, observations, nullptr, value_logits,
                                            reward_logits, policy_logits, hidden_out, static_cast<hipStream_t>(stream)));
    return MZMCTS_OK;
}

int mzmcts_fc_recurrent_inference(mzmcts_engine* eng, const float* hidden, const int64_t* action, float* value_logits,
                                  float* reward_logits, float* policy_logits, float* hidden_out, void* stream) {
    if (!eng || !hidden || !action || !value_logits || !reward_logits || !policy_logits || !hidden_out)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_fc_recurrent_inference: null argument");
    if (!eng->fc_ready) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_fc_recurrent_inference: call mzmcts_fc_configure first");
    if (mzhost_use_narrow(eng))
        MZ_HIP(eng, mz::launch_fc_inference_narrow(eng->p, eng->fc, eng->fc_weights, false, hidden, action, value_logits,
                                                   reward_logits, policy_logits, hidden_out,
                                                   static_cast<hipStream_t>(stream)));
    else
        MZ_HIP(eng, mz::launch_fc_inference(eng->p, eng->fc, eng->fc_weights, false, hidden, action, value_logits,
                                            reward_logits, policy_logits, hidden_out, static_cast<hipStream_t>(stream)));
    return MZMCTS_OK;
}

int64_t mzmcts_fused_lds_bytes(mzmcts_engine* eng, int32_t hidden_in_lds) {
    if (!eng || !eng->fc_ready) return 0;
    if (mzhost_use_narrow(eng)) {
        mz::NarrowLayout nl{};
        mz::plan_narrow_layout(eng->p, eng->fc, kLdsPerWorkgroup, &nl);
        return nl.total_bytes;
    }
    mz::FusedLayout lay{};
    if (!mz::plan_fused_layout(eng->p, eng->fc, hidden_in_lds != 0, kLdsPerWorkgroup, &lay)) return 0;
    return lay.total_bytes;
}

}  // extern "C"

// One whole-move kernel (narrow or generic, see mzhost_use_narrow) with the given per-move control block.
int mzhost_launch_fused_move(mzmcts_engine* eng, const float* observations, const mz::MoveCtl& ctl, bool hidden_in_lds,
                             hipStream_t stream) {
    if (mzhost_use_narrow(eng)) {
        mz::NarrowLayout nl{};
        mz::plan_narrow_layout(eng->p, eng->fc, kLdsPerWorkgroup, &nl);
        ProfScope scope(eng, stream, kProfFused);
        MZ_HIP(eng, mz::launch_search_fused_narrow(eng->p, eng->fc, nl, eng->fc_weights, observations, ctl, eng->p.S,
                                                   eng->publish_tree ? 1 : 0, stream, scope.get()));
        eng->tree_published = eng->publish_tree;
    } else {
        mz::FusedLayout lay{};
        if (!mz::plan_fused_layout(eng->p, eng->fc, hidden_in_lds, kLdsPerWorkgroup, &lay))
            return fail(eng, MZMCTS_ERR_INVALID, "fused search: the trees of one workgroup do not fit in 160 KB of LDS (use a "
                                                 "wider group_width or the lock-step path)");
        ProfScope scope(eng, stream, kProfFused);
        MZ_HIP(eng, mz::launch_search_fused_fc(eng->p, eng->fc, lay, eng->fc_weights, observations, ctl, eng->p.S, stream,
                                               scope.get()));
        eng->tree_published = true;
    }
    return MZMCTS_OK;
}

extern "C" {

int mzmcts_search_fused_fc(mzmcts_engine* eng, const float* observations, int32_t hidden_in_lds, void* stream_) {
    if (!eng || !observations) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_search_fused_fc: null argument");
    if (!eng->fc_ready) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_search_fused_fc: call mzmcts_fc_configure first");
    if (!eng->search_begun) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_search_fused_fc called before begin_search");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    mz::MoveCtl ctl{};
    ctl.noise = eng->noise_this_search ? eng->d_noise : nullptr;
    ctl.rng_skip = eng->d_skip;
    int rc = mzhost_launch_fused_move(eng, observations, ctl, hidden_in_lds != 0, stream);
    if (rc) return rc;
    eng->roots_ready = true;
    eng->sim = eng->p.S;
    return MZMCTS_OK;
}



#ifdef MZ_STAMPS
// diagnostic build only: per-phase cycle sums of the fused kernel (see tools/stamp_fused.py)
int mzmcts_debug_read_stamps(unsigned long long* out, int32_t reset) {
    unsigned long long narrow[16];
    if (mz::read_stamp_sums(out, reset != 0) != hipSuccess || mz::read_stamp_sums_narrow(narrow, reset != 0) != hipSuccess)
        return -2;
    for (int i = 0; i < 16; ++i) out[i] += narrow[i];
    return 0;
}
#endif

int mzmcts_set_profiling(mzmcts_engine* eng, int32_t enabled) {
    if (!eng) return MZMCTS_ERR_INVALID;
    eng->profiling = enabled != 0;
    return MZMCTS_OK;
}

int mzmcts_set_device_noise(mzmcts_engine* eng, int32_t enabled) {
    if (!eng) return MZMCTS_ERR_INVALID;
    if (enabled && !(eng->cfg.root_dirichlet_alpha > 0.0 && eng->cfg.root_dirichlet_alpha <= 1.0))
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_set_device_noise: the device draws Dirichlet noise for 0 < "
                                             "root_dirichlet_alpha <= 1 only (larger shapes go through legacy_gauss "
                                             "and its cached value: host)");
    eng->device_noise = enabled != 0;
    return MZMCTS_OK;
}

int mzmcts_get_noise(mzmcts_engine* eng, double* noise_out) {
    if (!eng || !noise_out) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_get_noise: null argument");
    if (eng->noise_on_device && !eng->have_readout)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_get_noise: device-drawn rows reach the host with mzmcts_readout");
    std::memcpy(noise_out, eng->h_noise, sizeof(double) * static_cast<size_t>(eng->p.E) * eng->p.A);
    return MZMCTS_OK;
}

int mzmcts_set_select_queue(mzmcts_engine* eng, int32_t trees_per_wavefront) {
    if (!eng) return MZMCTS_ERR_INVALID;
    if (trees_per_wavefront < 0 || trees_per_wavefront > 4096)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_set_select_queue: trees_per_wavefront must be in [0, 4096]");
    eng->select_queue_trees = trees_per_wavefront;
    return MZMCTS_OK;
}

int mzmcts_get_profile(mzmcts_engine* eng, mzmcts_profile* out, int32_t reset) {
    if (!eng || !out) return MZMCTS_ERR_INVALID;
    for (size_t i = 0; i < eng->events_used; ++i) {
        EventPair& ev = eng->events[i];
        MZ_HIP(eng, hipEventSynchronize(ev.end));
        float ms = 0.f;
        MZ_HIP(eng, hipEventElapsedTime(&ms, ev.begin, ev.end));
        switch (ev.kind) {
            case kProfSelect:
                eng->prof.select_ms += ms;
                eng->prof.select_launches += 1;
                break;
            case kProfBackup:
                eng->prof.expand_backup_ms += ms;
                eng->prof.expand_backup_launches += 1;
                break;
            case kProfStep:
                eng->prof.step_ms += ms;
                eng->prof.step_launches += 1;
                break;
            case kProfFused:
                eng->prof.fused_ms += ms;
                eng->prof.fused_launches += 1;
                break;
            default:
                eng->prof.root_ms += ms;
                eng->prof.root_launches += 1;
                break;
        }
    }
    eng->events_used = 0;
    *out = eng->prof;
    if (reset) eng->prof = mzmcts_profile{};
    return MZMCTS_OK;
}

int64_t mzmcts_device_bytes(const mzmcts_engine* eng) { return eng ? eng->device_bytes : 0; }
}  // extern "C"
