// mzmcts_moves.hip -- batches of moves queued back to back without host round trips (include/mzmcts.h
// mzmcts_moves_*): the host mirror of every env's numpy stream, the speculative draw of a batch's exploration noise,
// the rewind of a mirror whose env left the pre-drawn path, and the collection of a batch's results.
#include "engine_host.h"

extern "C" {

// ---- batches of moves without host round trips ---------------------------------------------------------------
// Stream bookkeeping.  Per env the numpy stream is consumed, move after move, as
//     [Dirichlet(m)] [tie-breaks of search m] [select_action(m)]      (self_play.py:303-315, 372-378, 223-246)
// The host draws every Dirichlet row of a batch up front on its mirror, assuming search m spends one word on
// the unavoidable first-simulation tie (none with a single legal action) and select_action consumes what the
// temperature implies (0 words at T = 0, 2 at T = 1); the kernels consume the tie-break and sampling words on
// the device copy and skip the Dirichlet words (rng_skip ring).  An env whose search spent a different number
// of tie-break words stalls from the next move on (kernel_common.h); collect() puts the mirror back to the
// state recorded after the last noise row that was really used and replays what the device consumed.
// The NEXT batch may be drawn the same way while the current one is still running (predraw_next): collect()
// then also redraws, from the true stream position, the rows of every env whose current batch did not end as
// assumed, before submit_next() uploads them.

// Take env e's mirror back (or forward) to `target`, a state recorded while drawing sets[0..n_sets) (oldest
// first).  Only the first regeneration of a set is backed up, so: if a backed-up block covers the target, use
// it directly; otherwise start from the latest backup before the target and walk forward.
static void restore_stream(mzmcts_engine* eng, int e, const MoveRecord& target, ChainSet* const* sets, int n_sets) {
    mz::HostStream& s = eng->mirror(e);
    const ChainSet* before = nullptr;
    const ChainSet* covering = nullptr;
    for (int i = 0; i < n_sets; ++i) {
        const ChainSet* c = sets[i];
        if (!c || !c->drawn || !c->env_twisted[e]) continue;
        if (c->twist_words[e] < target.words)
            before = c;
        else if (!covering)
            covering = c;
    }
    if (before) {
        std::memcpy(s.key, before->twist_keys.data() + static_cast<size_t>(e) * mz::kMtN, sizeof(s.key));
        s.pos = mz::kMtN;
        s.words = before->twist_words[e];
        s.skip(target.words - s.words);
    } else {
        if (covering) std::memcpy(s.key, covering->twist_keys.data() + static_cast<size_t>(e) * mz::kMtN, sizeof(s.key));
        s.pos = target.pos;
        s.words = target.words;
    }
    s.has_gauss = target.has_gauss;
    s.gauss = target.gauss;
}

static mz::MoveInputsExtra move_extras(mzmcts_engine* eng, bool own_inputs);

static int ensure_batch_capacity(mzmcts_engine* eng, int n_moves) {
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (n_moves <= b.capacity) return 0;
    if (b.in_flight || b.set[0].drawn || b.set[1].drawn)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves: a larger batch needs new buffers; collect the batches in flight first");
    const size_t E = static_cast<size_t>(eng->p.E), A = static_cast<size_t>(eng->p.A), M = static_cast<size_t>(n_moves);
    auto align = [](size_t v) { return (v + 255) / 256 * 256; };
    b.o_skip = align(sizeof(double) * M * E * A);
    b.o_temp = align(b.o_skip + sizeof(uint32_t) * M * E);
    b.o_limit = align(b.o_temp + sizeof(double) * E);
    b.o_expect = align(b.o_limit + sizeof(int32_t) * E);
    b.in_bytes = align(b.o_expect + sizeof(uint32_t) * E);
    b.o_actions = 0;
    b.o_visits = align(sizeof(int32_t) * E);
    b.o_rvs = align(b.o_visits + sizeof(int32_t) * E * A);
    b.o_pred = align(b.o_rvs + sizeof(double) * E);
    b.o_depth = align(b.o_pred + sizeof(float) * E);
    b.o_ties = align(b.o_depth + sizeof(int32_t) * E);
    b.o_sample = align(b.o_ties + sizeof(uint32_t) * E);
    b.o_dsum = align(b.o_sample + sizeof(uint32_t) * E);
    b.out_stride = align(b.o_dsum + sizeof(int32_t) * E);
    int rc;  // (earlier, smaller buffers stay registered with the engine and are freed with it)
    if ((rc = dev_alloc(eng, &b.d_in, b.in_bytes))) return rc;
    if ((rc = dev_alloc(eng, &b.d_out, b.out_stride * M))) return rc;
    for (int q = 0; q < 2; ++q)
        if ((rc = pinned_alloc(eng, &b.h_out_set[q], b.out_stride * M))) return rc;
    b.h_out = b.h_out_set[b.host_set];
    if (!b.d_stall && (rc = dev_alloc(eng, &b.d_stall, E))) return rc;
    if (!b.done) MZ_HIP(eng, hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
    if (!b.move_done) MZ_HIP(eng, hipEventCreateWithFlags(&b.move_done, hipEventDisableTiming));
    if (!b.copy_stream) MZ_HIP(eng, hipStreamCreateWithFlags(&b.copy_stream, hipStreamNonBlocking));
    for (ChainSet& c : b.set) {
        if ((rc = pinned_alloc(eng, &c.h_in, b.in_bytes))) return rc;
        c.legal.assign(E * A, 0);
        c.nlegal.assign(E, 0);
        c.to_play.assign(E, 0);
        c.start.resize(E);
        c.start_lag.assign(E, 0);
        c.rec.resize(M * E);
        c.env_twisted.assign(E, 0);
        c.twist_words.assign(E, 0);
        c.twist_keys.resize(E * mz::kMtN);
        c.temperature.assign(E, 0.0);
        c.tail_ties.assign(E, 0);
        c.tail_sample.assign(E, 0);
        c.deferred.assign(E, 0);
    }
    b.capacity = n_moves;
    // Run both transfers once at full size: the runtime sets up its large-copy path on first use (tens of
    // milliseconds), which would otherwise land in the first full-size batch.
    MZ_HIP(eng, hipMemcpy(b.d_in, b.set[0].h_in, b.in_bytes, hipMemcpyHostToDevice));
    for (int q = 0; q < 2; ++q) MZ_HIP(eng, hipMemcpy(b.h_out_set[q], b.d_out, b.out_stride * M, hipMemcpyDeviceToHost));
    return 0;
}

static int check_move_inputs(mzmcts_engine* eng, int32_t n_moves, const int32_t* legal, const int32_t* num_legal,
                             const int32_t* to_play, const double* temperature, const char* who) {
    if (!eng || !legal || !num_legal || !to_play || !temperature) return fail(eng, MZMCTS_ERR_INVALID, std::string(who) + ": null argument");
    if (!eng->fc_ready) return fail(eng, MZMCTS_ERR_INVALID, std::string(who) + ": call mzmcts_fc_configure first");
    if (n_moves < 1 || n_moves > 4096) return fail(eng, MZMCTS_ERR_INVALID, std::string(who) + ": n_moves out of range");
    const int E = eng->p.E, A = eng->p.A;
    for (int e = 0; e < E; ++e) {
        const int n = num_legal[e];
        if (n < 0 || n > A)
            return fail(eng, MZMCTS_ERR_LEGAL_RANGE, "Legal actions should be a subset of the action space.");
        for (int i = 0; i < n; ++i) {
            const int a = legal[static_cast<size_t>(e) * A + i];
            if (a < 0 || a >= A)
                return fail(eng, MZMCTS_ERR_LEGAL_RANGE, "Legal actions should be a subset of the action space.");
        }
        const double t = temperature[e];
        if (!(t == 0.0 || std::isinf(t) || (mz::exact_inverse_temperature(t) && std::pow(eng->p.S, 1.0 / t) < 9.0e15)))
            return fail(eng, MZMCTS_ERR_INVALID, std::string(who) + ": the device samples actions at temperature 0, inf or 1/k, "
                                                                    "k = 1..4, only (visit_count ** (1 / T) needs the host's pow)");
    }
    return 0;
}

// words select_action consumes on the device; +inf draws a bounded integer by rejection: unknown in advance
static int assumed_sample_words(double t) { return (t == 0.0) ? 0 : (std::isinf(t) ? -1 : 2); }
// The first simulation always ties: the root has no visits yet, so every child scores 0 (sqrt(0) in ucb_score,
// self_play.py:385-390) and select_child draws numpy.random.choice over all n of them -- one masked 32-bit word
// when n is a power of two, a rejection loop otherwise (one word is the likeliest outcome and the one assumed).
// Later ties need exactly equal fp64 scores.
static uint32_t assumed_tie_words(int n) { return n > 1 ? 1u : 0u; }

// Draw env e's rows of set c from the mirror's current state.  `tail`: the mirror stands right after the
// previous batch's last noise row and that batch has not finished -- first step over what its last move is
// assumed to consume.  Returns false (nothing drawn) when that cannot be known in advance.
static bool draw_env_rows(mzmcts_engine* eng, ChainSet& c, int e, bool tail, const ChainSet* under) {
    mzmcts_engine::MoveBatch& b = eng->batch;
    const int E = eng->p.E, A = eng->p.A, n_moves = c.n_moves;
    const size_t EA = static_cast<size_t>(E) * A;
    double* h_noise = reinterpret_cast<double*>(c.h_in);
    uint32_t* h_skip = reinterpret_cast<uint32_t*>(c.h_in + b.o_skip);
    double* h_temp = reinterpret_cast<double*>(c.h_in + b.o_temp);
    int32_t* h_limit = reinterpret_cast<int32_t*>(c.h_in + b.o_limit);
    uint32_t* h_expect = reinterpret_cast<uint32_t*>(c.h_in + b.o_expect);
    const int n = c.nlegal[e];
    const double t = c.temperature[e];
    const int assumed = assumed_sample_words(t);
    const uint32_t tie_words = assumed_tie_words(n);
    h_temp[e] = t;
    h_limit[e] = (n == 0) ? 0 : (assumed < 0 ? 1 : n_moves);
    h_expect[e] = tie_words;
    c.env_twisted[e] = 0;
    c.deferred[e] = 0;
    mz::HostStream& s = eng->mirror(e);
    uint32_t lag0 = eng->lag[e];
    if (tail && n > 0) {
        const int under_n = under->nlegal[e];
        const int under_sample = assumed_sample_words(under->temperature[e]);
        if (under_n > 0 && under_sample < 0) {  // the batch underneath samples at T = inf: draw these rows at collect()
            c.deferred[e] = 1;
            h_limit[e] = 0;
            for (int m = 0; m < n_moves; ++m) {
                h_skip[static_cast<size_t>(m) * E + e] = 0;
                for (int i = 0; i < A; ++i) h_noise[static_cast<size_t>(m) * EA + static_cast<size_t>(e) * A + i] = 0.0;
            }
            return false;
        }
    }
    c.start[e] = MoveRecord{s.pos, s.has_gauss, s.gauss, s.words};
    c.start_lag[e] = lag0;
    s.twist_backup = c.twist_keys.data() + static_cast<size_t>(e) * mz::kMtN;
    s.twisted = false;
    c.tail_ties[e] = 0;
    c.tail_sample[e] = 0;
    if (tail && n > 0 && under->nlegal[e] > 0) {
        c.tail_ties[e] = assumed_tie_words(under->nlegal[e]);
        c.tail_sample[e] = static_cast<uint32_t>(assumed_sample_words(under->temperature[e]));
        s.skip(static_cast<uint64_t>(c.tail_ties[e]) + c.tail_sample[e]);
        lag0 = 0;  // the batch underneath hands the device copy over in step with the mirror
    }
    const double alpha = eng->cfg.root_dirichlet_alpha;
    for (int m = 0; m < n_moves; ++m) {
        double* row = h_noise + static_cast<size_t>(m) * EA + static_cast<size_t>(e) * A;
        for (int i = 0; i < A; ++i) row[i] = 0.0;
        uint32_t skip = 0;
        if (n > 0 && m < h_limit[e]) {
            if (m > 0) s.skip(static_cast<uint64_t>(tie_words) + static_cast<uint64_t>(assumed));
            const uint64_t before = s.words;
            if (c.add_noise) s.dirichlet(alpha, n, row);
            skip = static_cast<uint32_t>(s.words - before) + (m == 0 ? lag0 : 0u);
        }
        h_skip[static_cast<size_t>(m) * E + e] = skip;
        c.rec[static_cast<size_t>(m) * E + e] = MoveRecord{s.pos, s.has_gauss, s.gauss, s.words};
    }
    if (n > 0) eng->lag[e] = 0;
    c.env_twisted[e] = s.twisted ? 1 : 0;
    c.twist_words[e] = s.twist_words;
    s.twist_backup = nullptr;
    s.twisted = false;
    return true;
}

static void fill_set(mzmcts_engine* eng, ChainSet& c, int32_t n_moves, const int32_t* legal, const int32_t* num_legal,
                     const int32_t* to_play, int32_t add_noise, const double* temperature) {
    const int E = eng->p.E, A = eng->p.A;
    std::memcpy(c.legal.data(), legal, sizeof(int32_t) * static_cast<size_t>(E) * A);
    std::memcpy(c.nlegal.data(), num_legal, sizeof(int32_t) * E);
    std::memcpy(c.to_play.data(), to_play, sizeof(int32_t) * E);
    std::memcpy(c.temperature.data(), temperature, sizeof(double) * E);
    c.n_moves = n_moves;
    c.add_noise = add_noise != 0;
}

static int upload_set(mzmcts_engine* eng, ChainSet& c, hipStream_t stream) {
    mzmcts_engine::MoveBatch& b = eng->batch;
    const int E = eng->p.E, A = eng->p.A;
    std::memcpy(eng->h_legal, c.legal.data(), sizeof(int32_t) * static_cast<size_t>(E) * A);
    std::memcpy(eng->h_nlegal, c.nlegal.data(), sizeof(int32_t) * E);
    std::memcpy(eng->h_to_play, c.to_play.data(), sizeof(int32_t) * E);
    MZ_HIP(eng, hipMemcpyAsync(eng->d_upload, eng->h_upload, eng->upload_bytes_no_noise, hipMemcpyHostToDevice, stream));
    MZ_HIP(eng, hipMemcpyAsync(b.d_in, c.h_in, b.in_bytes, hipMemcpyHostToDevice, stream));
    MZ_HIP(eng, hipMemsetAsync(b.d_stall, 0, static_cast<size_t>(E), stream));
    b.enqueued = 0;
    b.in_flight = true;
    eng->search_begun = false;
    eng->roots_ready = false;
    eng->have_readout = false;
    return MZMCTS_OK;
}

int mzmcts_moves_prepare(mzmcts_engine* eng, int32_t n_moves, const int32_t* legal, const int32_t* num_legal,
                         const int32_t* to_play, int32_t add_noise, const double* temperature, void* stream_) {
    int rc = check_move_inputs(eng, n_moves, legal, num_legal, to_play, temperature, "mzmcts_moves_prepare");
    if (rc) return rc;
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (b.in_flight || b.set[b.cur ^ 1].drawn)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_prepare: collect the previous batch first");
    if ((rc = ensure_batch_capacity(eng, n_moves))) return rc;
    ChainSet& c = b.set[b.cur];
    b.device_inputs = false;
    fill_set(eng, c, n_moves, legal, num_legal, to_play, add_noise, temperature);
    eng->for_each_env([&](int lo, int hi) {
        for (int e = lo; e < hi; ++e) draw_env_rows(eng, c, e, false, nullptr);
    });
    c.drawn = true;
    c.speculative = false;
    return upload_set(eng, c, static_cast<hipStream_t>(stream_));
}

// ---- batches whose inputs live on the device -------------------------------------------------------------------------
// (games whose legal action sets change from move to move: the host cannot know a move's legal set -- nor, therefore,
// the length of its Dirichlet row -- before the moves before it have been played on the device)
static int ensure_inputs_capacity(mzmcts_engine* eng, int n_moves) {
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (n_moves <= b.inputs_capacity) return 0;
    const size_t E = static_cast<size_t>(eng->p.E), A = static_cast<size_t>(eng->p.A);
    auto align = [](size_t v) { return (v + 255) / 256 * 256; };
    b.o2_nlegal = 0;
    b.o2_to_play = align(sizeof(int32_t) * E);
    b.o2_words = align(b.o2_to_play + sizeof(int32_t) * E);
    b.o2_legal = align(b.o2_words + sizeof(uint32_t) * E);
    b.in2_stride = align(b.o2_legal + sizeof(int32_t) * E * A);
    int rc;
    if ((rc = dev_alloc(eng, &b.d_inputs, b.in2_stride * static_cast<size_t>(n_moves)))) return rc;
    for (int q = 0; q < 2; ++q)
        if ((rc = pinned_alloc(eng, &b.h_inputs_set[q], b.in2_stride * static_cast<size_t>(n_moves)))) return rc;
    b.h_inputs = b.h_inputs_set[b.host_set];
    b.inputs_capacity = n_moves;
    return 0;
}

int mzmcts_moves_prepare_device(mzmcts_engine* eng, int32_t n_moves, const int32_t* legal_dev, const int32_t* num_legal_dev,
                                const int32_t* to_play_dev, int32_t add_noise, const double* temperature, void* stream_) {
    if (!eng || !legal_dev || !num_legal_dev || !to_play_dev || !temperature)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_prepare_device: null argument");
    // (a fully-connected network is needed by mzmcts_moves_enqueue only: lock-step moves bring their own network)
    if (n_moves < 1 || n_moves > 4096) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_prepare_device: n_moves out of range");
    if (add_noise && !(eng->cfg.root_dirichlet_alpha > 0.0 && eng->cfg.root_dirichlet_alpha <= 1.0))
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_prepare_device: the device draws Dirichlet noise for 0 < "
                                             "root_dirichlet_alpha <= 1 only");
    const int E = eng->p.E;
    for (int e = 0; e < E; ++e) {
        const double t = temperature[e];
        if (!(t == 0.0 || std::isinf(t) || (mz::exact_inverse_temperature(t) && std::pow(eng->p.S, 1.0 / t) < 9.0e15)))
            return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_prepare_device: the device samples actions at temperature 0, "
                                                 "inf or 1/k, k = 1..4, only");
    }
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (b.in_flight || b.set[b.cur ^ 1].drawn)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_prepare_device: collect the previous batch first");
    int rc;
    if ((rc = ensure_batch_capacity(eng, n_moves))) return rc;
    if ((rc = ensure_inputs_capacity(eng, n_moves))) return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    b.host_set ^= 1;                                 // (the batch before keeps its downloads: the host may still be filing them)
    b.h_out = b.h_out_set[b.host_set];
    b.h_inputs = b.h_inputs_set[b.host_set];
    b.downloaded = 0;
    ChainSet& c = b.set[b.cur];
    c.n_moves = n_moves;
    c.add_noise = add_noise != 0;
    std::memcpy(c.temperature.data(), temperature, sizeof(double) * E);
    // control block: words the mirrors consumed since the device copies last moved (move 0 steps over them), the
    // temperatures, a move limit of the whole batch; no noise rows, no assumed tie-break counts
    uint32_t* skip = reinterpret_cast<uint32_t*>(c.h_in + b.o_skip);
    std::memset(skip, 0, sizeof(uint32_t) * static_cast<size_t>(n_moves) * E);
    for (int e = 0; e < E; ++e) {
        skip[e] = eng->lag[e];
        eng->lag[e] = 0;
        reinterpret_cast<int32_t*>(c.h_in + b.o_limit)[e] = n_moves;
        reinterpret_cast<uint32_t*>(c.h_in + b.o_expect)[e] = 0u;
    }
    std::memcpy(c.h_in + b.o_temp, temperature, sizeof(double) * E);
    MZ_HIP(eng, hipMemcpyAsync(b.d_in + b.o_skip, c.h_in + b.o_skip, b.in_bytes - b.o_skip, hipMemcpyHostToDevice, stream));
    MZ_HIP(eng, hipMemsetAsync(b.d_stall, 0, static_cast<size_t>(E), stream));
    b.enqueued = 0;
    b.in_flight = true;
    b.device_inputs = true;
    b.temperature_threshold = 0;                     // (mzmcts_moves_temperature_threshold sets it per batch)
    b.finished = nullptr;
    b.lockstep_open = false;
    b.dev_legal = legal_dev;
    b.dev_nlegal = num_legal_dev;
    b.dev_to_play = to_play_dev;
    c.drawn = true;
    c.speculative = false;
    eng->search_begun = false;
    eng->roots_ready = false;
    eng->have_readout = false;
    return MZMCTS_OK;
}

int mzmcts_moves_inputs(mzmcts_engine* eng, int32_t* num_legal, int32_t* legal, int32_t* to_play) {
    if (!eng) return MZMCTS_ERR_INVALID;
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (!b.device_inputs || b.in_flight)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_inputs: collect a mzmcts_moves_prepare_device batch first");
    const size_t E = static_cast<size_t>(eng->p.E), A = static_cast<size_t>(eng->p.A);
    for (int m = 0; m < b.enqueued; ++m) {
        const uint8_t* blk = b.h_inputs + b.in2_stride * static_cast<size_t>(m);
        if (num_legal) std::memcpy(num_legal + m * E, blk + b.o2_nlegal, sizeof(int32_t) * E);
        if (to_play) std::memcpy(to_play + m * E, blk + b.o2_to_play, sizeof(int32_t) * E);
        if (legal) std::memcpy(legal + m * E * A, blk + b.o2_legal, sizeof(int32_t) * E * A);
    }
    return MZMCTS_OK;
}

int mzmcts_moves_predraw_next(mzmcts_engine* eng, int32_t n_moves, const int32_t* legal, const int32_t* num_legal,
                              const int32_t* to_play, int32_t add_noise, const double* temperature) {
    int rc = check_move_inputs(eng, n_moves, legal, num_legal, to_play, temperature, "mzmcts_moves_predraw_next");
    if (rc) return rc;
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (!b.in_flight) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_predraw_next: no batch in flight (use mzmcts_moves_prepare)");
    if (b.device_inputs) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_predraw_next: a device-input batch draws its noise on the device");
    if (b.set[b.cur ^ 1].drawn) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_predraw_next: the next batch is already drawn");
    if (n_moves > b.capacity)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_predraw_next: larger than the batch in flight (its buffers are in use)");
    ChainSet& under = b.set[b.cur];
    ChainSet& c = b.set[b.cur ^ 1];
    fill_set(eng, c, n_moves, legal, num_legal, to_play, add_noise, temperature);
    eng->for_each_env([&](int lo, int hi) {
        for (int e = lo; e < hi; ++e) draw_env_rows(eng, c, e, true, &under);
    });
    c.drawn = true;
    c.speculative = true;
    return MZMCTS_OK;
}

int mzmcts_moves_submit_next(mzmcts_engine* eng, void* stream_) {
    if (!eng) return MZMCTS_ERR_INVALID;
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (b.in_flight) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_submit_next: collect the batch in flight first");
    ChainSet& c = b.set[b.cur ^ 1];
    if (!c.drawn) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_submit_next: no pre-drawn batch (mzmcts_moves_predraw_next)");
    b.cur ^= 1;
    c.speculative = false;
    return upload_set(eng, c, static_cast<hipStream_t>(stream_));
}

int mzmcts_moves_discard_next(mzmcts_engine* eng) {
    if (!eng) return MZMCTS_ERR_INVALID;
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (b.in_flight) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_discard_next: collect the batch in flight first");
    ChainSet& c = b.set[b.cur ^ 1];
    if (!c.drawn) return MZMCTS_OK;
    ChainSet* sets[1] = {&c};
    eng->for_each_env([&](int lo, int hi) {
        for (int e = lo; e < hi; ++e) {
            if (c.deferred[e]) continue;  // nothing was drawn for this env
            restore_stream(eng, e, c.start[e], sets, 1);
            // what the finished batch's last move consumed was confirmed by its collect(): step over it again
            eng->mirror(e).skip(static_cast<uint64_t>(c.tail_ties[e]) + c.tail_sample[e]);
            eng->lag[e] = (c.tail_ties[e] | c.tail_sample[e]) ? 0u : c.start_lag[e];
        }
    });
    c.drawn = false;
    return MZMCTS_OK;
}

int mzmcts_moves_enqueue(mzmcts_engine* eng, const float* observations, void* stream_) {
    if (!eng || !observations) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_enqueue: null argument");
    mzmcts_engine::MoveBatch& b = eng->batch;
    const ChainSet& c = b.set[b.cur];
    if (!b.in_flight || b.enqueued >= c.n_moves)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_enqueue: no prepared move left in the batch");
    if (!eng->fc_ready) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_enqueue: call mzmcts_fc_configure first");
    if (b.lockstep_open) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_enqueue: a lock-step move is open (mzmcts_moves_end_lockstep)");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const size_t E = static_cast<size_t>(eng->p.E), A = static_cast<size_t>(eng->p.A);
    const int m = b.enqueued;
    uint8_t* out = b.d_out + b.out_stride * static_cast<size_t>(m);
    mz::MoveCtl ctl{};
    ctl.noise = c.add_noise ? reinterpret_cast<const double*>(b.d_in) + static_cast<size_t>(m) * E * A : nullptr;
    ctl.rng_skip = reinterpret_cast<const uint32_t*>(b.d_in + b.o_skip) + static_cast<size_t>(m) * E;
    const int32_t *host_legal = eng->p.root_action, *host_nlegal = eng->p.root_children, *host_to_play = eng->p.root_to_play;
    if (b.device_inputs) {
        // the search reads the caller's device arrays; first a small kernel records them for the host, steps over the
        // mirrors' pending words and draws the noise row (of this move's legal count) on the device
        eng->p.root_action = const_cast<int32_t*>(b.dev_legal);
        eng->p.root_children = const_cast<int32_t*>(b.dev_nlegal);
        eng->p.root_to_play = const_cast<int32_t*>(b.dev_to_play);
        uint8_t* blk = b.d_inputs + b.in2_stride * static_cast<size_t>(m);
        hipError_t err = mz::launch_move_inputs(eng->p, ctl.rng_skip, b.d_stall, reinterpret_cast<const int32_t*>(b.d_in + b.o_limit), m,
                                                c.add_noise, reinterpret_cast<int32_t*>(blk + b.o2_nlegal),
                                                reinterpret_cast<int32_t*>(blk + b.o2_to_play),
                                                reinterpret_cast<uint32_t*>(blk + b.o2_words),
                                                reinterpret_cast<int32_t*>(blk + b.o2_legal), move_extras(eng, false), stream);
        b.finished = nullptr;
        if (err != hipSuccess) {
            eng->p.root_action = const_cast<int32_t*>(host_legal), eng->p.root_children = const_cast<int32_t*>(host_nlegal);
            eng->p.root_to_play = const_cast<int32_t*>(host_to_play);
            return hip_fail(eng, err, "move_inputs_kernel");
        }
        ctl.noise = c.add_noise ? eng->p.noise_rows : nullptr;
        ctl.rng_skip = nullptr;
    }
    ctl.temperature = reinterpret_cast<const double*>(b.d_in + b.o_temp);
    ctl.move_limit = reinterpret_cast<const int32_t*>(b.d_in + b.o_limit);
    ctl.stall = b.d_stall;
    ctl.move_index = m;
    ctl.expected_ties = (m > 0 && !b.device_inputs) ? reinterpret_cast<const uint32_t*>(b.d_in + b.o_expect) : nullptr;
    ctl.actions = reinterpret_cast<int32_t*>(out + b.o_actions);
    ctl.visits = reinterpret_cast<int32_t*>(out + b.o_visits);
    ctl.root_value_sum = reinterpret_cast<double*>(out + b.o_rvs);
    ctl.root_predicted = reinterpret_cast<float*>(out + b.o_pred);
    ctl.max_depth = reinterpret_cast<int32_t*>(out + b.o_depth);
    ctl.tie_words = reinterpret_cast<uint32_t*>(out + b.o_ties);
    ctl.sample_words = reinterpret_cast<uint32_t*>(out + b.o_sample);
    ctl.depth_sum = reinterpret_cast<int32_t*>(out + b.o_dsum);
    if (b.device_inputs && b.temperature_threshold > 0) {
        ctl.game_moves = reinterpret_cast<int32_t*>(b.d_game_moves);
        ctl.temperature_threshold = b.temperature_threshold;
    }
    int rc = mzhost_launch_fused_move(eng, observations, ctl, true, stream);
    eng->p.root_action = const_cast<int32_t*>(host_legal);
    eng->p.root_children = const_cast<int32_t*>(host_nlegal);
    eng->p.root_to_play = const_cast<int32_t*>(host_to_play);
    if (rc) return rc;
    b.enqueued = m + 1;
    return MZMCTS_OK;
}

int mzmcts_moves_temperature_threshold(mzmcts_engine* eng, int32_t threshold, const int32_t* game_moves, void* stream_) {
    if (!eng || threshold < 0 || (threshold > 0 && !game_moves))
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_temperature_threshold: bad argument");
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (!b.in_flight || !b.device_inputs || b.enqueued != 0)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_temperature_threshold: call right after mzmcts_moves_prepare_device");
    b.temperature_threshold = threshold;
    if (threshold == 0) return MZMCTS_OK;
    const size_t bytes = sizeof(int32_t) * static_cast<size_t>(eng->p.E);
    int rc;
    if (!b.d_game_moves && (rc = dev_alloc(eng, &b.d_game_moves, bytes))) return rc;
    // (pageable source: the copy is staged before the call returns, the caller's array may go)
    MZ_HIP(eng, hipMemcpyAsync(b.d_game_moves, game_moves, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream_)));
    return MZMCTS_OK;
}

int mzmcts_moves_finished(mzmcts_engine* eng, const uint8_t* finished_dev) {
    if (!eng) return MZMCTS_ERR_INVALID;
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (!b.in_flight || !b.device_inputs)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_finished: no device-input batch in flight");
    b.finished = finished_dev;
    return MZMCTS_OK;
}

// what move_inputs_kernel does beyond recording the inputs for the host (see MoveInputsExtra)
static mz::MoveInputsExtra move_extras(mzmcts_engine* eng, bool own_inputs) {
    mzmcts_engine::MoveBatch& b = eng->batch;
    mz::MoveInputsExtra x{};
    if (own_inputs) {
        x.own_legal = eng->own_root_action;
        x.own_nlegal = eng->own_root_children;
        x.own_to_play = eng->own_root_to_play;
    }
    if (b.temperature_threshold > 0) {
        x.game_moves = reinterpret_cast<int32_t*>(b.d_game_moves);
        x.finished = b.finished;
    }
    return x;
}

// ---- a move of a device-input batch searched lock-step (the caller runs the network between the tree launches) --------
int mzmcts_moves_begin_lockstep(mzmcts_engine* eng, void* stream_) {
    if (!eng) return MZMCTS_ERR_INVALID;
    mzmcts_engine::MoveBatch& b = eng->batch;
    const ChainSet& c = b.set[b.cur];
    if (!b.in_flight || !b.device_inputs || b.enqueued >= c.n_moves)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_begin_lockstep: no prepared move left in a device-input batch");
    if (b.lockstep_open) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_begin_lockstep: the move before is still open");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int m = b.enqueued;
    const size_t E = static_cast<size_t>(eng->p.E);
    // the kernel reads the caller's device arrays (the env kernels' outputs), records them for the host, copies them
    // into the engine's own root inputs -- which every later launch of the move reads, a replayed hipGraph included --,
    // steps over the mirrors' pending words and draws the noise row on the device
    mz::TreeParams view = eng->p;
    view.root_action = const_cast<int32_t*>(b.dev_legal);
    view.root_children = const_cast<int32_t*>(b.dev_nlegal);
    view.root_to_play = const_cast<int32_t*>(b.dev_to_play);
    uint8_t* blk = b.d_inputs + b.in2_stride * static_cast<size_t>(m);
    eng->own_root_action = eng->p.root_action;
    eng->own_root_children = eng->p.root_children;
    eng->own_root_to_play = eng->p.root_to_play;
    hipError_t err = mz::launch_move_inputs(view, reinterpret_cast<const uint32_t*>(b.d_in + b.o_skip) + static_cast<size_t>(m) * E,
                                            b.d_stall, reinterpret_cast<const int32_t*>(b.d_in + b.o_limit), m, c.add_noise,
                                            reinterpret_cast<int32_t*>(blk + b.o2_nlegal), reinterpret_cast<int32_t*>(blk + b.o2_to_play),
                                            reinterpret_cast<uint32_t*>(blk + b.o2_words), reinterpret_cast<int32_t*>(blk + b.o2_legal),
                                            move_extras(eng, true), stream);
    b.finished = nullptr;
    if (err != hipSuccess) return hip_fail(eng, err, "move_inputs_kernel");
    b.lockstep_open = true;
    eng->noise_this_search = c.add_noise;
    eng->noise_on_device = false;        // (the batch's collect() accounts for the words: readout is not used)
    eng->skip_applied = true;
    eng->search_begun = true;
    eng->tie_words_applied = true;
    eng->roots_ready = false;
    eng->have_readout = false;
    eng->sim = 0;
    return MZMCTS_OK;
}

int mzmcts_moves_end_lockstep(mzmcts_engine* eng, void* stream_) {
    if (!eng) return MZMCTS_ERR_INVALID;
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (!b.lockstep_open) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_end_lockstep: no lock-step move is open");
    if (!eng->roots_ready || eng->sim != eng->p.S)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_end_lockstep: expand_roots and all simulations must have run");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int m = b.enqueued;
    uint8_t* out = b.d_out + b.out_stride * static_cast<size_t>(m);
    mz::MoveCtl ctl{};
    ctl.temperature = reinterpret_cast<const double*>(b.d_in + b.o_temp);
    ctl.move_index = m;
    ctl.actions = reinterpret_cast<int32_t*>(out + b.o_actions);
    ctl.visits = reinterpret_cast<int32_t*>(out + b.o_visits);
    ctl.root_value_sum = reinterpret_cast<double*>(out + b.o_rvs);
    ctl.root_predicted = reinterpret_cast<float*>(out + b.o_pred);
    ctl.max_depth = reinterpret_cast<int32_t*>(out + b.o_depth);
    ctl.tie_words = reinterpret_cast<uint32_t*>(out + b.o_ties);
    ctl.sample_words = reinterpret_cast<uint32_t*>(out + b.o_sample);
    ctl.depth_sum = reinterpret_cast<int32_t*>(out + b.o_dsum);
    if (b.temperature_threshold > 0) {
        ctl.game_moves = reinterpret_cast<int32_t*>(b.d_game_moves);
        ctl.temperature_threshold = b.temperature_threshold;
    }
    hipError_t err = mz::launch_lockstep_move_finish(eng->p, ctl, stream);
    if (err != hipSuccess) return hip_fail(eng, err, "lockstep_move_finish_kernel");
    // the move's blocks go to the host now, under the batch's remaining searches (collect then finds them there)
    if (b.downloaded == m) {
        MZ_HIP(eng, hipEventRecord(b.move_done, stream));
        MZ_HIP(eng, hipStreamWaitEvent(b.copy_stream, b.move_done, 0));
        MZ_HIP(eng, hipMemcpyAsync(b.h_out + b.out_stride * static_cast<size_t>(m), out, b.out_stride, hipMemcpyDeviceToHost, b.copy_stream));
        MZ_HIP(eng, hipMemcpyAsync(b.h_inputs + b.in2_stride * static_cast<size_t>(m), b.d_inputs + b.in2_stride * static_cast<size_t>(m),
                                   b.in2_stride, hipMemcpyDeviceToHost, b.copy_stream));
        b.downloaded = m + 1;
    }
    b.lockstep_open = false;
    b.enqueued = m + 1;
    eng->search_begun = false;
    eng->roots_ready = false;
    eng->skip_applied = false;
    return MZMCTS_OK;
}

const int32_t* mzmcts_moves_actions(mzmcts_engine* eng, int32_t move) {
    if (!eng || !eng->batch.in_flight || move < 0 || move >= eng->batch.set[eng->batch.cur].n_moves) return nullptr;
    return reinterpret_cast<const int32_t*>(eng->batch.d_out + eng->batch.out_stride * static_cast<size_t>(move) +
                                            eng->batch.o_actions);
}

int mzmcts_moves_ring(mzmcts_engine* eng, void** host_base, int64_t* move_stride, int64_t* offsets, int32_t* capacity) {
    if (!eng || !host_base || !move_stride || !offsets || !capacity) return MZMCTS_ERR_INVALID;
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (!b.h_out) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_ring: no batch has been prepared yet");
    *host_base = b.h_out;
    *move_stride = static_cast<int64_t>(b.out_stride);
    *capacity = b.capacity;
    offsets[0] = static_cast<int64_t>(b.o_actions);
    offsets[1] = static_cast<int64_t>(b.o_visits);
    offsets[2] = static_cast<int64_t>(b.o_rvs);
    offsets[3] = static_cast<int64_t>(b.o_pred);
    offsets[4] = static_cast<int64_t>(b.o_depth);
    return MZMCTS_OK;
}

int mzmcts_moves_inputs_ring(mzmcts_engine* eng, void** host_base, int64_t* move_stride, int64_t* offsets) {
    if (!eng || !host_base || !move_stride || !offsets) return MZMCTS_ERR_INVALID;
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (!b.h_inputs) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_inputs_ring: no device-input batch has been prepared yet");
    *host_base = b.h_inputs;
    *move_stride = static_cast<int64_t>(b.in2_stride);
    offsets[0] = static_cast<int64_t>(b.o2_nlegal);
    offsets[1] = static_cast<int64_t>(b.o2_to_play);
    offsets[2] = static_cast<int64_t>(b.o2_legal);
    return MZMCTS_OK;
}

int mzmcts_moves_collect(mzmcts_engine* eng, int32_t* moves_done, int32_t* actions, int32_t* visits, double* root_value_sum,
                         float* root_predicted, int32_t* max_depth, void* stream_) {
    if (!eng) return MZMCTS_ERR_INVALID;
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (!b.in_flight) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_collect: no batch in flight");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int E = eng->p.E, A = eng->p.A, M = b.enqueued;
    const bool trace = std::getenv("MZMCTS_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = trace ? now() : 0.0;
    MZ_HIP(eng, hipEventRecord(b.done, stream));
    MZ_HIP(eng, hipEventSynchronize(b.done));
    const double t_kernels = trace ? now() : 0.0;
    const int have = b.device_inputs ? std::min(b.downloaded, M) : 0;      // moves downloaded while the batch ran
    if (M > have) {
        MZ_HIP(eng, hipMemcpyAsync(b.h_out + b.out_stride * static_cast<size_t>(have), b.d_out + b.out_stride * static_cast<size_t>(have),
                                   b.out_stride * static_cast<size_t>(M - have), hipMemcpyDeviceToHost, stream));
        if (b.device_inputs)
            MZ_HIP(eng, hipMemcpyAsync(b.h_inputs + b.in2_stride * static_cast<size_t>(have), b.d_inputs + b.in2_stride * static_cast<size_t>(have),
                                       b.in2_stride * static_cast<size_t>(M - have), hipMemcpyDeviceToHost, stream));
    }
    if (have > 0) MZ_HIP(eng, hipStreamSynchronize(b.copy_stream));
    b.downloaded = 0;
    MZ_HIP(eng, hipMemcpyAsync(eng->h_error_flag, eng->p.error_flag, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    MZ_HIP(eng, hipEventRecord(b.done, stream));
    MZ_HIP(eng, hipEventSynchronize(b.done));
    const double t_copied = trace ? now() : 0.0;
    b.in_flight = false;
    ChainSet& c = b.set[b.cur];
    ChainSet& next = b.set[b.cur ^ 1];
    if (eng->h_error_flag[0] != 0)
        return fail(eng, MZMCTS_ERR_INVALID, (eng->h_error_flag[0] & 8) ? "device error flag set: unexpected DPP lane mapping"
                                             : (eng->h_error_flag[0] & 2)
                                                 ? "device error flag set: tree links are inconsistent"
                                                 : "device error flag set: a UCB score was NaN (no maximum to select)");
    auto block = [&](int m, size_t off) { return b.h_out + b.out_stride * static_cast<size_t>(m) + off; };
    ChainSet* sets[2] = {&c, next.drawn ? &next : nullptr};
    std::atomic<int64_t> played_total{0}, depth_total{0};
    if (b.device_inputs) {
        // nothing was drawn ahead on the mirrors: every word of the batch -- noise, tie-breaks, action sampling -- was
        // consumed on the device copies; the mirrors step over them
        auto inputs = [&](int m, size_t off) { return b.h_inputs + b.in2_stride * static_cast<size_t>(m) + off; };
        eng->for_each_env([&](int lo, int hi) {
            int64_t local = 0, local_depth = 0;
            for (int e = lo; e < hi; ++e) {
                int k = 0;
                while (k < M && reinterpret_cast<const int32_t*>(block(k, b.o_actions))[e] >= 0) ++k;
                if (moves_done) moves_done[e] = k;
                uint64_t words = 0;
                for (int m = 0; m < M; ++m) {
                    const bool live = m < k;
                    const size_t me = static_cast<size_t>(m) * E + e;
                    if (actions) actions[me] = live ? reinterpret_cast<const int32_t*>(block(m, b.o_actions))[e] : -1;
                    if (visits)
                        for (int i = 0; i < A; ++i)
                            visits[me * A + i] = live ? reinterpret_cast<const int32_t*>(block(m, b.o_visits))[static_cast<size_t>(e) * A + i] : 0;
                    if (root_value_sum) root_value_sum[me] = live ? reinterpret_cast<const double*>(block(m, b.o_rvs))[e] : 0.0;
                    if (root_predicted) root_predicted[me] = live ? reinterpret_cast<const float*>(block(m, b.o_pred))[e] : 0.f;
                    if (max_depth) max_depth[me] = live ? reinterpret_cast<const int32_t*>(block(m, b.o_depth))[e] : 0;
                    if (live) {
                        words += reinterpret_cast<const uint32_t*>(inputs(m, b.o2_words))[e];
                        words += reinterpret_cast<const uint32_t*>(block(m, b.o_ties))[e];
                        words += reinterpret_cast<const uint32_t*>(block(m, b.o_sample))[e];
                        if (eng->profiling) local_depth += reinterpret_cast<const int32_t*>(block(m, b.o_dsum))[e];
                    }
                }
                // (an env that searched nothing kept the pending words of its mirror: they were not stepped over)
                if (k == 0) eng->lag[e] += reinterpret_cast<const uint32_t*>(c.h_in + b.o_skip)[e];
                eng->behind[e] += words;             // (stepped over when the mirror is next asked for: mirror())
                local += k;
            }
            played_total.fetch_add(local, std::memory_order_relaxed);
            depth_total.fetch_add(local_depth, std::memory_order_relaxed);
        });
        c.drawn = false;
        eng->prof.simulations += played_total.load() * eng->p.S;
        eng->prof.select_depth_sum += depth_total.load();
        return MZMCTS_OK;
    }
    eng->for_each_env([&](int lo, int hi) {
        int64_t local = 0, local_depth = 0;
        for (int e = lo; e < hi; ++e) {
            const bool active = c.nlegal[e] > 0;
            int k = 0;  // moves of this env that were searched: the first k of the batch
            if (active) {
                // played moves are a prefix of the batch (a stall is sticky, a move limit is a prefix): if the last
                // one ran, all of them did -- one read instead of M for nearly every env
                if (M > 0 && reinterpret_cast<const int32_t*>(block(M - 1, b.o_actions))[e] >= 0)
                    k = M;
                else
                    while (k < M && reinterpret_cast<const int32_t*>(block(k, b.o_actions))[e] >= 0) ++k;
            }
            if (moves_done) moves_done[e] = k;
            const bool per_move = actions || visits || root_value_sum || root_predicted || max_depth || eng->profiling;
            for (int m = 0; per_move && m < M; ++m) {
                const bool live = m < k;
                const size_t me = static_cast<size_t>(m) * E + e;
                if (actions) actions[me] = live ? reinterpret_cast<const int32_t*>(block(m, b.o_actions))[e] : -1;
                if (visits)
                    for (int i = 0; i < A; ++i)
                        visits[me * A + i] = live ? reinterpret_cast<const int32_t*>(block(m, b.o_visits))[static_cast<size_t>(e) * A + i] : 0;
                if (root_value_sum) root_value_sum[me] = live ? reinterpret_cast<const double*>(block(m, b.o_rvs))[e] : 0.0;
                if (root_predicted) root_predicted[me] = live ? reinterpret_cast<const float*>(block(m, b.o_pred))[e] : 0.f;
                if (max_depth) max_depth[me] = live ? reinterpret_cast<const int32_t*>(block(m, b.o_depth))[e] : 0;
                if (live && eng->profiling) local_depth += reinterpret_cast<const int32_t*>(block(m, b.o_dsum))[e];
            }
            local += k;
            // The mirror ran ahead over this batch (and over the next one, if it is pre-drawn): put it where the
            // device copy really is, unless everything went as the draws assumed.
            mz::HostStream& s = eng->mirror(e);
            bool redraw_next = next.drawn && next.deferred[e];
            if (active) {
                if (k == 0) {  // nothing was searched (batch collected before its first move ran): undo every draw
                    restore_stream(eng, e, c.start[e], sets, 2);
                    // (rows drawn on top of a running batch start before that batch's last tie-break / sampling
                    // words, which its collect() has confirmed since: step over them again)
                    s.skip(static_cast<uint64_t>(c.tail_ties[e]) + c.tail_sample[e]);
                    eng->lag[e] = (c.tail_ties[e] | c.tail_sample[e]) ? 0u : c.start_lag[e];
                    redraw_next = next.drawn;
                } else {
                    const uint32_t ties = reinterpret_cast<const uint32_t*>(block(k - 1, b.o_ties))[e];
                    const uint32_t sampled = reinterpret_cast<const uint32_t*>(block(k - 1, b.o_sample))[e];
                    const bool as_assumed = next.drawn && !next.deferred[e] && next.nlegal[e] > 0 && k == c.n_moves &&
                                            ties == next.tail_ties[e] && sampled == next.tail_sample[e];
                    if (!as_assumed) {
                        const MoveRecord& r = c.rec[static_cast<size_t>(k - 1) * E + e];
                        if (s.words != r.words) restore_stream(eng, e, r, sets, 2);
                        s.skip(ties);
                        s.skip(sampled);
                        eng->lag[e] = 0;
                        redraw_next = next.drawn;
                    }
                }
            } else if (next.drawn && next.nlegal[e] > 0 && !next.deferred[e]) {
                redraw_next = false;  // inactive here, active next: drawn from the exact state already
            }
            if (redraw_next) draw_env_rows(eng, next, e, false, nullptr);
        }
        played_total.fetch_add(local, std::memory_order_relaxed);
        depth_total.fetch_add(local_depth, std::memory_order_relaxed);
    });
    c.drawn = false;
    if (trace)
        std::fprintf(stderr, "[mzmcts] moves_collect M=%d: wait for kernels %.1f us, download %.1f us, reconcile %.1f us\n", M,
                     t_kernels - t_begin, t_copied - t_kernels, now() - t_copied);
    eng->prof.simulations += played_total.load() * eng->p.S;
    eng->prof.select_depth_sum += depth_total.load();
    return MZMCTS_OK;
}

}  // extern "C"
