// mzmcts_capi.hip -- host side of libmzmcts.so: the C ABI declared in include/mzmcts.h.
//
// Owns the device pools, the per-tree RNG mirrors and the staging buffers; launches the kernels of
// mcts_kernels.hip.  Launch functions do no allocation and no synchronisation (hipGraph-capturable);
// everything blocking says so in the header.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <initializer_list>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mzhist.h"
#include "../../include/mzmcts.h"
#include "fc_net_device.h"
#include "kernel_common.h"
#include "narrow_device.h"
#include "np_legacy_rng.h"
#include "tree_layout.h"

namespace mz {
int default_group_width(int A);
hipError_t launch_fc_inference(const TreeParams& p, const FcNet& net, const float* weights, bool initial, const float* in,
                               const int64_t* action, float* value_logits, float* reward_logits, float* policy_logits,
                               float* hidden_out, hipStream_t stream);
bool plan_fused_layout(const TreeParams& p, const FcNet& net, bool want_hidden_in_lds, size_t lds_limit,
                       FusedLayout* out);
bool narrow_supported(const TreeParams& p, const FcNet& net);
bool plan_narrow_layout(const TreeParams& p, const FcNet& net, size_t lds_limit, NarrowLayout* out);
hipError_t launch_search_fused_narrow(const TreeParams& p, const FcNet& net, const NarrowLayout& lay, const float* weights,
                                      const float* observations, const MoveCtl& ctl, int n_sims, int publish_tree,
                                      hipStream_t stream, const LaunchTiming* timing);
hipError_t launch_fc_inference_narrow(const TreeParams& p, const FcNet& net, const float* weights, bool initial,
                                      const float* in, const int64_t* action, float* value_logits, float* reward_logits,
                                      float* policy_logits, float* hidden_out, hipStream_t stream);
hipError_t launch_search_fused_fc(const TreeParams& p, const FcNet& net, const FusedLayout& lay, const float* weights,
                                  const float* observations, const MoveCtl& ctl, int n_sims, hipStream_t stream,
                                  const LaunchTiming* timing);
hipError_t launch_select(const TreeParams& p, int sim, float* hidden_out, int64_t* action_out, hipStream_t stream,
                         const LaunchTiming* timing);
hipError_t launch_gather_dynamics_input(const TreeParams& p, const int64_t* action, float* out, int plane, int action_space,
                                        hipStream_t stream);
hipError_t launch_expand_roots(const TreeParams& p, const float* value_logits, const float* reward_logits,
                               const float* policy_logits, const float* root_hidden, const double* inj_reward,
                               const double* inj_priors, const double* noise, const uint32_t* rng_skip,
                               bool injected, hipStream_t stream, const LaunchTiming* timing);
hipError_t launch_expand_backup(const TreeParams& p, int sim, const float* value_logits, const float* reward_logits,
                                const float* policy_logits, const double* inj_value, const double* inj_reward,
                                const double* inj_priors, bool injected, hipStream_t stream,
                                const LaunchTiming* timing);
hipError_t launch_copy_slab(const float* src, float* dst, size_t n, hipStream_t stream);
hipError_t launch_seed_streams(uint32_t* keys, int32_t* pos, const uint32_t* seeds, int E, hipStream_t stream);
}  // namespace mz

#ifdef MZ_STAMPS
namespace mz {
hipError_t read_stamp_sums(unsigned long long* out, bool reset);
hipError_t read_stamp_sums_narrow(unsigned long long* out, bool reset);
}
#endif

namespace {
thread_local std::string g_create_error;

enum ProfKind { kProfSelect = 0, kProfBackup = 1, kProfRoot = 2, kProfFused = 3 };
constexpr size_t kLdsPerWorkgroup = 160 * 1024;  // gfx950
struct EventPair {
    hipEvent_t begin, end;
    int kind;
};

// Persistent, process-wide worker pool for the per-env host work of a move (Dirichlet draws, action
// sampling, readout unpacking).  Workers spin briefly on the job generation before sleeping: in a
// self-play loop the next job arrives within microseconds, and a condition-variable wake-up costs more
// than the work itself at a few thousand envs.
class WorkerPool {
  public:
    explicit WorkerPool(int workers) {
        for (int i = 0; i < workers; ++i) threads_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_.store(true, std::memory_order_release);
            generation_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    int size() const { return static_cast<int>(threads_.size()); }
    // body(lo, hi) over [0, n) in small chunks claimed dynamically by the workers and the caller.  The job is
    // complete when every CHUNK is done, not when every worker has reported: on a busy host a worker that is
    // descheduled (or still asleep) delays nothing it has not claimed.
    void run(int n, const std::function<void(int, int)>& body) {
        std::lock_guard<std::mutex> serial(run_mu_);  // one job at a time (engines share the pool)
        const int per_thread = 4;
        int chunk = n / ((size() + 1) * per_thread);
        if (chunk < 16) chunk = 16;
        const uint64_t g = generation_.load(std::memory_order_relaxed) + 1;
        const uint32_t total = static_cast<uint32_t>((n + chunk - 1) / chunk);
        body_.store(&body, std::memory_order_relaxed);
        n_.store(n, std::memory_order_relaxed);
        chunk_.store(chunk, std::memory_order_relaxed);
        total_.store(total, std::memory_order_relaxed);
        done_.store(0, std::memory_order_relaxed);
        next_.store(g << 32, std::memory_order_release);  // publishes the fields above for generation g
        {
            std::lock_guard<std::mutex> lock(mu_);
            generation_.store(g, std::memory_order_release);
        }
        cv_.notify_all();
        work(g);
        while (done_.load(std::memory_order_acquire) < total) __builtin_ia32_pause();
        // Close the job: a worker that read the claim word before this point and was descheduled must fail its
        // compare-exchange instead of claiming a chunk of whatever job comes next.
        next_.store((g << 32) | kClosed, std::memory_order_release);
        body_.store(nullptr, std::memory_order_relaxed);
    }

  private:
    static constexpr uint32_t kClosed = 0xffffffffu;
    // claim chunks of generation g until none is left (or the job has moved on)
    void work(uint64_t g) {
        for (;;) {
            uint64_t cur = next_.load(std::memory_order_acquire);
            if ((cur >> 32) != g) return;
            const uint32_t idx = static_cast<uint32_t>(cur & 0xffffffffu);
            if (idx >= total_.load(std::memory_order_relaxed)) return;
            // a successful exchange proves the claim word still belongs to generation g, hence so do the fields
            if (!next_.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel)) continue;
            const int chunk = chunk_.load(std::memory_order_relaxed);
            const int lo = static_cast<int>(idx) * chunk, hi = std::min(n_.load(std::memory_order_relaxed), lo + chunk);
            (*body_.load(std::memory_order_relaxed))(lo, hi);  // the job cannot complete before this chunk is counted
            done_.fetch_add(1, std::memory_order_release);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            // spin for the next job, then fall back to sleeping
            bool have = false;
            for (int spin = 0; spin < 20000; ++spin) {
                if (generation_.load(std::memory_order_acquire) != seen) {
                    have = true;
                    break;
                }
                __builtin_ia32_pause();
            }
            if (!have) {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait(lock, [&] { return generation_.load(std::memory_order_acquire) != seen; });
            }
            seen = generation_.load(std::memory_order_acquire);
            if (stop_.load(std::memory_order_acquire)) return;
            work(seen);
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_;
    std::atomic<const std::function<void(int, int)>*> body_{nullptr};
    std::atomic<int> n_{0}, chunk_{0};
    std::atomic<uint32_t> total_{0}, done_{0};
    std::atomic<uint64_t> next_{0};
    std::atomic<uint64_t> generation_{0};
    std::atomic<bool> stop_{false};
};

int host_worker_count(int n_items) {
    int hw = static_cast<int>(std::thread::hardware_concurrency());
    if (hw <= 0) hw = 1;
    if (const char* env = std::getenv("MZMCTS_HOST_THREADS")) hw = std::max(1, std::atoi(env));
    (void)n_items;
    return std::max(0, std::min(hw, 16) - 1);
}

WorkerPool& shared_pool() {
    static WorkerPool pool(host_worker_count(0));
    return pool;
}

}  // namespace

struct mzmcts_engine {
    mzmcts_config cfg{};
    mz::TreeParams p{};
    std::string error;
    int sim = 0;            // simulations launched since expand_roots
    bool roots_ready = false;
    bool search_begun = false;
    bool have_readout = false;
    bool owns_hidden = false;
    int64_t device_bytes = 0;

    // device staging for the per-move host inputs
    double* d_noise = nullptr;
    uint32_t* d_skip = nullptr;
    uint32_t* d_seeds = nullptr;
    bool noise_this_search = false;

    // pinned host staging
    int32_t* h_legal = nullptr;      // [E][A]
    int32_t* h_nlegal = nullptr;     // [E]
    int32_t* h_to_play = nullptr;    // [E]
    double* h_noise = nullptr;       // [E][A]
    uint32_t* h_skip = nullptr;      // [E]
    uint8_t* h_slab0 = nullptr;      // [E][block_stride]
    double* h_root_value_sum = nullptr;
    float* h_root_predicted = nullptr;
    int32_t* h_max_depth = nullptr;
    int64_t* h_depth_sum = nullptr;
    uint32_t* h_tie_words = nullptr;
    mz::MinMax* h_min_max = nullptr;
    int32_t* h_error_flag = nullptr;

    // host RNG mirrors; lag[e] = words the host stream is ahead of the device copy
    std::vector<mz::HostStream> streams;
    std::vector<uint32_t> lag;

    // cache of the last readout (sample_actions / search_statistics)
    std::vector<int32_t> last_visits;       // [E][A] per slot
    std::vector<double> last_root_value_sum;
    std::vector<int32_t> last_root_visits;

    // fully-connected network for the in-kernel inference paths
    bool fc_ready = false;
    int fused_variant = MZMCTS_FUSED_AUTO;  // which whole-move kernel mzmcts_search_fused_fc launches
    bool publish_tree = true;               // fused kernels copy the whole tree out (export_tree) or the root only
    bool tree_published = true;             // false after a root-only fused search
    mz::FcNet fc{};
    const float* fc_weights = nullptr;

    // profiling
    bool profiling = false;
    std::vector<EventPair> events;
    size_t events_used = 0;
    mzmcts_profile prof{};

    std::vector<void*> device_allocs;
    std::vector<void*> pinned_allocs;

    // packed per-move upload ([legal | num_legal | to_play | rng_skip | noise]) and per-tree download
    uint8_t* h_upload = nullptr;
    uint8_t* d_upload = nullptr;
    size_t upload_bytes = 0, upload_bytes_no_noise = 0;
    uint8_t* h_download = nullptr;
    uint8_t* d_download = nullptr;
    size_t download_bytes = 0;
    bool tie_words_applied = false;

    // batches of moves queued back to back (mzmcts_moves_*)
    struct MoveRecord {
        int32_t pos, has_gauss;
        double gauss;
        uint64_t words;
    };
    // One batch's host side: the noise rows, and what is needed to take the RNG mirror back to any point of it.
    struct ChainSet {
        int n_moves = 0;
        bool add_noise = false;
        bool drawn = false;
        bool speculative = false;                    // drawn on top of a batch that was still in flight
        uint8_t* h_in = nullptr;                     // pinned [noise M*E*A f64 | skip M*E u32 | temperature E f64 |
                                                     //         limit E i32 | expected ties E u32]
        std::vector<int32_t> legal, nlegal, to_play; // [E][A], [E], [E]
        std::vector<MoveRecord> start;               // [E] mirror state before this batch's first draw
        std::vector<uint32_t> start_lag;             // [E]
        std::vector<MoveRecord> rec;                 // [M][E] mirror state right after move m's noise was drawn
        std::vector<uint8_t> env_twisted;            // [E] the 624-word block was regenerated during the draws
        std::vector<uint64_t> twist_words;           // [E] word count at the first such regeneration
        std::vector<uint32_t> twist_keys;            // [E][624] the block just before it
        std::vector<double> temperature;             // [E]
        std::vector<uint32_t> tail_ties, tail_sample;  // [E] words assumed for the last move of the batch underneath
        std::vector<uint8_t> deferred;               // [E] speculative set: rows not drawn yet (unknown word counts)
    };
    struct MoveBatch {
        int capacity = 0, enqueued = 0;
        bool in_flight = false;                      // set[cur] is uploaded (prepare / submit_next) and not collected
        int cur = 0;
        ChainSet set[2];
        size_t out_stride = 0;                       // bytes of one move's output block
        size_t o_actions = 0, o_visits = 0, o_rvs = 0, o_pred = 0, o_depth = 0, o_ties = 0, o_sample = 0, o_dsum = 0;
        size_t in_bytes = 0, o_skip = 0, o_temp = 0, o_limit = 0, o_expect = 0;
        uint8_t* d_in = nullptr;
        uint8_t *h_out = nullptr, *d_out = nullptr;  // [M] output blocks
        uint8_t* d_stall = nullptr;
        hipEvent_t done = nullptr;
    } batch;

    // pending asynchronous readout (mzmcts_readout_begin)
    hipEvent_t readout_event = nullptr;
    bool readout_pending = false;

    void for_each_env(const std::function<void(int, int)>& body) {
        WorkerPool& pool = shared_pool();
        if (pool.size() > 0 && p.E >= 512)
            pool.run(p.E, body);
        else
            body(0, p.E);
    }
};

namespace {

int fail(mzmcts_engine* eng, int code, const std::string& msg) {
    if (eng) eng->error = msg;
    g_create_error = msg;
    return code;
}

int hip_fail(mzmcts_engine* eng, hipError_t err, const char* what) {
    return fail(eng, MZMCTS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(err));
}

#define MZ_HIP(eng, call)                                         \
    do {                                                          \
        hipError_t err__ = (call);                                \
        if (err__ != hipSuccess) return hip_fail(eng, err__, #call); \
    } while (0)

template <typename T>
int dev_alloc(mzmcts_engine* eng, T** out, size_t count, bool zero = true) {
    void* ptr = nullptr;
    const size_t bytes = count * sizeof(T);
    MZ_HIP(eng, hipMalloc(&ptr, bytes ? bytes : 16));
    if (zero) MZ_HIP(eng, hipMemset(ptr, 0, bytes ? bytes : 16));
    eng->device_allocs.push_back(ptr);
    eng->device_bytes += static_cast<int64_t>(bytes);
    *out = static_cast<T*>(ptr);
    return 0;
}

template <typename T>
int pinned_alloc(mzmcts_engine* eng, T** out, size_t count) {
    void* ptr = nullptr;
    const size_t bytes = count * sizeof(T);
    MZ_HIP(eng, hipHostMalloc(&ptr, bytes ? bytes : 16, hipHostMallocDefault));
    std::memset(ptr, 0, bytes ? bytes : 16);
    eng->pinned_allocs.push_back(ptr);
    *out = static_cast<T*>(ptr);
    return 0;
}

bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) return false;
    return st != hipStreamCaptureStatusNone;
}

// In profiling mode (never while capturing) hand the launcher an event pair that HIP binds to the
// kernel dispatch itself, so the elapsed time is the kernel's own duration.
struct ProfScope {
    mz::LaunchTiming timing;
    ProfScope(mzmcts_engine* eng, hipStream_t s, int kind) {
        if (!eng->profiling || stream_is_capturing(s)) return;
        if (eng->events_used == eng->events.size()) {
            EventPair np{};
            if (hipEventCreate(&np.begin) != hipSuccess || hipEventCreate(&np.end) != hipSuccess) return;
            eng->events.push_back(np);
        }
        EventPair& pair = eng->events[eng->events_used++];
        pair.kind = kind;
        timing.start = pair.begin;
        timing.stop = pair.end;
    }
    const mz::LaunchTiming* get() const { return timing.start ? &timing : nullptr; }
};

}  // namespace

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
    if (c->group_width != 0) {
        const int g = c->group_width;
        if (g < p.group || g > 64 || (g & (g - 1)) != 0) {
            delete eng;
            return fail(nullptr, MZMCTS_ERR_INVALID, "mzmcts_create: group_width must be a power of two in [pow2(A), 64]");
        }
        p.group = g;
    }
    p.links_offset = 16u * static_cast<uint32_t>(A);
    p.block_stride = mz::round_up(32u * static_cast<uint32_t>(A), 64u);
    p.discount = c->discount;
    p.noise_frac = c->root_exploration_fraction;

    int rc = 0;
    auto cleanup_on = [&](int code) {
        if (code != 0) {
            g_create_error = eng->error;
            mzmcts_destroy(eng);
        }
        return code;
    };
    const size_t K = static_cast<size_t>(S) + 1;
    if ((rc = dev_alloc(eng, &p.blocks, K * E * p.block_stride))) return cleanup_on(rc);
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
        const size_t o_err = o_tw + sizeof(uint32_t) * E;
        eng->download_bytes = o_err + sizeof(int32_t) * 4;
        if ((rc = dev_alloc(eng, &eng->d_download, eng->download_bytes))) return cleanup_on(rc);
        if ((rc = pinned_alloc(eng, &eng->h_download, eng->download_bytes))) return cleanup_on(rc);
        p.root_value_sum = reinterpret_cast<double*>(eng->d_download + o_rvs);
        p.min_max = reinterpret_cast<mz::MinMax*>(eng->d_download + o_mm);
        p.depth_sum = reinterpret_cast<int64_t*>(eng->d_download + o_ds);
        p.root_predicted = reinterpret_cast<float*>(eng->d_download + o_pred);
        p.max_depth = reinterpret_cast<int32_t*>(eng->d_download + o_md);
        p.tie_words = reinterpret_cast<uint32_t*>(eng->d_download + o_tw);
        p.error_flag = reinterpret_cast<int32_t*>(eng->d_download + o_err);
        eng->h_root_value_sum = reinterpret_cast<double*>(eng->h_download + o_rvs);
        eng->h_min_max = reinterpret_cast<mz::MinMax*>(eng->h_download + o_mm);
        eng->h_depth_sum = reinterpret_cast<int64_t*>(eng->h_download + o_ds);
        eng->h_root_predicted = reinterpret_cast<float*>(eng->h_download + o_pred);
        eng->h_max_depth = reinterpret_cast<int32_t*>(eng->h_download + o_md);
        eng->h_tie_words = reinterpret_cast<uint32_t*>(eng->h_download + o_tw);
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

    if ((rc = pinned_alloc(eng, &eng->h_slab0, static_cast<size_t>(E) * p.block_stride))) return cleanup_on(rc);

    eng->streams.resize(E);
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
    for (void* ptr : eng->device_allocs) (void)hipFree(ptr);
    for (void* ptr : eng->pinned_allocs) (void)hipHostFree(ptr);
    delete eng;
}

int mzmcts_seed(mzmcts_engine* eng, const uint32_t* seeds, void* stream_) {
    if (!eng || !seeds) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_seed: null argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int E = eng->p.E;
    eng->for_each_env([&](int lo, int hi) {
        for (int e = lo; e < hi; ++e) eng->streams[e].seed(seeds[e]);
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
    const mz::HostStream& s = eng->streams[env];  // the host mirror is authoritative between calls
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
    eng->for_each_env([&](int lo, int hi) {
        for (int e = lo; e < hi; ++e) {
            const int n = eng->h_nlegal[e];
            double* row = eng->h_noise + static_cast<size_t>(e) * A;
            for (int i = 0; i < A; ++i) row[i] = 0.0;
            if (n == 0) {
                eng->h_skip[e] = 0;  // inactive: the stream is left alone, pending lag is kept
                continue;
            }
            if (add_noise) {
                mz::HostStream& s = eng->streams[e];
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
    MZ_HIP(eng, hipMemcpyAsync(eng->d_upload, eng->h_upload, add_noise ? eng->upload_bytes : eng->upload_bytes_no_noise,
                               hipMemcpyHostToDevice, stream));
    if (trace)
        std::fprintf(stderr, "[mzmcts] begin_search E=%d: host prep+dirichlet %.1f us, memcpy+enqueue %.1f us\n", E,
                     t_dirichlet - t_start, now() - t_dirichlet);
    eng->search_begun = true;
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
                                            inj_priors, eng->noise_this_search ? eng->d_noise : nullptr, eng->d_skip,
                                            injected, stream, scope.get()));
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
    MZ_HIP(eng, mz::launch_select(eng->p, eng->sim, eng->p.H > 0 ? parent_hidden_out : nullptr, action_out, stream,
                                  scope.get()));
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
    MZ_HIP(eng, hipMemcpyAsync(eng->h_slab0, p.blocks, static_cast<size_t>(p.E) * p.block_stride, hipMemcpyDeviceToHost, stream));
    MZ_HIP(eng, hipMemcpyAsync(eng->h_download, eng->d_download, eng->download_bytes, hipMemcpyDeviceToHost, stream));
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
            const uint8_t* blk = eng->h_slab0 + static_cast<size_t>(e) * p.block_stride;
            const mz::ChildStats* st = reinterpret_cast<const mz::ChildStats*>(blk);
            const mz::ChildLinks* lk = reinterpret_cast<const mz::ChildLinks*>(blk + p.links_offset);
            const int n = eng->h_nlegal[e];
            const bool is_active = n > 0;
            for (int i = 0; i < A; ++i) {
                const size_t o = static_cast<size_t>(e) * A + i;
                const bool live = i < n;
                eng->last_visits[o] = live ? lk[i].visits : 0;
                if (out) {
                    if (out->visits) out->visits[o] = live ? lk[i].visits : 0;
                    if (out->child_value_sum) out->child_value_sum[o] = live ? st[i].value_sum : 0.0;
                    if (out->child_prior) out->child_prior[o] = live ? st[i].prior : 0.0;
                    if (out->child_reward) out->child_reward[o] = live ? static_cast<double>(lk[i].reward) : 0.0;
                    if (out->child_expanded) out->child_expanded[o] = (live && lk[i].child_node >= 0) ? 1 : 0;
                }
            }
            eng->last_root_value_sum[e] = is_active ? eng->h_root_value_sum[e] : 0.0;
            eng->last_root_visits[e] = is_active ? sims : 0;
            if (is_active) {
                // the tie-breaks ran on the device copy of the stream: bring the host mirror level (once)
                if (first) eng->streams[e].skip(eng->h_tie_words[e]);
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
            mz::HostStream& s = eng->streams[e];
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
                const int slot = path[static_cast<size_t>(d) * E + e] & 0xffff;
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
    std::vector<uint8_t> buf(static_cast<size_t>(K) * p.block_stride);
    MZ_HIP(eng, hipMemcpy2DAsync(buf.data(), p.block_stride, p.blocks + static_cast<size_t>(env) * p.block_stride,
                                 static_cast<size_t>(p.E) * p.block_stride, p.block_stride, K, hipMemcpyDeviceToHost,
                                 stream));
    MZ_HIP(eng, hipStreamSynchronize(stream));
    const int n_root = eng->h_nlegal[env];
    for (int k = 0; k < K; ++k) {
        const uint8_t* blk = buf.data() + static_cast<size_t>(k) * p.block_stride;
        const mz::ChildStats* st = reinterpret_cast<const mz::ChildStats*>(blk);
        const mz::ChildLinks* lk = reinterpret_cast<const mz::ChildLinks*>(blk + p.links_offset);
        const bool written = k <= eng->sim;
        for (int i = 0; i < A; ++i) {
            const size_t o = static_cast<size_t>(k) * A + i;
            const bool live = written && (k > 0 || i < n_root);
            if (visits) visits[o] = live ? lk[i].visits : 0;
            if (value_sum) value_sum[o] = live ? st[i].value_sum : 0.0;
            if (prior) prior[o] = live ? st[i].prior : 0.0;
            if (reward) reward[o] = live ? static_cast<double>(lk[i].reward) : 0.0;
            if (child_node) child_node[o] = live ? lk[i].child_node : -1;
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
static bool use_narrow(const mzmcts_engine* eng) {
    if (!eng->fc_ready || eng->fused_variant == MZMCTS_FUSED_GENERIC) return false;
    mz::NarrowLayout lay{};
    return mz::narrow_supported(eng->p, eng->fc) && mz::plan_narrow_layout(eng->p, eng->fc, kLdsPerWorkgroup, &lay);
}

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
    if (use_narrow(eng)) return MZMCTS_FUSED_NARROW;
    mz::FusedLayout lay{};
    return mz::plan_fused_layout(eng->p, eng->fc, true, kLdsPerWorkgroup, &lay) ? MZMCTS_FUSED_GENERIC : 0;
}

int mzmcts_fc_initial_inference(mzmcts_engine* eng, const float* observations, float* value_logits, float* reward_logits,
                                float* policy_logits, float* hidden_out, void* stream) {
    if (!eng || !observations || !value_logits || !reward_logits || !policy_logits || !hidden_out)
        return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_fc_initial_inference: null argument");
    if (!eng->fc_ready) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_fc_initial_inference: call mzmcts_fc_configure first");
    if (use_narrow(eng))
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
    if (use_narrow(eng))
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
    if (use_narrow(eng)) {
        mz::NarrowLayout nl{};
        mz::plan_narrow_layout(eng->p, eng->fc, kLdsPerWorkgroup, &nl);
        return nl.total_bytes;
    }
    mz::FusedLayout lay{};
    if (!mz::plan_fused_layout(eng->p, eng->fc, hidden_in_lds != 0, kLdsPerWorkgroup, &lay)) return 0;
    return lay.total_bytes;
}

// One whole-move kernel (narrow or generic, see use_narrow) with the given per-move control block.
static int launch_fused_move(mzmcts_engine* eng, const float* observations, const mz::MoveCtl& ctl, bool hidden_in_lds,
                             hipStream_t stream) {
    if (use_narrow(eng)) {
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

int mzmcts_search_fused_fc(mzmcts_engine* eng, const float* observations, int32_t hidden_in_lds, void* stream_) {
    if (!eng || !observations) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_search_fused_fc: null argument");
    if (!eng->fc_ready) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_search_fused_fc: call mzmcts_fc_configure first");
    if (!eng->search_begun) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_search_fused_fc called before begin_search");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    mz::MoveCtl ctl{};
    ctl.noise = eng->noise_this_search ? eng->d_noise : nullptr;
    ctl.rng_skip = eng->d_skip;
    int rc = launch_fused_move(eng, observations, ctl, hidden_in_lds != 0, stream);
    if (rc) return rc;
    eng->roots_ready = true;
    eng->sim = eng->p.S;
    return MZMCTS_OK;
}


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
using ChainSet = mzmcts_engine::ChainSet;
using MoveRecord = mzmcts_engine::MoveRecord;

// Take env e's mirror back (or forward) to `target`, a state recorded while drawing sets[0..n_sets) (oldest
// first).  Only the first regeneration of a set is backed up, so: if a backed-up block covers the target, use
// it directly; otherwise start from the latest backup before the target and walk forward.
static void restore_stream(mzmcts_engine* eng, int e, const MoveRecord& target, ChainSet* const* sets, int n_sets) {
    mz::HostStream& s = eng->streams[e];
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
    if ((rc = pinned_alloc(eng, &b.h_out, b.out_stride * M))) return rc;
    if (!b.d_stall && (rc = dev_alloc(eng, &b.d_stall, E))) return rc;
    if (!b.done) MZ_HIP(eng, hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
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
    MZ_HIP(eng, hipMemcpy(b.h_out, b.d_out, b.out_stride * M, hipMemcpyDeviceToHost));
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
    mz::HostStream& s = eng->streams[e];
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
    fill_set(eng, c, n_moves, legal, num_legal, to_play, add_noise, temperature);
    eng->for_each_env([&](int lo, int hi) {
        for (int e = lo; e < hi; ++e) draw_env_rows(eng, c, e, false, nullptr);
    });
    c.drawn = true;
    c.speculative = false;
    return upload_set(eng, c, static_cast<hipStream_t>(stream_));
}

int mzmcts_moves_predraw_next(mzmcts_engine* eng, int32_t n_moves, const int32_t* legal, const int32_t* num_legal,
                              const int32_t* to_play, int32_t add_noise, const double* temperature) {
    int rc = check_move_inputs(eng, n_moves, legal, num_legal, to_play, temperature, "mzmcts_moves_predraw_next");
    if (rc) return rc;
    mzmcts_engine::MoveBatch& b = eng->batch;
    if (!b.in_flight) return fail(eng, MZMCTS_ERR_INVALID, "mzmcts_moves_predraw_next: no batch in flight (use mzmcts_moves_prepare)");
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
            eng->streams[e].skip(static_cast<uint64_t>(c.tail_ties[e]) + c.tail_sample[e]);
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
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const size_t E = static_cast<size_t>(eng->p.E), A = static_cast<size_t>(eng->p.A);
    const int m = b.enqueued;
    uint8_t* out = b.d_out + b.out_stride * static_cast<size_t>(m);
    mz::MoveCtl ctl{};
    ctl.noise = c.add_noise ? reinterpret_cast<const double*>(b.d_in) + static_cast<size_t>(m) * E * A : nullptr;
    ctl.rng_skip = reinterpret_cast<const uint32_t*>(b.d_in + b.o_skip) + static_cast<size_t>(m) * E;
    ctl.temperature = reinterpret_cast<const double*>(b.d_in + b.o_temp);
    ctl.move_limit = reinterpret_cast<const int32_t*>(b.d_in + b.o_limit);
    ctl.stall = b.d_stall;
    ctl.move_index = m;
    ctl.expected_ties = m > 0 ? reinterpret_cast<const uint32_t*>(b.d_in + b.o_expect) : nullptr;
    ctl.actions = reinterpret_cast<int32_t*>(out + b.o_actions);
    ctl.visits = reinterpret_cast<int32_t*>(out + b.o_visits);
    ctl.root_value_sum = reinterpret_cast<double*>(out + b.o_rvs);
    ctl.root_predicted = reinterpret_cast<float*>(out + b.o_pred);
    ctl.max_depth = reinterpret_cast<int32_t*>(out + b.o_depth);
    ctl.tie_words = reinterpret_cast<uint32_t*>(out + b.o_ties);
    ctl.sample_words = reinterpret_cast<uint32_t*>(out + b.o_sample);
    ctl.depth_sum = reinterpret_cast<int32_t*>(out + b.o_dsum);
    int rc = launch_fused_move(eng, observations, ctl, true, stream);
    if (rc) return rc;
    b.enqueued = m + 1;
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
    if (M > 0) MZ_HIP(eng, hipMemcpyAsync(b.h_out, b.d_out, b.out_stride * static_cast<size_t>(M), hipMemcpyDeviceToHost, stream));
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
            mz::HostStream& s = eng->streams[e];
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

// ---- stand-alone host streams ----------------------------------------------------------------------
struct mzmcts_rng {
    mz::HostStream s;
};

mzmcts_rng* mzmcts_rng_create(uint32_t seed) {
    auto* r = new mzmcts_rng();
    r->s.seed(seed);
    return r;
}
void mzmcts_rng_destroy(mzmcts_rng* r) { delete r; }
void mzmcts_rng_reseed(mzmcts_rng* r, uint32_t seed) { r->s.seed(seed); }
uint32_t mzmcts_rng_next_u32(mzmcts_rng* r) { return r->s.u32(); }
double mzmcts_rng_random_sample(mzmcts_rng* r) { return r->s.uniform(); }
uint32_t mzmcts_rng_choice(mzmcts_rng* r, uint32_t n) { return r->s.below(n); }
int32_t mzmcts_rng_choice_p(mzmcts_rng* r, const double* p, int32_t n) { return r->s.choice_p(p, n); }
void mzmcts_rng_choice_p_many(mzmcts_rng* r, const double* p, int32_t n, int32_t count, int32_t* out) {
    for (int32_t i = 0; i < count; ++i) out[i] = r->s.choice_p(p, n);
}
int32_t mzmcts_rng_choice_priorities(mzmcts_rng* r, const float* priorities, int32_t n, float* prob_out) {
    // position_probs = priorities / sum(priorities): a left-to-right float32 sum (Python's sum over float32
    // scalars), a float32 division per entry, then RandomState.choice(n, p=position_probs)
    float total = 0.f;
    for (int32_t i = 0; i < n; ++i) total = total + priorities[i];
    std::vector<double> p(static_cast<size_t>(n));
    for (int32_t i = 0; i < n; ++i) p[i] = static_cast<double>(priorities[i] / total);
    const int32_t idx = r->s.choice_p(p.data(), n);
    if (prob_out && idx >= 0 && idx < n) *prob_out = priorities[idx] / total;
    return idx;
}
void mzmcts_rng_dirichlet(mzmcts_rng* r, double alpha, int32_t k, double* out) { r->s.dirichlet(alpha, k, out); }
void mzmcts_rng_export(const mzmcts_rng* r, uint32_t* key, int32_t* pos, int32_t* has_gauss, double* cached) {
    std::memcpy(key, r->s.key, sizeof(r->s.key));
    *pos = r->s.pos;
    *has_gauss = r->s.has_gauss;
    *cached = r->s.gauss;
}
void mzmcts_rng_import(mzmcts_rng* r, const uint32_t* key, int32_t pos, int32_t has_gauss, double cached) {
    std::memcpy(r->s.key, key, sizeof(r->s.key));
    r->s.pos = pos;
    r->s.has_gauss = has_gauss;
    r->s.gauss = cached;
}
int32_t mzmcts_rng_select_action(mzmcts_rng* r, const int32_t* visits, int32_t n, double temperature) {
    return r->s.select_action(visits, n, temperature);
}

}  // extern "C"


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
    std::atomic<bool> overflow{false};
    for_envs([&](int lo, int hi) {
        int local_longest = 0;
        for (int e = lo; e < hi; ++e) {
            int len = h->length[e], count = 0;
            const int k = std::min(mv->moves_done[e], M);
            for (int m = 0; m < k; ++m) {
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
            const int32_t* legal = mv->legal + static_cast<size_t>(e) * A;
            const int n_legal = mv->num_legal[e];
            int len = h->length[e];
            int out_slot = h->fin_offset[e];
            const int k = std::min(mv->moves_done[e], M);
            for (int m = 0; m < k; ++m) {
                const size_t me = static_cast<size_t>(m) * E + e;
                const int32_t* visits = reinterpret_cast<const int32_t*>(at(mv->visits, mv->visits_stride, m)) + static_cast<size_t>(e) * A;
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
