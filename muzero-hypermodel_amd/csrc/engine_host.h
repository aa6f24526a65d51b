// engine_host.h -- host-side state of one engine handle (include/mzmcts.h), shared by the translation units that
// implement the C ABI:
//     mzmcts_capi.hip    create / destroy, the lock-step search steps, readout, the fused whole-move launch, profiling
//     mzmcts_moves.hip   batches of moves without host round trips: RNG mirror bookkeeping, speculation and rewind
//     mzmcts_rng.hip     stand-alone numpy-compatible host streams
//     mzhist.hip         game-history filer (include/mzhist.h)
// The launchers of the kernels (mcts_kernels.hip, fused_narrow.hip) are declared here; the helpers are inline so that
// every translation unit shares one worker pool and one error string.
#pragma once
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
hipError_t launch_select(const TreeParams& p, int sim, float* hidden_out, int64_t* action_out, int queue_trees,
                         hipStream_t stream, const LaunchTiming* timing);
hipError_t launch_root_noise(const TreeParams& p, uint32_t* rng_skip, hipStream_t stream);
hipError_t launch_move_inputs(const TreeParams& p, const uint32_t* rng_skip, const uint8_t* stall, const int32_t* move_limit,
                              int move_index, bool draw_noise, int32_t* nlegal_out, int32_t* to_play_out, uint32_t* words_out,
                              int32_t* legal_out, const MoveInputsExtra& extra, hipStream_t stream);
hipError_t launch_lockstep_move_finish(const TreeParams& p, const MoveCtl& ctl, hipStream_t stream);
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
hipError_t launch_expand_backup_select(const TreeParams& p, int sim, const float* value_logits, const float* reward_logits,
                                       const float* policy_logits, const double* inj_value, const double* inj_reward,
                                       const double* inj_priors, bool injected, float* hidden_out, int64_t* action_out,
                                       hipStream_t stream, const LaunchTiming* timing);
hipError_t launch_copy_slab(const float* src, float* dst, size_t n, hipStream_t stream);
hipError_t launch_seed_streams(uint32_t* keys, int32_t* pos, const uint32_t* seeds, int E, hipStream_t stream);
}  // namespace mz

#ifdef MZ_STAMPS
namespace mz {
hipError_t read_stamp_sums(unsigned long long* out, bool reset);
hipError_t read_stamp_sums_narrow(unsigned long long* out, bool reset);
}
#endif

inline thread_local std::string g_create_error;

enum ProfKind { kProfSelect = 0, kProfBackup = 1, kProfRoot = 2, kProfFused = 3, kProfStep = 4 };
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

inline int host_worker_count(int n_items) {
    int hw = static_cast<int>(std::thread::hardware_concurrency());
    if (hw <= 0) hw = 1;
    if (const char* env = std::getenv("MZMCTS_HOST_THREADS")) hw = std::max(1, std::atoi(env));
    (void)n_items;
    return std::max(0, std::min(hw, 16) - 1);
}

inline WorkerPool& shared_pool() {
    static WorkerPool pool(host_worker_count(0));
    return pool;
}


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

    // host RNG mirrors; lag[e] = words the host stream is ahead of the device copy.  behind[e] = words the device copy
    // consumed that the mirror has not stepped over yet (device-input move batches draw everything on the device: their
    // collect only counts); mirror(e) steps over them the first time anything asks for env e's stream again.
    std::vector<mz::HostStream> streams;
    std::vector<uint32_t> lag;
    std::vector<uint64_t> behind;
    mz::HostStream& mirror(int e) {
        mz::HostStream& s = streams[static_cast<size_t>(e)];
        if (behind[static_cast<size_t>(e)]) {
            s.skip(behind[static_cast<size_t>(e)]);
            behind[static_cast<size_t>(e)] = 0;
        }
        return s;
    }

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
    bool device_noise = false;       // mzmcts_set_device_noise: exploration noise drawn by root_noise_kernel
    bool noise_on_device = false;    // ... and it was, for the search in progress (readout fetches rows + word counts)
    uint32_t* h_noise_words = nullptr;
    int select_queue_trees = 0;      // mzmcts_set_select_queue: 0 = one descent per lane group
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
        uint8_t *h_out = nullptr, *d_out = nullptr;  // [M] output blocks (h_out = h_out_set[host_set])
        // Two pinned download sets, alternating per device-input batch: a batch's blocks are downloaded move by move
        // on `copy_stream` while the batch runs (mzmcts_moves_end_lockstep), and the views the host took of the batch
        // before stay valid while it does.  `downloaded` = moves of the batch in flight whose downloads are queued.
        uint8_t* h_out_set[2] = {nullptr, nullptr};
        uint8_t* h_inputs_set[2] = {nullptr, nullptr};
        int host_set = 0, downloaded = 0;
        hipStream_t copy_stream = nullptr;
        hipEvent_t move_done = nullptr;
        uint8_t* d_stall = nullptr;
        hipEvent_t done = nullptr;
        // batches whose inputs live on the device (mzmcts_moves_prepare_device): the legal sets / players to move the
        // caller's kernels rewrite between the moves, and per move what the search was run with (for the host afterwards)
        bool device_inputs = false;
        const int32_t *dev_legal = nullptr, *dev_nlegal = nullptr, *dev_to_play = nullptr;
        int inputs_capacity = 0;
        size_t in2_stride = 0, o2_nlegal = 0, o2_to_play = 0, o2_words = 0, o2_legal = 0;
        uint8_t *d_inputs = nullptr, *h_inputs = nullptr;   // [M] blocks: nlegal i32[E] | to_play i32[E] | noise words u32[E] | legal i32[E][A]
        // play_game's temperature threshold inside a batch (MoveCtl::game_moves): moves of each env's current game, kept
        // on the device over the batch; `finished` = the env kernels' done flags of the move before (one-shot)
        int32_t temperature_threshold = 0;
        uint8_t* d_game_moves = nullptr;             // i32[E]
        const uint8_t* finished = nullptr;
        bool lockstep_open = false;                  // between mzmcts_moves_begin_lockstep and mzmcts_moves_end_lockstep
    } batch;
    int32_t *own_root_action = nullptr, *own_root_children = nullptr, *own_root_to_play = nullptr;   // (scratch of move_extras)
    bool skip_applied = false;   // the search in progress had its pending words stepped over already (lock-step batch moves)

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


inline int fail(mzmcts_engine* eng, int code, const std::string& msg) {
    if (eng) eng->error = msg;
    g_create_error = msg;
    return code;
}

inline int hip_fail(mzmcts_engine* eng, hipError_t err, const char* what) {
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

inline bool stream_is_capturing(hipStream_t s) {
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


using ChainSet = mzmcts_engine::ChainSet;
using MoveRecord = mzmcts_engine::MoveRecord;

// defined in mzmcts_capi.hip, also used by the move batches (not part of the ABI)
__attribute__((visibility("hidden"))) bool mzhost_use_narrow(const mzmcts_engine* eng);
__attribute__((visibility("hidden"))) int mzhost_launch_fused_move(mzmcts_engine* eng, const float* observations,
                                                                     const mz::MoveCtl& ctl, bool hidden_in_lds,
                                                                     hipStream_t stream);
