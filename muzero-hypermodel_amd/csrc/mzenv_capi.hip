// mzenv_capi.hip -- host side of the device-resident environments (include/mzenv.h).
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/mzenv.h"
#include "env_layout.h"
#include "np_legacy_rng.h"

namespace mz {
hipError_t launch_env_reset(const EnvParams& p, const uint8_t* mask, hipStream_t stream);
hipError_t launch_env_step(const EnvParams& p, const int32_t* actions, float* reward, uint8_t* done, hipStream_t stream);
hipError_t launch_env_advance(const EnvParams& p, const int32_t* actions, float* reward, uint8_t* done, float* obs_after,
                              float* obs_next, int32_t* legal, int32_t* num_legal, int32_t* to_play, hipStream_t stream);
hipError_t launch_env_observe(const EnvParams& p, float* obs, int32_t* legal, int32_t* num_legal, int32_t* to_play,
                              hipStream_t stream);
hipError_t launch_seed_streams(uint32_t* keys, int32_t* pos, const uint32_t* seeds, int E, hipStream_t stream);
}  // namespace mz

struct mzenv {
    mz::EnvParams p{};
    int32_t players = 1;
    int32_t shape[3] = {1, 1, 1};
    int32_t device = 0;
    std::string error;
    std::vector<void*> allocs;
};

namespace {
thread_local std::string g_env_error;
int env_fail(mzenv* env, int code, const std::string& msg) {
    if (env) env->error = msg;
    g_env_error = msg;
    return code;
}
#define MZENV_HIP(env, call)                                                                                 \
    do {                                                                                                     \
        hipError_t err__ = (call);                                                                           \
        if (err__ != hipSuccess) return env_fail(env, -2, std::string(#call) + ": " + hipGetErrorString(err__)); \
    } while (0)

template <typename T>
int env_alloc(mzenv* env, T** out, size_t count) {
    void* ptr = nullptr;
    MZENV_HIP(env, hipMalloc(&ptr, count * sizeof(T) ? count * sizeof(T) : 16));
    MZENV_HIP(env, hipMemset(ptr, 0, count * sizeof(T) ? count * sizeof(T) : 16));
    env->allocs.push_back(ptr);
    *out = static_cast<T*>(ptr);
    return 0;
}
}  // namespace

extern "C" {

const char* mzenv_last_error(const mzenv* env) { return env ? env->error.c_str() : g_env_error.c_str(); }

int mzenv_create(int32_t game, int32_t num_envs, int32_t device, const uint32_t* seeds, mzenv** out) {
    if (!out || !seeds || num_envs <= 0 || game < 0 || game > 2) return env_fail(nullptr, -1, "mzenv_create: bad argument");
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return env_fail(nullptr, -2, "mzenv_create: no HIP device available; the device environments have no CPU fallback");
    if (device < 0 || device >= n_dev) return env_fail(nullptr, -1, "mzenv_create: bad device ordinal");
    if (hipSetDevice(device) != hipSuccess) return env_fail(nullptr, -2, "hipSetDevice failed");
    auto* env = new mzenv();
    env->device = device;
    mz::EnvParams& p = env->p;
    p.game = game;
    p.E = num_envs;
    int rc = 0;
    auto bail = [&](int code) {
        g_env_error = env->error;
        mzenv_destroy(env);
        return code;
    };
    if (game == MZENV_CARTPOLE) {
        p.A = 2;
        p.cells = 0;
        p.obs_floats = 4;
        env->players = 1;
        env->shape[0] = 1, env->shape[1] = 1, env->shape[2] = 4;
        if ((rc = env_alloc(env, &p.state, static_cast<size_t>(num_envs) * 4))) return bail(rc);
        if ((rc = env_alloc(env, &p.steps, num_envs))) return bail(rc);
        if ((rc = env_alloc(env, &p.mt_key, static_cast<size_t>(num_envs) * mz::kMtN))) return bail(rc);
        if ((rc = env_alloc(env, &p.mt_pos, num_envs))) return bail(rc);
        uint32_t* d_seeds = nullptr;
        if ((rc = env_alloc(env, &d_seeds, num_envs))) return bail(rc);
        hipError_t err = hipMemcpy(d_seeds, seeds, sizeof(uint32_t) * num_envs, hipMemcpyHostToDevice);
        if (err == hipSuccess) err = mz::launch_seed_streams(p.mt_key, p.mt_pos, d_seeds, num_envs, nullptr);
        if (err == hipSuccess) err = hipDeviceSynchronize();
        if (err != hipSuccess) return bail(env_fail(env, -2, std::string("seeding: ") + hipGetErrorString(err)));
    } else {
        const bool ttt = game == MZENV_TICTACTOE;
        p.A = ttt ? 9 : 7;
        p.cells = ttt ? 9 : 42;
        p.obs_floats = 3 * p.cells;
        env->players = 2;
        env->shape[0] = 3, env->shape[1] = ttt ? 3 : 6, env->shape[2] = ttt ? 3 : 7;
        if ((rc = env_alloc(env, &p.board, static_cast<size_t>(num_envs) * p.cells))) return bail(rc);
        if ((rc = env_alloc(env, &p.player, num_envs))) return bail(rc);
    }
    *out = env;
    return 0;
}

void mzenv_destroy(mzenv* env) {
    if (!env) return;
    (void)hipSetDevice(env->device);
    (void)hipDeviceSynchronize();
    for (void* ptr : env->allocs) (void)hipFree(ptr);
    delete env;
}

int mzenv_shape(const mzenv* env, int32_t* num_actions, int32_t* num_players, int32_t* obs_shape3) {
    if (!env) return -1;
    if (num_actions) *num_actions = env->p.A;
    if (num_players) *num_players = env->players;
    if (obs_shape3)
        for (int i = 0; i < 3; ++i) obs_shape3[i] = env->shape[i];
    return 0;
}

int mzenv_reset(mzenv* env, const uint8_t* mask, void* stream) {
    if (!env) return -1;
    MZENV_HIP(env, mz::launch_env_reset(env->p, mask, static_cast<hipStream_t>(stream)));
    return 0;
}

int mzenv_step(mzenv* env, const int32_t* actions, float* reward_out, uint8_t* done_out, void* stream) {
    if (!env || !actions || !reward_out || !done_out) return env_fail(env, -1, "mzenv_step: null argument");
    MZENV_HIP(env, mz::launch_env_step(env->p, actions, reward_out, done_out, static_cast<hipStream_t>(stream)));
    return 0;
}

int mzenv_observe(mzenv* env, float* obs_out, int32_t* legal_out, int32_t* num_legal_out, int32_t* to_play_out,
                  void* stream) {
    if (!env || !obs_out || !legal_out || !num_legal_out || !to_play_out)
        return env_fail(env, -1, "mzenv_observe: null argument");
    MZENV_HIP(env, mz::launch_env_observe(env->p, obs_out, legal_out, num_legal_out, to_play_out,
                                          static_cast<hipStream_t>(stream)));
    return 0;
}

int mzenv_advance(mzenv* env, const int32_t* actions, float* reward_out, uint8_t* done_out, float* obs_after_out,
                  float* obs_next_out, int32_t* legal_out, int32_t* num_legal_out, int32_t* to_play_out, void* stream_) {
    if (!env || !actions || !reward_out || !done_out || !obs_after_out || !obs_next_out || !legal_out || !num_legal_out ||
        !to_play_out)
        return env_fail(env, -1, "mzenv_advance: null argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MZENV_HIP(env, mz::launch_env_advance(env->p, actions, reward_out, done_out, obs_after_out, obs_next_out, legal_out,
                                          num_legal_out, to_play_out, stream));
    return 0;
}

}  // extern "C"
