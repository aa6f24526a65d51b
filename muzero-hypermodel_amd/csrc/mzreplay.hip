// mzreplay.hip -- device-resident replay store and target kernels (include/mzreplay.h).
//
// Layout (game-slot major, a slot holds one game padded to max_moves = L):
//     obs f32 [G][L+1][obs] | actions i32 [G][L+1] | rewards f64 [G][L+1] | to_play i8 [G][L+1]
//     child_visits f64 [G][L][A] | root_values f64 [G][L] | length i32 [G]
// Both kernels are gather / short-scan work over these rows: HBM-bound, no contraction.
//   priorities_kernel   one workgroup per new game, one thread per position: the td_steps-long discounted
//                       reward sum in the reference's order, |root - target| ** alpha, block max
//   make_batch_kernel   one workgroup per sample: threads 0..U evaluate the U+1 unroll targets, all threads
//                       copy the policy rows and the (stacked) observation planes
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mzreplay.h"

namespace {

struct StoreParams {
    int32_t G, L, A, obs_floats, C, H, W, stacked, td_steps, unroll;
    double alpha;
    float* obs;
    int32_t* actions;
    double* rewards;
    int8_t* to_play;
    double* child_visits;
    double* root_values;
    int32_t* length;
    float* reanalysed;           // [G][L] Reanalyse's fresh root values (float32, as the reference stores them)
    uint8_t* has_reanalysed;     // [G]
    const double* discount_pow;  // [td_steps + 1]
};

// ReplayBuffer.compute_target_value (replay_buffer.py:222-256), fp64, the reference's operation order
__device__ __forceinline__ double target_value(const StoreParams& p, int slot, int n, int index) {
    const double* rewards = p.rewards + static_cast<size_t>(slot) * (p.L + 1);
    const int8_t* to_play = p.to_play + static_cast<size_t>(slot) * (p.L + 1);
    const int bootstrap = index + p.td_steps;
    double value = 0.0;
    if (bootstrap < n && p.has_reanalysed[slot]) {
        // reanalysed_predicted_root_values is a numpy float32 array (replay_buffer.py:226-231, 351-353): under
        // NumPy 2 promotion the bootstrap product and every `value += python_float` stay float32
        float last = p.reanalysed[static_cast<size_t>(slot) * p.L + bootstrap];
        if (to_play[bootstrap] != to_play[index]) last = -last;
        float v32 = last * static_cast<float>(p.discount_pow[p.td_steps]);
        const int stop32 = bootstrap + 1 < n + 1 ? bootstrap + 1 : n + 1;
        for (int k = index + 1, i = 0; k < stop32; ++k, ++i) {
            const double r = rewards[k];
            const double signed_r = (to_play[index] == to_play[index + i]) ? r : -r;
            v32 = v32 + static_cast<float>(signed_r * p.discount_pow[i]);
        }
        return static_cast<double>(v32);
    }
    if (bootstrap < n) {
        double last = p.root_values[static_cast<size_t>(slot) * p.L + bootstrap];
        if (to_play[bootstrap] != to_play[index]) last = -last;
        value = last * p.discount_pow[p.td_steps];
    }
    // reward_history[index + 1 : bootstrap + 1]; the history has n + 1 entries
    const int stop = bootstrap + 1 < n + 1 ? bootstrap + 1 : n + 1;
    for (int k = index + 1, i = 0; k < stop; ++k, ++i) {
        const double r = rewards[k];
        const double signed_r = (to_play[index] == to_play[index + i]) ? r : -r;
        value += signed_r * p.discount_pow[i];
    }
    return value;
}

__global__ __launch_bounds__(256) void priorities_kernel(StoreParams p, const int32_t* __restrict__ slots,
                                                         float* __restrict__ priorities,  // [n][L]
                                                         float* __restrict__ game_priority) {
    __shared__ float block_max[256];
    const int slot = slots[blockIdx.x];
    const int n = p.length[slot];
    float best = -INFINITY;
    for (int i = threadIdx.x; i < p.L; i += blockDim.x) {
        float pr = 0.f;
        if (i < n) {
            const double rv = p.root_values[static_cast<size_t>(slot) * p.L + i];
            pr = static_cast<float>(pow(fabs(rv - target_value(p, slot, n, i)), p.alpha));
            best = fmaxf(best, pr);
        }
        priorities[static_cast<size_t>(blockIdx.x) * p.L + i] = pr;
    }
    block_max[threadIdx.x] = best;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) block_max[threadIdx.x] = fmaxf(block_max[threadIdx.x], block_max[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) game_priority[blockIdx.x] = block_max[0];
}

// GameHistory.get_stacked_observations (self_play.py:514-548) of one position, by the whole workgroup
__device__ __forceinline__ void stacked_observation(const StoreParams& p, int slot, int pos, float* out) {
    const int32_t* actions = p.actions + static_cast<size_t>(slot) * (p.L + 1);
    const int plane = p.H * p.W;
    const int out_channels = p.C + p.stacked * (p.C + 1);
    const float* game_obs = p.obs + static_cast<size_t>(slot) * (p.L + 1) * p.obs_floats;
    for (int t = threadIdx.x; t < out_channels * plane; t += blockDim.x) {
        const int ch = t / plane, px = t - ch * plane;
        float v;
        if (ch < p.C) {
            v = game_obs[static_cast<size_t>(pos) * p.obs_floats + t];
        } else {
            const int k = (ch - p.C) / (p.C + 1);        // k-th past frame: index pos - 1 - k
            const int c = (ch - p.C) - k * (p.C + 1);    // its channel; c == C is the action plane
            const int past = pos - 1 - k;
            if (past < 0)
                v = 0.f;
            else if (c < p.C)
                v = game_obs[static_cast<size_t>(past) * p.obs_floats + c * plane + px];
            else
                v = static_cast<float>(actions[past + 1]);
        }
        out[t] = v;
    }
}

// every position of one game, stacked: the input batch of Reanalyse's initial_inference (replay_buffer.py:335-346)
__global__ __launch_bounds__(128) void game_observations_kernel(StoreParams p, int slot, float* __restrict__ obs_out) {
    const int pos = blockIdx.x;
    if (pos >= p.length[slot]) return;
    stacked_observation(p, slot, pos, obs_out + static_cast<size_t>(pos) * (p.C + p.stacked * (p.C + 1)) * p.H * p.W);
}

__global__ __launch_bounds__(128) void make_batch_kernel(StoreParams p, const int32_t* __restrict__ slots,
                                                         const int32_t* __restrict__ positions,
                                                         const int32_t* __restrict__ absorbing,  // [B][U+1]
                                                         float* __restrict__ obs_out, int64_t* __restrict__ actions_out,
                                                         double* __restrict__ values_out, double* __restrict__ rewards_out,
                                                         double* __restrict__ policies_out, double* __restrict__ scale_out) {
    const int b = blockIdx.x;
    const int slot = slots[b], pos = positions[b];
    const int n = p.length[slot];
    const int U1 = p.unroll + 1;
    const double* rewards = p.rewards + static_cast<size_t>(slot) * (p.L + 1);
    const int32_t* actions = p.actions + static_cast<size_t>(slot) * (p.L + 1);
    // ---- make_target (replay_buffer.py:258-295): one thread per unroll step
    for (int u = threadIdx.x; u < U1; u += blockDim.x) {
        const int cur = pos + u;
        const size_t o = static_cast<size_t>(b) * U1 + u;
        double value = 0.0, reward = 0.0;
        int64_t action;
        if (cur < n) {
            value = target_value(p, slot, n, cur);
            reward = rewards[cur];
            action = actions[cur];
        } else if (cur == n) {
            reward = rewards[cur];
            action = actions[cur];
        } else {
            action = absorbing[o];  // numpy.random.choice(action_space), drawn by the caller in the reference's order
        }
        values_out[o] = value;
        rewards_out[o] = reward;
        actions_out[o] = action;
        const int remaining = n + 1 - pos;  // len(action_history) - game_pos
        scale_out[o] = static_cast<double>(p.unroll < remaining ? p.unroll : remaining);
    }
    // ---- policies: child_visits[cur], or the uniform policy at and past the end of the game
    const double uniform = 1 / static_cast<double>(p.A);
    for (int t = threadIdx.x; t < U1 * p.A; t += blockDim.x) {
        const int u = t / p.A, a = t - u * p.A;
        const int cur = pos + u;
        policies_out[static_cast<size_t>(b) * U1 * p.A + t] =
            cur < n ? p.child_visits[(static_cast<size_t>(slot) * p.L + cur) * p.A + a] : uniform;
    }
    stacked_observation(p, slot, pos, obs_out + static_cast<size_t>(b) * (p.C + p.stacked * (p.C + 1)) * p.H * p.W);
}

}  // namespace

struct mzreplay {
    mzreplay_config cfg{};
    StoreParams p{};
    std::string error;
    std::vector<void*> allocs;
    int64_t bytes = 0;
    int32_t* d_slots = nullptr;      // staging for kernel arguments (capacity / batch sized, grown on demand)
    int32_t* d_positions = nullptr;
    int32_t* d_absorbing = nullptr;
    float* d_priorities = nullptr;
    float* d_game_priority = nullptr;
    size_t staging_games = 0, staging_batch = 0;
};

namespace {
std::string g_create_error;

int fail(mzreplay* s, const std::string& msg) {
    if (s) s->error = msg;
    g_create_error = msg;
    return -1;
}

#define RP_HIP(s, call)                                                                   \
    do {                                                                                  \
        hipError_t err__ = (call);                                                        \
        if (err__ != hipSuccess) return fail(s, std::string(#call) + ": " + hipGetErrorString(err__)); \
    } while (0)

template <typename T>
int dev_alloc(mzreplay* s, T** out, size_t count) {
    void* ptr = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    RP_HIP(s, hipMalloc(&ptr, bytes));
    RP_HIP(s, hipMemset(ptr, 0, bytes));
    s->allocs.push_back(ptr);
    s->bytes += static_cast<int64_t>(bytes);
    *out = static_cast<T*>(ptr);
    return 0;
}
}  // namespace

extern "C" {

const char* mzreplay_last_error(const mzreplay* s) { return s ? s->error.c_str() : g_create_error.c_str(); }

int mzreplay_create(const mzreplay_config* c, mzreplay** out) {
    if (!c || !out || !c->discount_powers) return fail(nullptr, "mzreplay_create: null argument");
    *out = nullptr;
    if (c->capacity <= 0 || c->max_moves <= 0 || c->num_actions <= 0 || c->obs_channels <= 0 || c->obs_height <= 0 ||
        c->obs_width <= 0 || c->stacked_observations < 0 || c->td_steps < 0 || c->num_unroll_steps < 0)
        return fail(nullptr, "mzreplay_create: sizes must be positive");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(nullptr, "mzreplay_create: no HIP device (the replay store has no CPU fallback)");
    mzreplay* s = new mzreplay();
    s->cfg = *c;
    if (hipSetDevice(c->device) != hipSuccess) {
        delete s;
        return fail(nullptr, "mzreplay_create: bad device");
    }
    StoreParams& p = s->p;
    p.G = c->capacity;
    p.L = c->max_moves;
    p.A = c->num_actions;
    p.C = c->obs_channels;
    p.H = c->obs_height;
    p.W = c->obs_width;
    p.obs_floats = p.C * p.H * p.W;
    p.stacked = c->stacked_observations;
    p.td_steps = c->td_steps;
    p.unroll = c->num_unroll_steps;
    p.alpha = c->per_alpha;
    const size_t G = p.G, L = p.L;
    double* d_pow = nullptr;
    int rc = 0;
    rc |= dev_alloc(s, &p.obs, G * (L + 1) * p.obs_floats);
    rc |= dev_alloc(s, &p.actions, G * (L + 1));
    rc |= dev_alloc(s, &p.rewards, G * (L + 1));
    rc |= dev_alloc(s, &p.to_play, G * (L + 1));
    rc |= dev_alloc(s, &p.child_visits, G * L * p.A);
    rc |= dev_alloc(s, &p.root_values, G * L);
    rc |= dev_alloc(s, &p.length, G);
    rc |= dev_alloc(s, &p.reanalysed, G * L);
    rc |= dev_alloc(s, &p.has_reanalysed, G);
    rc |= dev_alloc(s, &d_pow, static_cast<size_t>(p.td_steps) + 1);
    if (rc || hipMemcpy(d_pow, c->discount_powers, sizeof(double) * (p.td_steps + 1), hipMemcpyHostToDevice) != hipSuccess) {
        const std::string msg = s->error.empty() ? "mzreplay_create: device allocation failed" : s->error;
        mzreplay_destroy(s);
        return fail(nullptr, msg);
    }
    p.discount_pow = d_pow;
    s->cfg.discount_powers = nullptr;
    *out = s;
    return 0;
}

void mzreplay_destroy(mzreplay* s) {
    if (!s) return;
    for (void* ptr : s->allocs) (void)hipFree(ptr);
    delete s;
}

int64_t mzreplay_device_bytes(const mzreplay* s) { return s ? s->bytes : 0; }

int mzreplay_add_games(mzreplay* s, int32_t n, const int32_t* slots, const int32_t* lengths, const float* observations,
                       const int32_t* actions, const double* rewards, const int32_t* to_play, const double* child_visits,
                       const double* root_values, float* priorities, float* game_priority, void* stream_) {
    if (!s || !slots || !lengths || !observations || !actions || !rewards || !to_play || !child_visits || !root_values)
        return fail(s, "mzreplay_add_games: null argument");
    if (n <= 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const StoreParams& p = s->p;
    const size_t L = p.L, L1 = L + 1;
    for (int g = 0; g < n; ++g)
        if (slots[g] < 0 || slots[g] >= p.G || lengths[g] < 1 || lengths[g] > p.L)
            return fail(s, "mzreplay_add_games: slot or length out of range");
    if (static_cast<size_t>(n) > s->staging_games) {
        if (dev_alloc(s, &s->d_slots, static_cast<size_t>(n)) || dev_alloc(s, &s->d_priorities, static_cast<size_t>(n) * L) ||
            dev_alloc(s, &s->d_game_priority, static_cast<size_t>(n)))
            return -1;
        s->staging_games = static_cast<size_t>(n);
    }
    std::vector<int8_t> tp8(L1);
    for (int g = 0; g < n; ++g) {
        const size_t slot = static_cast<size_t>(slots[g]);
        const size_t len = static_cast<size_t>(lengths[g]);
        RP_HIP(s, hipMemcpyAsync(p.obs + slot * L1 * p.obs_floats, observations + static_cast<size_t>(g) * L1 * p.obs_floats,
                                 sizeof(float) * (len + 1) * p.obs_floats, hipMemcpyHostToDevice, stream));
        RP_HIP(s, hipMemcpyAsync(p.actions + slot * L1, actions + static_cast<size_t>(g) * L1, sizeof(int32_t) * (len + 1),
                                 hipMemcpyHostToDevice, stream));
        RP_HIP(s, hipMemcpyAsync(p.rewards + slot * L1, rewards + static_cast<size_t>(g) * L1, sizeof(double) * (len + 1),
                                 hipMemcpyHostToDevice, stream));
        for (size_t i = 0; i <= len; ++i) tp8[i] = static_cast<int8_t>(to_play[static_cast<size_t>(g) * L1 + i]);
        RP_HIP(s, hipMemcpyAsync(p.to_play + slot * L1, tp8.data(), len + 1, hipMemcpyHostToDevice, stream));
        RP_HIP(s, hipStreamSynchronize(stream));  // tp8 is reused for the next game
        RP_HIP(s, hipMemcpyAsync(p.child_visits + slot * L * p.A, child_visits + static_cast<size_t>(g) * L * p.A,
                                 sizeof(double) * len * p.A, hipMemcpyHostToDevice, stream));
        RP_HIP(s, hipMemcpyAsync(p.root_values + slot * L, root_values + static_cast<size_t>(g) * L, sizeof(double) * len,
                                 hipMemcpyHostToDevice, stream));
        RP_HIP(s, hipMemcpyAsync(p.length + slot, lengths + g, sizeof(int32_t), hipMemcpyHostToDevice, stream));
        RP_HIP(s, hipMemsetAsync(p.has_reanalysed + slot, 0, 1, stream));
    }
    RP_HIP(s, hipMemcpyAsync(s->d_slots, slots, sizeof(int32_t) * n, hipMemcpyHostToDevice, stream));
    priorities_kernel<<<dim3(n), dim3(256), 0, stream>>>(s->p, s->d_slots, s->d_priorities, s->d_game_priority);
    RP_HIP(s, hipGetLastError());
    if (priorities)
        RP_HIP(s, hipMemcpyAsync(priorities, s->d_priorities, sizeof(float) * n * L, hipMemcpyDeviceToHost, stream));
    if (game_priority)
        RP_HIP(s, hipMemcpyAsync(game_priority, s->d_game_priority, sizeof(float) * n, hipMemcpyDeviceToHost, stream));
    RP_HIP(s, hipStreamSynchronize(stream));
    return 0;
}

int mzreplay_game_observations(mzreplay* s, int32_t slot, int32_t length, float* observations, void* stream_) {
    if (!s || !observations || slot < 0 || slot >= s->p.G || length < 1 || length > s->p.L)
        return fail(s, "mzreplay_game_observations: bad argument");
    game_observations_kernel<<<dim3(length), dim3(128), 0, static_cast<hipStream_t>(stream_)>>>(s->p, slot, observations);
    RP_HIP(s, hipGetLastError());
    return 0;
}

int mzreplay_set_reanalysed(mzreplay* s, int32_t slot, const float* values, int32_t length, void* stream_) {
    if (!s || !values || slot < 0 || slot >= s->p.G || length < 1 || length > s->p.L)
        return fail(s, "mzreplay_set_reanalysed: bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    RP_HIP(s, hipMemcpyAsync(s->p.reanalysed + static_cast<size_t>(slot) * s->p.L, values, sizeof(float) * length,
                             hipMemcpyDefault, stream));
    RP_HIP(s, hipMemsetAsync(s->p.has_reanalysed + slot, 1, 1, stream));
    return 0;
}

int mzreplay_make_batch(mzreplay* s, int32_t batch, const int32_t* slots, const int32_t* positions,
                        const int32_t* absorbing_actions, float* observations, int64_t* actions, double* values,
                        double* rewards, double* policies, double* gradient_scale, void* stream_) {
    if (!s || !slots || !positions || !absorbing_actions || !observations || !actions || !values || !rewards || !policies ||
        !gradient_scale)
        return fail(s, "mzreplay_make_batch: null argument");
    if (batch <= 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const StoreParams& p = s->p;
    const size_t U1 = static_cast<size_t>(p.unroll) + 1;
    for (int b = 0; b < batch; ++b)
        if (slots[b] < 0 || slots[b] >= p.G || positions[b] < 0 || positions[b] > p.L)
            return fail(s, "mzreplay_make_batch: slot or position out of range");
    if (static_cast<size_t>(batch) > s->staging_batch) {
        int32_t* block = nullptr;
        if (dev_alloc(s, &block, static_cast<size_t>(batch) * (2 + U1))) return -1;
        s->d_positions = block;
        s->d_absorbing = block + 2 * static_cast<size_t>(batch);
        s->staging_batch = static_cast<size_t>(batch);
    }
    int32_t* d_slots = s->d_positions + batch;  // [positions B | slots B | absorbing B*(U+1)]
    RP_HIP(s, hipMemcpyAsync(s->d_positions, positions, sizeof(int32_t) * batch, hipMemcpyHostToDevice, stream));
    RP_HIP(s, hipMemcpyAsync(d_slots, slots, sizeof(int32_t) * batch, hipMemcpyHostToDevice, stream));
    RP_HIP(s, hipMemcpyAsync(s->d_absorbing, absorbing_actions, sizeof(int32_t) * batch * U1, hipMemcpyHostToDevice, stream));
    make_batch_kernel<<<dim3(batch), dim3(128), 0, stream>>>(s->p, d_slots, s->d_positions, s->d_absorbing, observations,
                                                            actions, values, rewards, policies, gradient_scale);
    RP_HIP(s, hipGetLastError());
    return 0;
}

}  // extern "C"
