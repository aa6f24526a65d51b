// env_kernels.hip -- device-resident batched game environments (include/mzenv.h).
//
// One thread per env: a move is a handful of byte operations per game, far below any roofline that
// matters; what these kernels buy is that E games advance without E host-side Python calls per move.
// Rules restate the reference's in-repo envs (games/tictactoe.py:242-305, games/connect4.py:219-304)
// with the Game wrappers' reward scaling; CartPole restates the classic-control equations (unpinned).
#include <hip/hip_runtime.h>

#include "env_layout.h"
#include "np_legacy_rng.h"

namespace mz {


// ---- tic-tac-toe ------------------------------------------------------------------------------------
__device__ __forceinline__ bool ttt_winner(const int8_t* b, int p) {
    const int t = 3 * p;
    for (int i = 0; i < 3; ++i) {
        if (b[3 * i] + b[3 * i + 1] + b[3 * i + 2] == t) return true;
        if (b[i] + b[i + 3] + b[i + 6] == t) return true;
    }
    return (b[0] + b[4] + b[8] == t) || (b[2] + b[4] + b[6] == t);
}

// ---- connect four (row 0 = bottom) ---------------------------------------------------------------------
__device__ __forceinline__ bool c4_winner(const int8_t* b, int p) {
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 7; ++c) {
            if (b[r * 7 + c] != p) continue;
            if (c + 3 < 7 && b[r * 7 + c + 1] == p && b[r * 7 + c + 2] == p && b[r * 7 + c + 3] == p) return true;
            if (r + 3 < 6 && b[(r + 1) * 7 + c] == p && b[(r + 2) * 7 + c] == p && b[(r + 3) * 7 + c] == p) return true;
            if (r + 3 < 6 && c + 3 < 7 && b[(r + 1) * 7 + c + 1] == p && b[(r + 2) * 7 + c + 2] == p &&
                b[(r + 3) * 7 + c + 3] == p)
                return true;
            if (r - 3 >= 0 && c + 3 < 7 && b[(r - 1) * 7 + c + 1] == p && b[(r - 2) * 7 + c + 2] == p &&
                b[(r - 3) * 7 + c + 3] == p)
                return true;
        }
    return false;
}

__device__ __forceinline__ double mt_uniform(uint32_t* key, int32_t* pos) {
    const int32_t a = static_cast<int32_t>(mt_next(key, pos) >> 5);
    const int32_t b = static_cast<int32_t>(mt_next(key, pos) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

__device__ __forceinline__ void env_reset_one(const EnvParams& p, int e) {
    if (p.game == 0) {
        // numpy RandomState(seed).uniform(-0.05, 0.05, size=4): low + (high - low) * random_sample()
        uint32_t* key = p.mt_key + static_cast<size_t>(e) * kMtN;
        int32_t pos = p.mt_pos[e];
        for (int i = 0; i < 4; ++i) p.state[4 * e + i] = -0.05 + (0.05 - -0.05) * mt_uniform(key, &pos);
        p.mt_pos[e] = pos;
        p.steps[e] = 0;
    } else {
        int8_t* b = p.board + static_cast<size_t>(e) * p.cells;
        for (int i = 0; i < p.cells; ++i) b[i] = 0;
        p.player[e] = 1;
    }
}

// One move of env e; returns whether the game ended.  a < 0: the env is left alone this move (e.g. its search
// was not run) -- nothing happened.
__device__ __forceinline__ bool env_step_one(const EnvParams& p, int e, int a, float* __restrict__ reward_out,
                                             uint8_t* __restrict__ done_out) {
    if (a < 0) {
        reward_out[e] = 0.f;
        done_out[e] = 0;
        return false;
    }
    float reward = 0.f;
    bool done = false;
    if (p.game == 0) {
        // classic-control cart-pole, Euler step (games/cartpole.py CartPolePhysics on the host)
        const double gravity = 9.8, mass_cart = 1.0, mass_pole = 0.1, half_length = 0.5, force_mag = 10.0, tau = 0.02;
        double* s = p.state + 4 * e;
        const double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
        const double force = (a == 1) ? force_mag : -force_mag;
        const double total_mass = mass_cart + mass_pole;
        const double pole_ml = mass_pole * half_length;
        const double cos_t = cos(theta), sin_t = sin(theta);
        const double temp = (force + pole_ml * (theta_dot * theta_dot) * sin_t) / total_mass;
        const double theta_acc =
            (gravity * sin_t - cos_t * temp) / (half_length * (4.0 / 3.0 - mass_pole * (cos_t * cos_t) / total_mass));
        const double x_acc = temp - pole_ml * theta_acc * cos_t / total_mass;
        s[0] = x + tau * x_dot;
        s[1] = x_dot + tau * x_acc;
        s[2] = theta + tau * theta_dot;
        s[3] = theta_dot + tau * theta_acc;
        const int steps = ++p.steps[e];
        const double theta_limit = 12 * 2 * 3.141592653589793 / 360;
        done = fabs(s[0]) > 2.4 || fabs(s[2]) > theta_limit || steps >= 500;
        reward = 1.0f;
    } else {
        int8_t* b = p.board + static_cast<size_t>(e) * p.cells;
        const int pl = p.player[e];
        bool won, full = true;
        if (p.game == 1) {
            b[a] = static_cast<int8_t>(pl);
            won = ttt_winner(b, pl);
            for (int i = 0; i < 9; ++i) full = full && b[i] != 0;
            reward = won ? 20.f : 0.f;  // Game.step: reward * 20
        } else {
            for (int r = 0; r < 6; ++r)
                if (b[r * 7 + a] == 0) {
                    b[r * 7 + a] = static_cast<int8_t>(pl);
                    break;
                }
            won = c4_winner(b, pl);
            for (int c = 0; c < 7; ++c) full = full && b[35 + c] != 0;
            reward = won ? 10.f : 0.f;  // Game.step: reward * 10
        }
        done = won || full;
        p.player[e] = static_cast<int8_t>(-pl);
    }
    reward_out[e] = reward;
    done_out[e] = done ? 1 : 0;
    return done;
}

__device__ __forceinline__ void env_observe_one(const EnvParams& p, int e, float* __restrict__ obs,
                                                int32_t* __restrict__ legal, int32_t* __restrict__ num_legal,
                                                int32_t* __restrict__ to_play) {
    float* o = obs + static_cast<size_t>(e) * p.obs_floats;
    int32_t* l = legal + static_cast<size_t>(e) * p.A;
    if (p.game == 0) {
        for (int i = 0; i < 4; ++i) o[i] = static_cast<float>(p.state[4 * e + i]);
        l[0] = 0;
        l[1] = 1;
        num_legal[e] = 2;
        to_play[e] = 0;
        return;
    }
    const int8_t* b = p.board + static_cast<size_t>(e) * p.cells;
    const int pl = p.player[e];
    for (int i = 0; i < p.cells; ++i) {
        o[i] = b[i] == 1 ? 1.f : 0.f;
        o[p.cells + i] = b[i] == -1 ? 1.f : 0.f;
        o[2 * p.cells + i] = static_cast<float>(pl);
    }
    int n = 0;
    if (p.game == 1) {
        for (int i = 0; i < 9; ++i)
            if (b[i] == 0) l[n++] = i;
    } else {
        for (int c = 0; c < 7; ++c)
            if (b[35 + c] == 0) l[n++] = c;
    }
    num_legal[e] = n;
    to_play[e] = pl == 1 ? 0 : 1;
}

__global__ __launch_bounds__(256) void env_reset_kernel(EnvParams p, const uint8_t* __restrict__ mask) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.E || (mask && !mask[e])) return;
    env_reset_one(p, e);
}

__global__ __launch_bounds__(256) void env_step_kernel(EnvParams p, const int32_t* __restrict__ actions,
                                                       float* __restrict__ reward_out, uint8_t* __restrict__ done_out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.E) return;
    env_step_one(p, e, actions[e], reward_out, done_out);
}

__global__ __launch_bounds__(256) void env_observe_kernel(EnvParams p, float* __restrict__ obs, int32_t* __restrict__ legal,
                                                          int32_t* __restrict__ num_legal, int32_t* __restrict__ to_play) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.E) return;
    env_observe_one(p, e, obs, legal, num_legal, to_play);
}

// step, observation of the position reached, reset of the envs whose game ended, observation to search next:
// the four steps of one self-play move in one launch (envs are independent: one thread runs all four for its env).
__global__ __launch_bounds__(256) void env_advance_kernel(EnvParams p, const int32_t* __restrict__ actions,
                                                          float* __restrict__ reward_out, uint8_t* __restrict__ done_out,
                                                          float* __restrict__ obs_after, float* __restrict__ obs_next,
                                                          int32_t* __restrict__ legal, int32_t* __restrict__ num_legal,
                                                          int32_t* __restrict__ to_play) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.E) return;
    const bool done = env_step_one(p, e, actions[e], reward_out, done_out);
    env_observe_one(p, e, obs_after, legal, num_legal, to_play);
    if (done) env_reset_one(p, e);
    env_observe_one(p, e, obs_next, legal, num_legal, to_play);
}

hipError_t launch_env_reset(const EnvParams& p, const uint8_t* mask, hipStream_t stream) {
    env_reset_kernel<<<dim3((p.E + 255) / 256), dim3(256), 0, stream>>>(p, mask);
    return hipGetLastError();
}
hipError_t launch_env_step(const EnvParams& p, const int32_t* actions, float* reward, uint8_t* done, hipStream_t stream) {
    env_step_kernel<<<dim3((p.E + 255) / 256), dim3(256), 0, stream>>>(p, actions, reward, done);
    return hipGetLastError();
}
hipError_t launch_env_advance(const EnvParams& p, const int32_t* actions, float* reward, uint8_t* done, float* obs_after,
                              float* obs_next, int32_t* legal, int32_t* num_legal, int32_t* to_play, hipStream_t stream) {
    env_advance_kernel<<<dim3((p.E + 255) / 256), dim3(256), 0, stream>>>(p, actions, reward, done, obs_after, obs_next, legal,
                                                                          num_legal, to_play);
    return hipGetLastError();
}
hipError_t launch_env_observe(const EnvParams& p, float* obs, int32_t* legal, int32_t* num_legal, int32_t* to_play,
                              hipStream_t stream) {
    env_observe_kernel<<<dim3((p.E + 255) / 256), dim3(256), 0, stream>>>(p, obs, legal, num_legal, to_play);
    return hipGetLastError();
}

}  // namespace mz
