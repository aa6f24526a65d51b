// env_layout.h -- device state of the batched environments (env_kernels.hip, mzenv_capi.hip).
#pragma once
#include <cstdint>

namespace mz {

struct EnvParams {
    int32_t game, E, A, cells, obs_floats;
    int8_t* board;     // [E][cells]  tictactoe / connect4: 0 empty, +1 first player, -1 second player
    int8_t* player;    // [E]         +1 / -1: the player to move
    double* state;     // [E][4]      cartpole: x, x_dot, theta, theta_dot
    int32_t* steps;    // [E]         cartpole
    uint32_t* mt_key;  // [E][624]    cartpole reset stream (numpy RandomState(seed))
    int32_t* mt_pos;   // [E]
};

}  // namespace mz
