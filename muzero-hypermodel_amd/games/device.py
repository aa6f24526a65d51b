"""Device-resident batched environments (include/mzenv.h): E games of one kind advance on the GPU with
one kernel per move instead of E host `Game.step` calls.  Same plugin semantics as the host `Game`
classes of this package (and therefore as the reference's games/*.py): observations, reward scaling,
legal-action order, `to_play`.  CartPole's physics is the unpinned restatement of games/cartpole.py.
"""
import ctypes

import numpy as np
import torch

from .. import _native

GAME_IDS = {"cartpole": 0, "tictactoe": 1, "connect4": 2}


class DeviceEnvs:
    def __init__(self, game, num_envs, seeds=None, device=None):
        self._lib = _native.load()
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceEnvs needs a HIP device; use the host Game plugins on CPU")
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.game, self.E = game, int(num_envs)
        seeds = np.ascontiguousarray(np.arange(self.E) if seeds is None else seeds, dtype=np.int64) & 0xFFFFFFFF
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        handle = ctypes.c_void_p()
        rc = self._lib.mzenv_create(GAME_IDS[game], self.E, self.device.index, _native.ptr(seeds, _native.c_u32_p),
                                    ctypes.byref(handle))
        if rc != 0:
            raise RuntimeError(self._lib.mzenv_last_error(None).decode())
        self._h = handle
        a, p = ctypes.c_int32(), ctypes.c_int32()
        shape = (ctypes.c_int32 * 3)()
        self._lib.mzenv_shape(self._h, ctypes.byref(a), ctypes.byref(p), shape)
        self.A, self.players, self.observation_shape = a.value, p.value, tuple(shape)
        self.constant_legal_actions = game == "cartpole"     # every action legal in every state
        self.max_episode_steps = {"cartpole": 500, "tictactoe": 9, "connect4": 42}[game]
        with torch.cuda.device(self.device):
            self.obs = torch.zeros((self.E, *self.observation_shape), dtype=torch.float32, device=self.device)
            self.legal = torch.zeros((self.E, self.A), dtype=torch.int32, device=self.device)
            self.num_legal = torch.zeros(self.E, dtype=torch.int32, device=self.device)
            self.to_play = torch.zeros(self.E, dtype=torch.int32, device=self.device)
            self.reward = torch.zeros(self.E, dtype=torch.float32, device=self.device)
            self.done = torch.zeros(self.E, dtype=torch.uint8, device=self.device)
            self._actions = torch.zeros(self.E, dtype=torch.int32, device=self.device)
        self.reset()

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self._lib.mzenv_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mzenv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, mask=None):
        """Game.reset() for the envs whose mask entry is non-zero (uint8 device tensor; None = all)."""
        self._keep = mask
        self._check(self._lib.mzenv_reset(self._h, None if mask is None else mask.data_ptr(), self._stream()))

    def step(self, actions, reward=None, done=None):
        """Game.step per env; `actions`: int array / tensor [E] (negative = leave that env alone).
        Returns (reward, done) device tensors: the caller's `reward` (f32 [E]) / `done` (u8 [E]) when given,
        otherwise this object's own buffers (valid until the next step)."""
        if torch.is_tensor(actions) and actions.is_cuda and actions.dtype == torch.int32 and actions.is_contiguous():
            src = actions                                    # e.g. the search's own action buffer: no copy
        elif torch.is_tensor(actions):
            src = self._actions.copy_(actions.to(torch.int32), non_blocking=True)
        else:
            src = self._actions.copy_(torch.from_numpy(np.ascontiguousarray(actions, dtype=np.int32)), non_blocking=True)
        reward = self.reward if reward is None else reward
        done = self.done if done is None else done
        self._keep_step = (src, reward, done)
        self._check(self._lib.mzenv_step(self._h, src.data_ptr(), reward.data_ptr(), done.data_ptr(), self._stream()))
        return reward, done

    def advance(self, actions, reward, done, obs_after, obs_next):
        """One self-play move of every env in one call: step, observation after the move (terminal ones
        included), reset of the finished envs, observation the next search sees.  All arguments are resident
        tensors (actions int32 [E]; outputs as in step / observe)."""
        self._keep_step = (actions, reward, done, obs_after, obs_next)
        self._check(self._lib.mzenv_advance(self._h, actions.data_ptr(), reward.data_ptr(), done.data_ptr(),
                                            obs_after.data_ptr(), obs_next.data_ptr(), self.legal.data_ptr(),
                                            self.num_legal.data_ptr(), self.to_play.data_ptr(), self._stream()))
        return obs_next

    def observe(self, obs=None):
        """(observations [E,C,H,W] f32, legal [E,A] i32, num_legal [E] i32, to_play [E] i32) device tensors;
        the observations go to the caller's `obs` buffer when given."""
        obs = self.obs if obs is None else obs
        self._check(self._lib.mzenv_observe(self._h, obs.data_ptr(), self.legal.data_ptr(),
                                            self.num_legal.data_ptr(), self.to_play.data_ptr(), self._stream()))
        return obs, self.legal, self.num_legal, self.to_play
