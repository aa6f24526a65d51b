"""ReplayBuffer with the reference's interface (replay_buffer.py:11-295) over the device-resident replay
store (include/mzreplay.h): finished games live on the GPU as packed arrays, initial priorities and the
training targets of a batch are computed there, and `get_batch` returns device tensors.

What stays on the host, as in the reference: which games / positions are sampled and the random actions of
absorbing states, drawn in the reference's order from a numpy-compatible legacy stream seeded with
config.seed (the reference seeds numpy's global stream in the buffer's own process, replay_buffer.py:31).
Sampling reads the priorities, which are mirrored on the host (a few floats per game).

Not carried over: Ray (`.remote`), `get_buffer()`'s live GameHistory objects are only kept when games
arrive as GameHistory (save_game), and `update_game_history` (Reanalyse).
"""
import ctypes

import numpy
import torch

from . import _native
from ._native import c_f32_p, c_f64_p, c_i32_p, ptr


class MzReplayConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("capacity", "max_moves", "num_actions", "obs_channels", "obs_height",
                                              "obs_width", "stacked_observations", "td_steps", "num_unroll_steps",
                                              "device")] + \
               [("per_alpha", ctypes.c_double), ("discount_powers", ctypes.c_void_p)]


class ReplayBuffer:
    def __init__(self, initial_checkpoint, initial_buffer, config, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("the device replay store needs a HIP device (there is no CPU fallback)")
        self.config = config
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._lib = _native.load()
        self.num_played_games = initial_checkpoint["num_played_games"]
        self.num_played_steps = initial_checkpoint["num_played_steps"]
        self.total_samples = 0
        self.A = len(config.action_space)
        self.C, self.H, self.W = (int(v) for v in config.observation_shape)
        self.L = int(config.max_moves)
        self.capacity = int(config.replay_buffer_size)
        self.U1 = int(config.num_unroll_steps) + 1
        # discount ** i as Python computes it (replay_buffer.py:240, 253)
        powers = numpy.array([config.discount ** i for i in range(config.td_steps + 1)], dtype=numpy.float64)
        cfg = MzReplayConfig(self.capacity, self.L, self.A, self.C, self.H, self.W, int(config.stacked_observations),
                             int(config.td_steps), int(config.num_unroll_steps), self.device.index,
                             float(config.PER_alpha), powers.ctypes.data)
        handle = ctypes.c_void_p()
        if self._lib.mzreplay_create(ctypes.byref(cfg), ctypes.byref(handle)) != 0:
            raise RuntimeError(self._lib.mzreplay_last_error(None).decode())
        self._h = handle
        self.rng = _native.HostRng(config.seed)           # numpy.random.seed(self.config.seed)
        self.buffer = {}                                  # game_id -> dict(length, priorities, game_priority[, history])
        for game_history in (initial_buffer or {}).values():
            self.save_game(game_history)

    # ---- plumbing -----------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self._lib.mzreplay_last_error(self._h).decode())

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mzreplay_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_bytes(self):
        return int(self._lib.mzreplay_device_bytes(self._h))

    def _slot(self, game_id):
        return game_id % self.capacity

    # ---- save_game (replay_buffer.py:33-65) ----------------------------------------------------------
    def save_game(self, game_history, shared_storage=None):
        n = len(game_history.root_values)
        obs = numpy.zeros((1, self.L + 1, self.C, self.H, self.W), dtype=numpy.float32)
        obs[0, : n + 1] = numpy.asarray(game_history.observation_history, dtype=numpy.float32).reshape(n + 1, self.C, self.H, self.W)
        act = numpy.zeros((1, self.L + 1), dtype=numpy.int32)
        act[0, : n + 1] = game_history.action_history
        rew = numpy.zeros((1, self.L + 1), dtype=numpy.float64)
        rew[0, : n + 1] = game_history.reward_history
        tp = numpy.zeros((1, self.L + 1), dtype=numpy.int32)
        tp[0, : n + 1] = game_history.to_play_history
        cv = numpy.zeros((1, self.L, self.A), dtype=numpy.float64)
        cv[0, :n] = game_history.child_visits
        rv = numpy.zeros((1, self.L), dtype=numpy.float64)
        rv[0, :n] = game_history.root_values
        self._add(numpy.array([n], dtype=numpy.int32), obs, act, rew, tp, cv, rv, [game_history], shared_storage)

    def save_games(self, packed, shared_storage=None):
        """A batch of finished games as arrays (self_play.PackedGames): no per-game Python objects."""
        n_games, width = len(packed), packed.actions.shape[1] - 1

        def padded(a, tail, dtype):
            out = numpy.zeros((n_games, tail) + a.shape[2:], dtype=dtype)
            out[:, : a.shape[1]] = a
            return out
        self._add(numpy.ascontiguousarray(packed.length, dtype=numpy.int32),
                  padded(packed.observations.reshape(n_games, width + 1, self.C, self.H, self.W), self.L + 1, numpy.float32),
                  padded(packed.actions, self.L + 1, numpy.int32), padded(packed.rewards, self.L + 1, numpy.float64),
                  padded(packed.to_play, self.L + 1, numpy.int32), padded(packed.child_visits, self.L, numpy.float64),
                  padded(packed.root_values, self.L, numpy.float64), [None] * n_games, shared_storage)

    def _add(self, lengths, obs, act, rew, tp, cv, rv, histories, shared_storage):
        n_games = len(lengths)
        ids = numpy.arange(self.num_played_games, self.num_played_games + n_games)
        slots = numpy.ascontiguousarray(ids % self.capacity, dtype=numpy.int32)
        pri = numpy.zeros((n_games, self.L), dtype=numpy.float32)
        game_pri = numpy.zeros(n_games, dtype=numpy.float32)
        with torch.cuda.device(self.device):
            self._check(self._lib.mzreplay_add_games(
                self._h, n_games, ptr(slots, c_i32_p), ptr(lengths, c_i32_p), ptr(obs, c_f32_p), ptr(act, c_i32_p),
                ptr(rew, c_f64_p), ptr(tp, c_i32_p), ptr(cv, c_f64_p), ptr(rv, c_f64_p), ptr(pri, c_f32_p),
                ptr(game_pri, c_f32_p), self._stream()))
        for g in range(n_games):
            n = int(lengths[g])
            entry = dict(length=n, history=histories[g])
            if self.config.PER:
                given = getattr(histories[g], "priorities", None) if histories[g] is not None else None
                entry["priorities"] = numpy.copy(given) if given is not None else pri[g, :n].copy()
                entry["game_priority"] = numpy.max(entry["priorities"]) if given is not None else game_pri[g]
                if histories[g] is not None:
                    histories[g].priorities, histories[g].game_priority = entry["priorities"], entry["game_priority"]
            self.buffer[self.num_played_games] = entry
            self.num_played_games += 1
            self.num_played_steps += n
            self.total_samples += n
            if self.config.replay_buffer_size < len(self.buffer):
                del_id = self.num_played_games - len(self.buffer)
                self.total_samples -= self.buffer[del_id]["length"]
                del self.buffer[del_id]
        if shared_storage:
            shared_storage.set_info("num_played_games", self.num_played_games)
            shared_storage.set_info("num_played_steps", self.num_played_steps)

    def get_buffer(self):
        return {gid: e["history"] for gid, e in self.buffer.items()}

    # ---- sampling (replay_buffer.py:135-195) -----------------------------------------------------------
    def sample_game(self, force_uniform=False):
        game_prob = None
        if self.config.PER and not force_uniform:
            game_probs = numpy.array([e["game_priority"] for e in self.buffer.values()], dtype="float32")
            game_probs /= numpy.sum(game_probs)
            game_index = self.rng.choice_p(game_probs)
            game_prob = game_probs[game_index]
        else:
            game_index = self.rng.choice(len(self.buffer))
        game_id = self.num_played_games - len(self.buffer) + game_index
        return game_id, self.buffer[game_id], game_prob

    def sample_n_games(self, n_games, force_uniform=False):
        ids = list(self.buffer.keys())
        if self.config.PER and not force_uniform:
            game_probs = numpy.array([self.buffer[g]["game_priority"] for g in ids], dtype="float32")
            game_probs /= numpy.sum(game_probs)
            prob_of = dict(zip(ids, game_probs))
            selected = [ids[i] for i in self.rng.choice_p_many(game_probs, n_games)]
        else:
            prob_of = {}
            selected = [ids[self.rng.choice(len(ids))] for _ in range(n_games)]
        return [(g, self.buffer[g], prob_of.get(g)) for g in selected]

    def sample_position(self, entry, force_uniform=False):
        position_prob = None
        if self.config.PER and not force_uniform:
            position_index, position_prob = self.rng.choice_priorities(entry["priorities"])
        else:
            position_index = self.rng.choice(entry["length"])
        return position_index, position_prob

    # ---- get_batch (replay_buffer.py:67-133) -----------------------------------------------------------
    def get_batch(self):
        B = self.config.batch_size
        index_batch, weight_batch = [], [] if self.config.PER else None
        slots = numpy.zeros(B, dtype=numpy.int32)
        positions = numpy.zeros(B, dtype=numpy.int32)
        absorbing = numpy.zeros((B, self.U1), dtype=numpy.int32)
        for b, (game_id, entry, game_prob) in enumerate(self.sample_n_games(B)):
            game_pos, pos_prob = self.sample_position(entry)
            # make_target draws numpy.random.choice(action_space) for every state past the end of the game
            for u in range(self.U1):
                if game_pos + u > entry["length"]:
                    absorbing[b, u] = self.config.action_space[self.rng.choice(self.A)]
            index_batch.append([game_id, game_pos])
            slots[b], positions[b] = self._slot(game_id), game_pos
            if self.config.PER:
                weight_batch.append(1 / (self.total_samples * game_prob * pos_prob))
        if self.config.PER:
            weight_batch = numpy.array(weight_batch, dtype="float32") / max(weight_batch)
        out = self.make_targets(slots, positions, absorbing)
        return index_batch, (out["observation"], out["action"], out["value"], out["reward"], out["policy"],
                             weight_batch, out["gradient_scale"])

    def make_targets(self, slots, positions, absorbing):
        """Device tensors of a batch of (slot, position) pairs: observation [B,C',H,W] f32, action [B,U+1] i64,
        value / reward / gradient_scale [B,U+1] f64, policy [B,U+1,A] f64."""
        B = len(slots)
        stacked = int(self.config.stacked_observations)
        dev = self.device
        out = dict(observation=torch.empty((B, self.C + stacked * (self.C + 1), self.H, self.W), dtype=torch.float32, device=dev),
                   action=torch.empty((B, self.U1), dtype=torch.int64, device=dev),
                   value=torch.empty((B, self.U1), dtype=torch.float64, device=dev),
                   reward=torch.empty((B, self.U1), dtype=torch.float64, device=dev),
                   policy=torch.empty((B, self.U1, self.A), dtype=torch.float64, device=dev),
                   gradient_scale=torch.empty((B, self.U1), dtype=torch.float64, device=dev))
        slots = numpy.ascontiguousarray(slots, dtype=numpy.int32)
        positions = numpy.ascontiguousarray(positions, dtype=numpy.int32)
        absorbing = numpy.ascontiguousarray(absorbing, dtype=numpy.int32)
        with torch.cuda.device(dev):
            self._check(self._lib.mzreplay_make_batch(
                self._h, B, ptr(slots, c_i32_p), ptr(positions, c_i32_p), ptr(absorbing, c_i32_p),
                out["observation"].data_ptr(), out["action"].data_ptr(), out["value"].data_ptr(), out["reward"].data_ptr(),
                out["policy"].data_ptr(), out["gradient_scale"].data_ptr(), self._stream()))
        self._keep = (slots, positions, absorbing)
        return out

    # ---- Reanalyse's two ends (replay_buffer.py:335-356) ------------------------------------------------
    def game_observations(self, game_id):
        """Stacked observations of every position of a stored game: CUDA tensor [n, C', H, W]."""
        n = self.buffer[game_id]["length"]
        stacked = int(self.config.stacked_observations)
        out = torch.empty((n, self.C + stacked * (self.C + 1), self.H, self.W), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self._lib.mzreplay_game_observations(self._h, self._slot(game_id), n, out.data_ptr(), self._stream()))
        return out

    def set_reanalysed_values(self, game_id, values):
        """game_history.reanalysed_predicted_root_values = values (float32 [n], tensor or array)."""
        if game_id not in self.buffer:      # the game could have been removed since its selection
            return
        n = self.buffer[game_id]["length"]
        v = torch.as_tensor(values, dtype=torch.float32).reshape(-1).contiguous()
        assert v.numel() == n
        self._keep_values = v
        with torch.cuda.device(self.device):
            self._check(self._lib.mzreplay_set_reanalysed(self._h, self._slot(game_id), v.data_ptr(), n, self._stream()))
        self.buffer[game_id]["reanalysed"] = True

    # ---- priorities (replay_buffer.py:197-220) ---------------------------------------------------------
    def update_priorities(self, priorities, index_info):
        for i in range(len(index_info)):
            game_id, game_pos = index_info[i]
            if next(iter(self.buffer)) <= game_id:
                entry = self.buffer[game_id]
                priority = priorities[i, :]
                start_index = game_pos
                end_index = min(game_pos + len(priority), len(entry["priorities"]))
                entry["priorities"][start_index:end_index] = priority[: end_index - start_index]
                entry["game_priority"] = numpy.max(entry["priorities"])


class Reanalyse:
    """Reanalyse (reference replay_buffer.py:297-361) against the device store: one batched initial_inference
    over all positions of a sampled game, support_to_scalar, values written back into the store."""

    def __init__(self, initial_checkpoint, config, device=None):
        from . import models
        self._models = models
        self.config = config
        torch.manual_seed(self.config.seed)
        self.device = torch.device(device if device is not None else "cuda")
        self.model = models.MuZeroNetwork(self.config)
        self.model.set_weights(initial_checkpoint["weights"])
        self.model.to(self.device)
        self.model.eval()
        self.num_reanalysed_games = initial_checkpoint.get("num_reanalysed_games", 0)

    @torch.no_grad()
    def reanalyse_game(self, replay_buffer, game_id=None):
        """One pass of the reference's loop body; returns (game_id, values tensor)."""
        if game_id is None:
            game_id, _, _ = replay_buffer.sample_game(force_uniform=True)
        values = None
        if self.config.use_last_model_value:
            observations = replay_buffer.game_observations(game_id)
            values = self._models.support_to_scalar(self.model.initial_inference(observations)[0], self.config.support_size)
            values = values.reshape(-1).float()
            replay_buffer.set_reanalysed_values(game_id, values)
        self.num_reanalysed_games += 1
        return game_id, values

    def reanalyse(self, replay_buffer, shared_storage, max_games=None):
        """The reference's loop without Ray: runs until shared_storage reports the end of training."""
        done = 0
        while shared_storage.get_info("training_step") < self.config.training_steps and not shared_storage.get_info("terminate"):
            self.model.set_weights(shared_storage.get_info("weights"))
            self.reanalyse_game(replay_buffer)
            shared_storage.set_info("num_reanalysed_games", self.num_reanalysed_games)
            done += 1
            if max_games is not None and done >= max_games:
                break
