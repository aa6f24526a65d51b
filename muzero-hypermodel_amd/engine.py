"""BatchedMCTS: the Python host that drives libmzmcts.so for E trees in lock step.

One search (= the reference's `MCTS.run`, self_play.py:261-362, for every env at once):

    roots      initial_inference on the [E,...] observation batch (PyTorch-ROCm)
               begin_search  (host: contract checks, Dirichlet noise on the per-env RNG mirrors)
               expand_roots  (HIP: decode, masked fp32 softmax, noise mix, per-search reset)
    S times    select        (HIP: UCB descent for all trees + hidden-state gather)
               recurrent_inference on the gathered [E,H] batch (PyTorch-ROCm, MFMA GEMMs/convs),
                             next state written straight into the pool slab
               expand_backup (HIP: decode, expand, LDS-staged backup, min-max)
    readout    root statistics to the host, action sampling on the RNG mirrors

After the first (eager, warm-up) search the S-simulation loop can be captured once into a hipGraph
(`torch.cuda.CUDAGraph`; the C-ABI launch functions neither allocate nor synchronise) and replayed
per move, which removes the per-launch host cost of ~30 small kernels per simulation.

PyTorch is plumbing here (device memory, streams, the network modules); the tree work is in the
HIP library and there is no fallback if it is missing.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _native
from ._native import MzConfig, MzFcDesc, MzProfile, MzRootStats, c_f32_p, c_f64_p, c_i32_p, c_i64_p, c_u32_p, ptr


def hidden_state_shape(config):
    """Shape of one hidden state (without batch) for a MuZeroConfig (models.py:80-127, 432-516)."""
    if config.network == "fullyconnected":
        return (config.encoding_size,)
    _, h, w = config.observation_shape
    if config.downsample:
        h, w = math.ceil(h / 16), math.ceil(w / 16)
    return (config.channels, h, w)


class BatchedMCTS:
    def __init__(self, config, num_envs, device=None, seeds=None, use_graph=False, group_width=0):
        if len(config.players) > 2:
            raise NotImplementedError("More than two player mode not implemented.")
        if list(config.players) != list(range(len(config.players))):
            raise NotImplementedError("players must be list(range(n)) (the reference's only supported form)")
        if list(config.action_space) != list(range(len(config.action_space))):
            raise NotImplementedError("action_space must be list(range(n)) (the reference's only supported form)")
        self._lib = _native.load()
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedMCTS needs a HIP device (MI355X); there is no CPU fallback")
        self.config = config
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.E = int(num_envs)
        self.A = len(config.action_space)
        self.S = int(config.num_simulations)
        self.F = 2 * config.support_size + 1
        self.state_shape = hidden_state_shape(config)
        self.H = int(np.prod(self.state_shape))
        self.use_graph = bool(use_graph)
        self._graph = None
        self._eager_searches = 0
        self._graph_model = None
        self._graph_generation = 0
        self._profiling = False
        self._group_width = int(group_width)
        self.stream = None          # optional dedicated torch.cuda.Stream (PipelinedSearch)
        self._fc_model = None
        self._ring_moves = ctypes.c_int32()
        self._batch_keep, self._batch_moves = [], 0
        self._fc_flat = None
        self.fused_hidden_in_lds = True
        # lock-step loop: expand_backup + next select in one launch (mzmcts_expand_backup_select).  Bit-identical and one
        # launch fewer per simulation; measured faster at many envs (TicTacToe, 2 x 32768 envs: +2-4 %; config #5 at 32768
        # envs: unchanged) and slower at few (Connect4, 2 x 4096 envs: -2 %): MZ_FUSED_STEP=on / off, default by env count
        mode = os.environ.get("MZ_FUSED_STEP", "auto")
        self.fused_step = mode == "on" or (mode != "off" and self.E >= 16384)
        self._device_noise = False

        with torch.cuda.device(self.device):
            self.pool = torch.empty((self.S + 1, self.E, self.H), dtype=torch.float32, device=self.device)
            self.batch_hidden = torch.zeros((self.E, self.H), dtype=torch.float32, device=self.device)
            self.batch_action = torch.zeros((self.E, 1), dtype=torch.int64, device=self.device)
        cfg = MzConfig(num_envs=self.E, num_actions=self.A, num_simulations=self.S,
                       num_players=len(config.players), support_size=int(config.support_size),
                       hidden_floats=self.H, device=self.device.index, group_width=int(group_width),
                       discount=float(config.discount), pb_c_base=float(config.pb_c_base),
                       pb_c_init=float(config.pb_c_init),
                       root_dirichlet_alpha=float(config.root_dirichlet_alpha),
                       root_exploration_fraction=float(config.root_exploration_fraction),
                       hidden_pool=self.pool.data_ptr())
        handle = ctypes.c_void_p()
        rc = self._lib.mzmcts_create(ctypes.byref(cfg), ctypes.byref(handle))
        _native.check(self._lib, None, rc)
        self._h = handle
        if seeds is None:
            seeds = [int(config.seed) + e for e in range(self.E)]  # worker e: config.seed + e (muzero.py:175)
        self.seed(seeds)

        # host result buffers
        E, A = self.E, self.A
        self._legal = np.zeros((E, A), dtype=np.int32)
        self._nlegal = np.zeros(E, dtype=np.int32)
        self._to_play = np.zeros(E, dtype=np.int32)
        self.noise = np.zeros((E, A), dtype=np.float64)
        self.stats = dict(
            visits=np.zeros((E, A), np.int32), child_value_sum=np.zeros((E, A)),
            child_prior=np.zeros((E, A)), child_reward=np.zeros((E, A)),
            child_expanded=np.zeros((E, A), np.int32), root_value_sum=np.zeros(E),
            root_visits=np.zeros(E, np.int32), max_tree_depth=np.zeros(E, np.int32),
            root_predicted_value=np.zeros(E), min_max=np.zeros((E, 2)),
            depth_sum=np.zeros(E, np.int64), tie_break_words=np.zeros(E, np.uint32))
        # ctypes pointers of the persistent host buffers (ndarray.ctypes is slow: build them once)
        self._p_legal, self._p_nlegal = ptr(self._legal, c_i32_p), ptr(self._nlegal, c_i32_p)
        self._p_to_play, self._p_noise = ptr(self._to_play, c_i32_p), ptr(self.noise, c_f64_p)
        self._actions = np.zeros(E, np.int32)
        self._slots = np.zeros(E, np.int32)
        self._p_actions, self._p_slots = ptr(self._actions, c_i32_p), ptr(self._slots, c_i32_p)
        self._temp_cache = (None, None)
        s = self.stats
        self._stats_struct = MzRootStats(
            ptr(s["visits"], c_i32_p), ptr(s["child_value_sum"], c_f64_p), ptr(s["child_prior"], c_f64_p),
            ptr(s["child_reward"], c_f64_p), ptr(s["child_expanded"], c_i32_p),
            ptr(s["root_value_sum"], c_f64_p), ptr(s["root_visits"], c_i32_p),
            ptr(s["max_tree_depth"], c_i32_p), ptr(s["root_predicted_value"], c_f64_p),
            ptr(s["min_max"], c_f64_p), ptr(s["depth_sum"], c_i64_p), ptr(s["tie_break_words"], c_u32_p))

    # ------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._graph = None
            self._lib.mzmcts_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        _native.check(self._lib, self._h, rc)

    def _stream(self):
        # (while a hipGraph is being captured the launches must go to the capturing stream, which torch has made the
        # current one -- on the engine's own stream they would run right away and be missing from every replay)
        if self.stream is not None and not torch.cuda.is_current_stream_capturing():
            stream = self.stream
        else:
            stream = torch.cuda.current_stream(self.device)
        return ctypes.c_void_p(stream.cuda_stream)

    # ---- RNG ------------------------------------------------------------------------------------
    def seed(self, seeds):
        seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.int64) & 0xFFFFFFFF, dtype=np.uint32)
        assert seeds.shape == (self.E,)
        self._check(self._lib.mzmcts_seed(self._h, ptr(seeds, c_u32_p), self._stream()))

    def set_rng_state(self, env, state):
        """state: the tuple numpy.random.get_state() returns."""
        key = np.ascontiguousarray(state[1], dtype=np.uint32)
        self._check(self._lib.mzmcts_rng_set_state(self._h, env, ptr(key, c_u32_p), int(state[2]),
                                                   int(state[3]), float(state[4]), self._stream()))

    def get_rng_state(self, env):
        key = np.zeros(624, dtype=np.uint32)
        pos, hg, g = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double()
        self._check(self._lib.mzmcts_rng_get_state(self._h, env, ptr(key, c_u32_p), ctypes.byref(pos),
                                                   ctypes.byref(hg), ctypes.byref(g), self._stream()))
        return ("MT19937", key, pos.value, hg.value, g.value)

    # ---- low-level steps (also what the parity tests drive) -------------------------------------
    def begin_search(self, legal_actions, to_play, add_exploration_noise=True, num_legal=None):
        """legal_actions: per env a sequence of actions (empty / None = env inactive this search), or --
        the loop-free form for large E -- an int32 array [E, A] whose row e holds num_legal[e] actions."""
        if num_legal is not None:
            self._legal[:] = legal_actions
            self._nlegal[:] = num_legal
        else:
            self._legal[:] = 0
            for e, legal in enumerate(legal_actions):
                n = 0 if legal is None else len(legal)
                if n > self.A:
                    raise AssertionError("Legal actions should be a subset of the action space.")
                self._nlegal[e] = n
                if n:
                    self._legal[e, :n] = legal
        self._to_play[:] = to_play
        self._check(self._lib.mzmcts_begin_search(
            self._h, self._p_legal, self._p_nlegal, self._p_to_play,
            1 if add_exploration_noise else 0, self._p_noise, self._stream()))

    def expand_roots(self, value_logits, reward_logits, policy_logits, root_hidden=None):
        v, p = self._f32(value_logits, self.F), self._f32(policy_logits, self.A)
        r = None if reward_logits is None else self._f32(reward_logits, self.F)
        h = None if root_hidden is None else self._f32(root_hidden.reshape(self.E, -1), self.H)
        self._keep = (v, r, p, h)
        self._check(self._lib.mzmcts_expand_roots(
            self._h, v.data_ptr(), None if r is None else r.data_ptr(), p.data_ptr(),
            None if h is None else h.data_ptr(), self._stream()))

    def expand_roots_injected(self, root_reward, root_priors):
        r = torch.as_tensor(np.asarray(root_reward, dtype=np.float64), device=self.device).contiguous()
        p = torch.as_tensor(np.asarray(root_priors, dtype=np.float64), device=self.device).contiguous()
        assert r.shape == (self.E,) and p.shape == (self.E, self.A)
        self._keep = (r, p)
        self._check(self._lib.mzmcts_expand_roots_injected(self._h, r.data_ptr(), p.data_ptr(), self._stream()))

    def select(self, gather=True):
        self._check(self._lib.mzmcts_select(
            self._h, self.batch_hidden.data_ptr() if gather and self.H else None,
            self.batch_action.data_ptr(), self._stream()))

    def select_planes(self):
        """select with the gather laid out as a residual network's dynamics input: [E, channels + 1, h, w], the last
        plane action / A (reference models.py:553-568)."""
        c, h, w = self.state_shape
        if getattr(self, "batch_planes", None) is None:
            self.batch_planes = torch.zeros((self.E, c + 1, h, w), dtype=torch.float32, device=self.device)
        self._check(self._lib.mzmcts_select_planes(self._h, self.batch_planes.data_ptr(), self.batch_action.data_ptr(),
                                                   h * w, self.A, self._stream()))
        return self.batch_planes

    def expand_backup(self, value_logits, reward_logits, policy_logits, next_hidden=None):
        v, r, p = self._f32(value_logits, self.F), self._f32(reward_logits, self.F), self._f32(policy_logits, self.A)
        h = None if next_hidden is None else self._f32(next_hidden.reshape(self.E, -1), self.H)
        self._keep = (v, r, p, h)
        self._check(self._lib.mzmcts_expand_backup(
            self._h, v.data_ptr(), r.data_ptr(), p.data_ptr(), None if h is None else h.data_ptr(),
            self._stream()))

    def expand_backup_injected(self, value, reward, priors):
        v = torch.as_tensor(np.asarray(value, dtype=np.float64), device=self.device).contiguous()
        r = torch.as_tensor(np.asarray(reward, dtype=np.float64), device=self.device).contiguous()
        p = torch.as_tensor(np.asarray(priors, dtype=np.float64), device=self.device).contiguous()
        assert v.shape == (self.E,) and r.shape == (self.E,) and p.shape == (self.E, self.A)
        self._keep = (v, r, p)
        self._check(self._lib.mzmcts_expand_backup_injected(self._h, v.data_ptr(), r.data_ptr(),
                                                            p.data_ptr(), self._stream()))

    # expand_backup of the simulation in flight + select of the next one in ONE launch (not for the last simulation)
    def expand_backup_select(self, value_logits, reward_logits, policy_logits, next_hidden=None, gather=True):
        v, r, p = self._f32(value_logits, self.F), self._f32(reward_logits, self.F), self._f32(policy_logits, self.A)
        h = None if next_hidden is None else self._f32(next_hidden.reshape(self.E, -1), self.H)
        self._keep = (v, r, p, h)
        self._check(self._lib.mzmcts_expand_backup_select(
            self._h, v.data_ptr(), r.data_ptr(), p.data_ptr(), None if h is None else h.data_ptr(),
            self.batch_hidden.data_ptr() if gather and self.H else None, self.batch_action.data_ptr(), self._stream()))

    def expand_backup_select_planes(self, value_logits, reward_logits, policy_logits):
        v, r, p = self._f32(value_logits, self.F), self._f32(reward_logits, self.F), self._f32(policy_logits, self.A)
        c, h, w = self.state_shape
        self._keep = (v, r, p)
        self._check(self._lib.mzmcts_expand_backup_select_planes(
            self._h, v.data_ptr(), r.data_ptr(), p.data_ptr(), None, self.batch_planes.data_ptr(),
            self.batch_action.data_ptr(), h * w, self.A, self._stream()))
        return self.batch_planes

    def expand_backup_select_injected(self, value, reward, priors, gather=False):
        v = torch.as_tensor(np.asarray(value, dtype=np.float64), device=self.device).contiguous()
        r = torch.as_tensor(np.asarray(reward, dtype=np.float64), device=self.device).contiguous()
        p = torch.as_tensor(np.asarray(priors, dtype=np.float64), device=self.device).contiguous()
        assert v.shape == (self.E,) and r.shape == (self.E,) and p.shape == (self.E, self.A)
        self._keep = (v, r, p)
        self._check(self._lib.mzmcts_expand_backup_select_injected(
            self._h, v.data_ptr(), r.data_ptr(), p.data_ptr(), self.batch_hidden.data_ptr() if gather and self.H else None,
            self.batch_action.data_ptr(), self._stream()))

    def _f32(self, t, width):
        assert t.is_cuda and t.dtype == torch.float32, "network outputs must be fp32 device tensors"
        t = t.contiguous()
        assert t.shape == (self.E, width), f"expected shape {(self.E, width)}, got {tuple(t.shape)}"
        return t

    def next_slab(self):
        """Pool slab (as an [E, *state_shape] view) the coming expand_backup will own."""
        k = self._lib.mzmcts_next_slab(self._h)
        return self.pool[k].view(self.E, *self.state_shape)

    def simulations_done(self):
        return self._lib.mzmcts_simulations_done(self._h)

    def last_paths(self, with_ties=False):
        depth = np.zeros(self.E, np.int32)
        actions = np.zeros((self.E, self.S), np.int32)
        ties = np.zeros((self.E, self.S), np.int32) if with_ties else None
        self._check(self._lib.mzmcts_last_paths(self._h, ptr(depth, c_i32_p), ptr(actions, c_i32_p),
                                                ptr(ties, c_i32_p), self._stream()))
        return depth, actions, ties

    def set_debug_ties(self, enabled=True):
        self._check(self._lib.mzmcts_set_debug_ties(self._h, 1 if enabled else 0))

    def export_tree(self, env):
        n = (self.S + 1) * self.A
        out = dict(visits=np.zeros(n, np.int32), value_sum=np.zeros(n), prior=np.zeros(n),
                   reward=np.zeros(n), child_node=np.zeros(n, np.int32))
        self._check(self._lib.mzmcts_export_tree(
            self._h, env, ptr(out["visits"], c_i32_p), ptr(out["value_sum"], c_f64_p),
            ptr(out["prior"], c_f64_p), ptr(out["reward"], c_f64_p), ptr(out["child_node"], c_i32_p),
            self._stream()))
        return {k: v.reshape(self.S + 1, self.A) for k, v in out.items()}

    # ---- the simulation loop ---------------------------------------------------------------------
    def _planes_path(self, model):
        return len(self.state_shape) == 3 and hasattr(model, "recurrent_inference_from_planes")

    def _pool_path(self, model):
        """Residual networks whose towers can read their input straight from the hidden-state pool (include/mzmcts.h
        mzmcts_board_tower_gathered): no [E, channels + 1, h, w] tensor between the descent and the network."""
        ok = getattr(self, "_pool_ok", None)
        if ok is None or ok[0] is not model:
            usable = (self._planes_path(model) and hasattr(model, "recurrent_inference_from_pool")
                      and model.pool_towers_supported(self.E, self.device, self.state_shape))
            self._pool_ok = ok = (model, usable)
        return ok[1]

    def tower_gather(self):
        """The descriptor of that input after a select() (valid for this engine's lifetime)."""
        g = getattr(self, "_gather_desc", None)
        if g is None:
            g = self._gather_desc = _native.MzTowerGather()
            self._check(self._lib.mzmcts_tower_gather_args(self._h, self.batch_action.data_ptr(), self.A, ctypes.byref(g)))
        return g

    def _select_for(self, model):
        """The descent of the simulation about to run (its gather feeds the network)."""
        if self._pool_path(model):
            self.select(gather=False)   # the towers gather for themselves
        elif self._planes_path(model):
            self.select_planes()     # residual networks: the gather writes the dynamics input (state planes + action plane)
        else:
            self.select()

    def _infer(self, model):
        slab = self.next_slab()
        if self._pool_path(model):
            return model.recurrent_inference_from_pool(self.tower_gather(), self.E, out_state=slab)
        if self._planes_path(model):
            return model.recurrent_inference_from_planes(self.batch_planes, out_state=slab)
        return model.recurrent_inference(self.batch_hidden.view(self.E, *self.state_shape), self.batch_action, out_state=slab)

    def _simulate_once(self, model):
        """One simulation as three steps (select, inference, expand_backup): what the parity tests drive."""
        self._select_for(model)
        value, reward, policy, _ = self._infer(model)
        self.expand_backup(value, reward, policy, None)

    def _simulate_all(self, model):
        """The S simulations of a search: select once, then per simulation the inference and ONE tree launch -- expand +
        backup of this simulation fused with the descent of the next (include/mzmcts.h mzmcts_expand_backup_select) --
        and a plain expand_backup for the last."""
        done = self.simulations_done()
        self._select_for(model)
        for s in range(done, self.S):
            value, reward, policy, _ = self._infer(model)
            if s + 1 < self.S and self.fused_step:
                if self._pool_path(model):
                    self.expand_backup_select(value, reward, policy, None, gather=False)   # the towers gather for themselves
                elif self._planes_path(model):
                    self.expand_backup_select_planes(value, reward, policy)
                else:
                    self.expand_backup_select(value, reward, policy, None)
            else:
                self.expand_backup(value, reward, policy, None)
                if s + 1 < self.S:
                    self._select_for(model)

    def _run_simulations(self, model):
        graph_ok = self.use_graph and not self._profiling
        # (a captured loop holds the addresses of the network's tensors: weights.FlatWeights moving them into its flat
        # buffer after the capture -- a first weight pull behind a first game -- makes the capture stale)
        generation = getattr(model, "_storage_generation", 0)
        if graph_ok and self._graph is not None and (self._graph_model is not model or self._graph_generation != generation):
            self._graph = None
        if graph_ok and self._graph is not None:
            self._check(self._lib.mzmcts_set_simulations_done(self._h, 0))
            self._graph.replay()
            self._check(self._lib.mzmcts_set_simulations_done(self._h, self.S))
            return
        if graph_ok and self._eager_searches >= 1:
            # capture the S-simulation loop once; capture records, the replay below executes
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(self.device)
            with torch.cuda.graph(graph):
                self._simulate_all(model)
            self._graph, self._graph_model, self._graph_generation = graph, model, generation
            self._check(self._lib.mzmcts_set_simulations_done(self._h, 0))
            graph.replay()
            self._check(self._lib.mzmcts_set_simulations_done(self._h, self.S))
            return
        self._simulate_all(model)
        self._eager_searches += 1

    # ---- fully-connected networks in-kernel ---------------------------------------------------------
    def configure_fused_fc(self, model, flat=None):
        """Hand a MuZeroFullyConnectedNetwork to the HIP library: afterwards `search(model, ...)` runs the
        whole move in one launch (trees, hidden states, activations and weights resident in LDS).
        `flat`: the weights.FlatWeights of `model` (created if omitted); its buffer is what the kernel
        reads, so an RCCL weight broadcast into it refreshes the in-kernel network as well."""
        from .weights import FlatWeights
        cfg = self.config
        if cfg.network != "fullyconnected":
            raise NotImplementedError("the fused search covers fully-connected networks; residual networks "
                                      "use the lock-step path with PyTorch-ROCm inference")
        flat = flat if flat is not None else FlatWeights(model)
        c, h, w = cfg.observation_shape
        desc = MzFcDesc()
        desc.observation_floats = c * h * w * (cfg.stacked_observations + 1) + cfg.stacked_observations * h * w
        desc.encoding_size = cfg.encoding_size
        for i, layers in enumerate((cfg.fc_representation_layers, cfg.fc_dynamics_layers, cfg.fc_reward_layers,
                                    cfg.fc_policy_layers, cfg.fc_value_layers)):
            if len(layers) > 3:
                raise NotImplementedError("fused FC search supports at most 3 hidden layers per MLP")
            desc.n_hidden[i] = len(layers)
            for k, width in enumerate(layers):
                desc.hidden[i][k] = int(width)
        self._check(self._lib.mzmcts_fc_configure(self._h, ctypes.byref(desc), flat.flat.data_ptr(), flat.numel))
        self._fc_model, self._fc_flat = model, flat
        self._fc_obs_floats = int(desc.observation_floats)
        self._fc_out = (torch.empty((self.E, self.F), dtype=torch.float32, device=self.device),
                        torch.empty((self.E, self.F), dtype=torch.float32, device=self.device),
                        torch.empty((self.E, self.A), dtype=torch.float32, device=self.device))
        return flat

    def group_width(self):
        """Lanes of a wavefront that own one tree."""
        g = self._group_width
        if g:
            return g
        g = 1
        while g < self.A and g < 64:
            g <<= 1
        return g

    FUSED_VARIANTS = {"auto": 0, "generic": 1, "narrow": 2}

    def set_fused_options(self, variant="auto", publish_tree=True):
        """Which whole-move kernel `search_fused` launches ("auto" | "generic" | "narrow", include/mzmcts.h) and
        whether it copies the whole tree out (export_tree / hidden pool) or only the root's children."""
        self._check(self._lib.mzmcts_set_fused_options(self._h, self.FUSED_VARIANTS[variant], 1 if publish_tree else 0))

    def fused_variant(self):
        return {0: None, 1: "generic", 2: "narrow"}[int(self._lib.mzmcts_fused_variant(self._h))]

    def fused_lds_bytes(self, hidden_in_lds=True):
        return int(self._lib.mzmcts_fused_lds_bytes(self._h, 1 if hidden_in_lds else 0))

    def fc_initial_inference(self, observations):
        """initial_inference by the library's own FC kernels: (value, reward, policy logits, hidden)."""
        obs = observations.to(self.device, dtype=torch.float32).reshape(self.E, -1).contiguous()
        v, r, p = self._fc_out
        hidden = torch.empty((self.E, self.H), dtype=torch.float32, device=self.device)
        self._check(self._lib.mzmcts_fc_initial_inference(self._h, obs.data_ptr(), v.data_ptr(), r.data_ptr(),
                                                          p.data_ptr(), hidden.data_ptr(), self._stream()))
        return v, r, p, hidden

    def fc_recurrent_inference(self, hidden, action, out_state=None):
        hidden = hidden.reshape(self.E, self.H).contiguous()
        action = action.reshape(self.E).contiguous()
        out = out_state if out_state is not None else torch.empty((self.E, self.H), dtype=torch.float32,
                                                                  device=self.device)
        v, r, p = self._fc_out
        self._check(self._lib.mzmcts_fc_recurrent_inference(self._h, hidden.data_ptr(), action.data_ptr(),
                                                            v.data_ptr(), r.data_ptr(), p.data_ptr(),
                                                            out.data_ptr(), self._stream()))
        return v, r, p, out

    def search_lockstep_fc(self, observations, legal_actions, to_play, add_exploration_noise=True, num_legal=None):
        """The lock-step pipeline with the library's FC kernels as the inference engine (no PyTorch modules):
        the reference shape the fused kernel is checked against bit for bit."""
        obs = torch.as_tensor(np.asarray(observations) if not torch.is_tensor(observations) else observations)
        with torch.cuda.device(self.device):
            v, r, p, hidden = self.fc_initial_inference(obs)
            self.begin_search(legal_actions, to_play, add_exploration_noise, num_legal)
            self.expand_roots(v, None, p, hidden)
            for _ in range(self.S):
                self.select()
                v, r, p, _ = self.fc_recurrent_inference(self.batch_hidden, self.batch_action,
                                                         out_state=self.pool[self._lib.mzmcts_next_slab(self._h)])
                self.expand_backup(v, r, p, None)
            return self.readout()

    def search_fused(self, observations, legal_actions, to_play, add_exploration_noise=True, num_legal=None):
        """MCTS.run for all envs in ONE kernel launch (after configure_fused_fc)."""
        if (torch.is_tensor(observations) and observations.is_cuda and observations.dtype == torch.float32
                and observations.is_contiguous()):
            obs = observations                      # already resident: no copies, no reshapes
        else:
            obs = torch.as_tensor(np.asarray(observations) if not torch.is_tensor(observations) else observations)
            obs = obs.to(self.device, dtype=torch.float32).reshape(self.E, -1).contiguous()
        assert obs.numel() == self.E * self._fc_obs_floats, "observation batch has the wrong size"
        with torch.cuda.device(self.device):
            self.begin_search(legal_actions, to_play, add_exploration_noise, num_legal)
            self._keep = (obs,)
            self._check(self._lib.mzmcts_search_fused_fc(self._h, obs.data_ptr(),
                                                         1 if self.fused_hidden_in_lds else 0, self._stream()))
            return self.readout()

    def search_fused_begin(self, observations, legal_actions, to_play, add_exploration_noise=True, num_legal=None):
        """Asynchronous form of search_fused: queue upload, kernel and download on this engine's stream and
        return; `readout()` later waits for them.  `observations` must be a resident fp32 CUDA tensor."""
        self.begin_search(legal_actions, to_play, add_exploration_noise, num_legal)
        self._keep = (observations,)
        self._check(self._lib.mzmcts_search_fused_fc(self._h, observations.data_ptr(),
                                                     1 if self.fused_hidden_in_lds else 0, self._stream()))
        self._check(self._lib.mzmcts_readout_begin(self._h, self._stream()))

    # ---- batches of moves without host round trips (include/mzmcts.h: mzmcts_moves_*) -------------------
    def _move_inputs(self, legal_actions, to_play, temperature, num_legal):
        if num_legal is not None:
            self._legal[:] = legal_actions
            self._nlegal[:] = num_legal
        else:
            self._legal[:] = 0
            for e, legal in enumerate(legal_actions):
                n = 0 if legal is None else len(legal)
                if n > self.A:
                    raise AssertionError("Legal actions should be a subset of the action space.")
                self._nlegal[e] = n
                if n:
                    self._legal[e, :n] = legal
        self._to_play[:] = to_play
        return np.ascontiguousarray(np.broadcast_to(np.asarray(temperature, dtype=np.float64), (self.E,)))

    def moves_prepare(self, n_moves, legal_actions, to_play, temperature, add_exploration_noise=True, num_legal=None):
        """Draw the exploration noise of the next `n_moves` moves and upload it; the legal action sets must stay
        the same over the batch, temperature (scalar or [E]) must be 0, 1 or inf."""
        t = self._move_inputs(legal_actions, to_play, temperature, num_legal)
        self._check(self._lib.mzmcts_moves_prepare(self._h, int(n_moves), self._p_legal, self._p_nlegal, self._p_to_play,
                                                   1 if add_exploration_noise else 0, ptr(t, c_f64_p), self._stream()))
        self._batch_keep, self._batch_moves = [], 0
        self._moves_ring()

    def moves_predraw_next(self, n_moves, legal_actions, to_play, temperature, add_exploration_noise=True,
                           num_legal=None):
        """While a batch is running: draw the following batch's noise (host work overlapped with the GPU)."""
        t = self._move_inputs(legal_actions, to_play, temperature, num_legal)
        self._check(self._lib.mzmcts_moves_predraw_next(self._h, int(n_moves), self._p_legal, self._p_nlegal,
                                                        self._p_to_play, 1 if add_exploration_noise else 0,
                                                        ptr(t, c_f64_p)))

    def moves_submit_next(self):
        """After moves_collect: upload the pre-drawn batch; moves_enqueue may follow."""
        self._check(self._lib.mzmcts_moves_submit_next(self._h, self._stream()))
        self._batch_keep, self._batch_moves = [], 0

    def moves_discard_next(self):
        """Drop a pre-drawn batch that will not be run (the RNG mirror goes back)."""
        self._check(self._lib.mzmcts_moves_discard_next(self._h))

    def moves_prepare_device(self, n_moves, legal_dev, num_legal_dev, to_play_dev, temperature, add_exploration_noise=True):
        """A batch of `n_moves` searches whose legal sets / players to move are DEVICE tensors (int32 [E, A], [E], [E])
        that the caller's environment kernels rewrite between the moves (games.device.DeviceEnvs.advance): board games.
        The exploration noise is drawn on the device; nothing about a move has to be known to the host in advance."""
        for t, shape in ((legal_dev, (self.E, self.A)), (num_legal_dev, (self.E,)), (to_play_dev, (self.E,))):
            assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and tuple(t.shape) == shape
        temps = np.ascontiguousarray(np.broadcast_to(np.asarray(temperature, dtype=np.float64), (self.E,)))
        self._batch_keep, self._batch_moves = [], 0
        self._batch_inputs_keep = (legal_dev, num_legal_dev, to_play_dev)
        self._check(self._lib.mzmcts_moves_prepare_device(self._h, int(n_moves), legal_dev.data_ptr(), num_legal_dev.data_ptr(),
                                                          to_play_dev.data_ptr(), 1 if add_exploration_noise else 0,
                                                          ptr(temps, c_f64_p), self._stream()))

    def moves_inputs(self, n_moves, copy=True):
        """What each move of the collected device-input batch was searched with: dict(num_legal [M, E], legal [M, E, A]
        (child slot -> action), to_play [M, E]).  copy=False: views of the library's pinned ring of this batch (valid
        until the batch after the next one is prepared: include/mzmcts.h mzmcts_moves_inputs_ring)."""
        if not copy:
            base, stride = ctypes.c_void_p(), ctypes.c_int64()
            offsets = (ctypes.c_int64 * 3)()
            self._check(self._lib.mzmcts_moves_inputs_ring(self._h, ctypes.byref(base), ctypes.byref(stride), offsets))
            raw = self._pinned_bytes(base.value, stride.value * int(n_moves))
            view = lambda off, inner: _ring_view(raw, off, stride.value, np.int32, int(n_moves), inner)
            return dict(num_legal=view(offsets[0], (self.E,)), to_play=view(offsets[1], (self.E,)),
                        legal=view(offsets[2], (self.E, self.A)))
        out = dict(num_legal=np.zeros((n_moves, self.E), np.int32), legal=np.zeros((n_moves, self.E, self.A), np.int32),
                   to_play=np.zeros((n_moves, self.E), np.int32))
        self._check(self._lib.mzmcts_moves_inputs(self._h, ptr(out["num_legal"], c_i32_p), ptr(out["legal"], c_i32_p),
                                                  ptr(out["to_play"], c_i32_p)))
        return out

    def moves_enqueue(self, observations):
        """Queue the next search of the prepared batch; `observations`: resident fp32 CUDA tensor [E, obs]."""
        assert observations.is_cuda and observations.dtype == torch.float32 and observations.is_contiguous()
        assert observations.numel() == self.E * self._fc_obs_floats, "observation batch has the wrong size"
        self._batch_keep.append(observations)
        self._batch_moves += 1
        self._check(self._lib.mzmcts_moves_enqueue(self._h, observations.data_ptr(), self._stream()))

    @torch.no_grad()
    def moves_enqueue_lockstep(self, model, observations):
        """Queue the next move of a device-input batch (moves_prepare_device) searched LOCK-STEP with `model` -- any
        network: root inference, root expansion with device-drawn noise, the S simulations (the captured hipGraph once
        there is one) and the action sampling on the device, all on the current stream, no host round trip.
        `observations`: resident fp32 CUDA tensor [E, C, H, W] (the environment kernels' output)."""
        if not (torch.is_tensor(observations) and observations.is_cuda and observations.dtype == torch.float32):
            raise TypeError("moves_enqueue_lockstep: observations must be a resident fp32 CUDA tensor")
        with torch.cuda.device(self.device):
            value, reward, policy, hidden = model.initial_inference(observations)
            self._check(self._lib.mzmcts_moves_begin_lockstep(self._h, self._stream()))
            self.expand_roots(value, reward.contiguous(), policy, hidden)
            self._run_simulations(model)
            self._check(self._lib.mzmcts_moves_end_lockstep(self._h, self._stream()))
        self._batch_keep.append(observations)
        self._batch_moves += 1

    def moves_temperature_threshold(self, threshold, game_moves):
        """play_game's temperature rule for the batch just prepared with moves_prepare_device (reference
        self_play.py:152-158): `game_moves` [E] = moves already played in each env's current game."""
        counts = np.ascontiguousarray(game_moves, dtype=np.int32)
        assert counts.shape == (self.E,)
        self._check(self._lib.mzmcts_moves_temperature_threshold(self._h, int(threshold or 0), ptr(counts, c_i32_p),
                                                                 self._stream()))
        if threshold:
            torch.cuda.current_stream(self.device).synchronize()    # (the upload reads `counts`)

    def moves_finished(self, done_dev):
        """The environment kernels' done flags (uint8 [E], device) of the move just played: envs flagged there start a
        new game at the batch's next move (their move counter restarts)."""
        assert done_dev.is_cuda and done_dev.dtype == torch.uint8 and done_dev.is_contiguous() and done_dev.numel() == self.E
        self._batch_keep.append(done_dev)
        self._check(self._lib.mzmcts_moves_finished(self._h, done_dev.data_ptr()))

    def moves_actions(self, move):
        """Device tensor (int32 [E]) holding move `move`'s sampled actions once its search has run."""
        addr = self._lib.mzmcts_moves_actions(self._h, int(move))
        if not addr:
            raise RuntimeError("no such move in the prepared batch")
        return _device_view(addr, self.E, torch.int32, self.device)

    def _pinned_bytes(self, address, nbytes):
        """numpy byte view of `nbytes` of the library's pinned memory at `address` (built once per (address, size))."""
        views = self.__dict__.setdefault("_pinned_views", {})
        key = (address, nbytes)
        if key not in views:
            if len(views) > 16:                    # (rings are re-allocated only when a larger batch is prepared)
                views.clear()
            views[key] = np.frombuffer((ctypes.c_uint8 * nbytes).from_address(address), dtype=np.uint8)
        return views[key]

    def _moves_ring(self):
        """numpy byte view of the library's pinned download ring of the batch prepared last."""
        base, stride = ctypes.c_void_p(), ctypes.c_int64()
        offsets = (ctypes.c_int64 * 8)()
        self._check(self._lib.mzmcts_moves_ring(self._h, ctypes.byref(base), ctypes.byref(stride), offsets,
                                                ctypes.byref(self._ring_moves)))
        return self._pinned_bytes(base.value, stride.value * self._ring_moves.value), stride.value, list(offsets)

    def moves_collect(self, copy=True):
        """Wait for the queued searches.  Returns dict(moves_done [E], actions [M,E], visits [M,E,A],
        root_value_sum [M,E], root_predicted [M,E], max_depth [M,E]); M = searches queued.
        copy=False: the per-move arrays are views of the library's pinned download ring (no unpacking pass):
        valid until the next moves_collect, and an env's entries in moves >= moves_done[e] are undefined."""
        M = self._batch_moves
        out = dict(moves_done=np.zeros(self.E, np.int32))
        if copy:
            out.update(actions=np.zeros((M, self.E), np.int32), visits=np.zeros((M, self.E, self.A), np.int32),
                       root_value_sum=np.zeros((M, self.E)), root_predicted=np.zeros((M, self.E), np.float32),
                       max_depth=np.zeros((M, self.E), np.int32))
            self._check(self._lib.mzmcts_moves_collect(
                self._h, ptr(out["moves_done"], c_i32_p), ptr(out["actions"], c_i32_p), ptr(out["visits"], c_i32_p),
                ptr(out["root_value_sum"], c_f64_p), ptr(out["root_predicted"], c_f32_p), ptr(out["max_depth"], c_i32_p),
                self._stream()))
        else:
            self._check(self._lib.mzmcts_moves_collect(self._h, ptr(out["moves_done"], c_i32_p), None, None, None, None,
                                                       None, self._stream()))
            raw, stride, offsets = self._moves_ring()
            view = lambda offset, dtype, inner: _ring_view(raw, offset, stride, dtype, M, inner)
            out.update(actions=view(offsets[0], np.int32, (self.E,)), visits=view(offsets[1], np.int32, (self.E, self.A)),
                       root_value_sum=view(offsets[2], np.float64, (self.E,)),
                       root_predicted=view(offsets[3], np.float32, (self.E,)), max_depth=view(offsets[4], np.int32, (self.E,)))
        self._batch_keep, self._batch_moves = [], 0
        return out

    def run_moves(self, observations, legal_actions, to_play, temperature, add_exploration_noise=True, num_legal=None):
        """`len(observations)` moves back to back (one resident observation tensor per move)."""
        with torch.cuda.device(self.device):
            self.moves_prepare(len(observations), legal_actions, to_play, temperature, add_exploration_noise, num_legal)
            for obs in observations:
                self.moves_enqueue(obs)
            return self.moves_collect()

    @torch.no_grad()
    def search(self, model, observations, legal_actions, to_play, add_exploration_noise=True, num_legal=None):
        """MCTS.run for all envs.  observations: [E, C, H, W] (numpy or tensor)."""
        if self._fc_model is not None and model is self._fc_model:
            return self.search_fused(observations, legal_actions, to_play, add_exploration_noise, num_legal)
        obs = torch.as_tensor(np.asarray(observations) if not torch.is_tensor(observations) else observations)
        obs = obs.to(self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            value, reward, policy, hidden = model.initial_inference(obs)
            self.begin_search(legal_actions, to_play, add_exploration_noise, num_legal)
            self.expand_roots(value, reward.contiguous(), policy, hidden)
            self._run_simulations(model)
            return self.readout()

    @torch.no_grad()
    def search_begin(self, model, observations, legal_actions, to_play, add_exploration_noise=True, num_legal=None):
        """Asynchronous form of the lock-step search: root inference, root expansion, the S simulations and the readout
        copies are queued on this engine's stream (`self.stream`, or the current one) and the call returns; `readout()`
        later waits for them.  `observations` must be a resident fp32 CUDA tensor.  Several engines on streams of their
        own overlap one engine's host work (contract checks, unpacking, action sampling) with the others' kernels."""
        stream = self.stream if self.stream is not None else torch.cuda.current_stream(self.device)
        if not (torch.is_tensor(observations) and observations.is_cuda and observations.dtype == torch.float32):
            raise TypeError("search_begin: observations must be a resident fp32 CUDA tensor (search() converts and copies)")
        with torch.cuda.device(self.device), torch.cuda.stream(stream):
            value, reward, policy, hidden = model.initial_inference(observations)
            self.begin_search(legal_actions, to_play, add_exploration_noise, num_legal)
            self.expand_roots(value, reward.contiguous(), policy, hidden)
            self._run_simulations(model)
            self._check(self._lib.mzmcts_readout_begin(self._h, self._stream()))

    def readout(self):
        self._check(self._lib.mzmcts_readout(self._h, ctypes.byref(self._stats_struct), self._stream()))
        if self._device_noise:          # the rows the GPU drew for this search (a host draw fills self.noise up front)
            self._check(self._lib.mzmcts_get_noise(self._h, self._p_noise))
        return self.stats

    def use_device_noise(self):
        """Exploration noise on the GPU whenever the config allows it (0 < root_dirichlet_alpha <= 1, every reference
        game); returns whether it is on."""
        if 0.0 < float(self.config.root_dirichlet_alpha) <= 1.0:
            self.set_device_noise(True)
        return self._device_noise

    def set_device_noise(self, enabled=True):
        """Draw the exploration noise (numpy.random.dirichlet, reference self_play.py:468-477) on the GPU instead of
        on the host mirrors of the RNG streams: same rows, same streams, no host work per env and move.  `self.noise`
        then holds a search's rows after its readout()."""
        self._check(self._lib.mzmcts_set_device_noise(self._h, 1 if enabled else 0))
        self._device_noise = bool(enabled)

    def sample_actions(self, temperature):
        """SelfPlay.select_action per env on its own RNG stream; returns (actions, slots)."""
        if temperature is self._temp_cache[0]:
            p_t = self._temp_cache[1]            # same ndarray object as last call: reuse its pointer
        else:
            if isinstance(temperature, np.ndarray) and temperature.dtype == np.float64 \
                    and temperature.shape == (self.E,) and temperature.flags["C_CONTIGUOUS"]:
                t = temperature
            else:
                t = np.ascontiguousarray(np.broadcast_to(np.asarray(temperature, dtype=np.float64), (self.E,)))
            p_t = ptr(t, c_f64_p)
            self._temp_cache = (temperature if t is temperature else None, p_t, t)
        self._check(self._lib.mzmcts_sample_actions(self._h, p_t, self._p_actions, self._p_slots))
        return self._actions.copy(), self._slots.copy()

    def search_statistics(self):
        """GameHistory.store_search_statistics targets: (child_visits [E,A], root_values [E])."""
        cv = np.zeros((self.E, self.A))
        rv = np.zeros(self.E)
        self._check(self._lib.mzmcts_search_statistics(self._h, ptr(cv, c_f64_p), ptr(rv, c_f64_p)))
        return cv, rv

    def set_select_queue(self, trees_per_wavefront=0):
        """Trees each wavefront of `select` works through: 0 / 1 = one descent per lane group (default), n = a
        wavefront-local queue of n trees (results do not depend on it; measured slower)."""
        self._check(self._lib.mzmcts_set_select_queue(self._h, int(trees_per_wavefront)))

    # ---- measurement -------------------------------------------------------------------------------
    def set_profiling(self, enabled):
        self._profiling = bool(enabled)
        self._check(self._lib.mzmcts_set_profiling(self._h, 1 if enabled else 0))

    def get_profile(self, reset=True):
        prof = MzProfile()
        self._check(self._lib.mzmcts_get_profile(self._h, ctypes.byref(prof), 1 if reset else 0))
        return {name: getattr(prof, name) for name, _ in MzProfile._fields_}

    def device_bytes(self):
        return int(self._lib.mzmcts_device_bytes(self._h)) + self.pool.numel() * 4

    def algorithmic_bytes_per_simulation(self, mean_depth):
        """SURVEY.md section 8(d) formula: bytes one simulation of one tree must move, split by kernel."""
        A, H, F = self.A, self.H, self.F
        two = 1 if len(self.config.players) == 2 else 0
        select = mean_depth * (8 + 24 * A) + 2 * 4 * H
        backup = (mean_depth + 1) * (28 + two) + 32 + 24 * A + 13 + 4 * (A + 2 * F)
        return dict(select=select, expand_backup=backup, total=select + backup)


def _ring_view(raw, offset, stride, dtype, n_moves, inner):
    """[n_moves] + inner view of one field of a pinned ring of per-move blocks `stride` bytes apart (read-only)."""
    itemsize = np.dtype(dtype).itemsize
    return np.lib.stride_tricks.as_strided(
        raw[offset:].view(dtype), shape=(n_moves,) + tuple(inner),
        strides=(stride,) + tuple(itemsize * int(np.prod(inner[i + 1:])) for i in range(len(inner))), writeable=False)


def _device_view(address, numel, dtype, device):
    """A torch tensor over device memory the library owns (no copy, no ownership)."""
    itemsize = torch.empty((), dtype=dtype).element_size()
    typestr = {torch.int32: "<i4", torch.float32: "<f4", torch.float64: "<f8", torch.uint8: "|u1"}[dtype]

    class _Span:
        __cuda_array_interface__ = {"shape": (int(numel),), "typestr": typestr, "data": (int(address), False),
                                    "version": 2, "strides": None}
    assert itemsize * numel > 0
    return torch.as_tensor(_Span(), device=device)


class PipelinedLockstep:
    """E envs as `groups` lock-step engines of E / groups envs, each with a replica of the network and a stream of its
    own: while the host finishes one group's move (readout, action sampling, the next move's contract checks and
    uploads) the GPU runs the other groups' simulations.  Env e of the whole set keeps its RNG stream (seeds are passed
    through in order), so the groups together play exactly what one engine of E envs plays."""

    def __init__(self, config, num_envs, model, groups=2, device=None, seeds=None, use_graph=True, group_width=0,
                 device_noise=False):
        import copy
        assert num_envs % groups == 0, "num_envs must be a multiple of groups"
        self.groups, self.per_group = groups, num_envs // groups
        seeds = list(range(num_envs)) if seeds is None else list(seeds)
        self.engines, self.models = [], []
        for g in range(groups):
            lo, hi = g * self.per_group, (g + 1) * self.per_group
            eng = BatchedMCTS(config, self.per_group, device=device, seeds=seeds[lo:hi], use_graph=use_graph,
                              group_width=group_width)
            eng.stream = torch.cuda.Stream(device=eng.device)
            if device_noise:
                eng.use_device_noise()
            self.engines.append(eng)
            # a replica per group: the network modules keep output buffers that a second search in flight would overwrite
            self.models.append(model if g == 0 else copy.deepcopy(model))

    def slice(self, g):
        return slice(g * self.per_group, (g + 1) * self.per_group)

    def refresh(self):
        """After a weight refresh of group 0's model: the other replicas follow -- parameters AND the constants cached
        from them (folded batch norms, packed / split tower weights), which a replayed hipGraph reads without ever
        coming back to Python.  Queued on each group's own stream behind the refresh of group 0 (reference
        self_play.py:37: every game that starts after the pull sees the pulled weights)."""
        state = self.models[0].state_dict()
        current = torch.cuda.current_stream(self.engines[0].device)
        for eng, m in zip(self.engines[1:], self.models[1:]):
            eng.stream.wait_stream(current)
            with torch.cuda.stream(eng.stream):
                m.set_weights(state)

    def begin(self, g, observations, legal, to_play, add_exploration_noise=True, num_legal=None):
        eng = self.engines[g]
        eng.stream.wait_stream(torch.cuda.current_stream(eng.device))
        eng.search_begin(self.models[g], observations, legal, to_play, add_exploration_noise, num_legal)

    def finish(self, g):
        """Wait for group g's search; returns its stats dict (valid until its next begin)."""
        return self.engines[g].readout()

    def close(self):
        for eng in self.engines:
            eng.close()


class PipelinedSearch:
    """Several env groups of one GPU searched concurrently: group g = engine g on its own HIP stream.

    The fused whole-move kernel is latency-bound per tree (its duration hardly depends on how many trees
    a launch carries), and the host work of a move (Dirichlet draws, action sampling, unpacking) is
    serial per group.  Splitting the E envs of a GPU into n groups and running them round-robin lets one
    group's host work overlap the other groups' kernels:

        for every group, in turn:  finish(g)  ->  host post/pre work  ->  begin(g) (async)

    Env indices stay global: group g owns envs [g*E/n, (g+1)*E/n) with their global RNG seeds."""

    def __init__(self, config, num_envs, model, flat, groups=4, device=None, seeds=None, group_width=0):
        assert num_envs % groups == 0, "num_envs must divide evenly into groups"
        self.E, self.groups, self.per_group = int(num_envs), int(groups), int(num_envs) // int(groups)
        if seeds is None:
            seeds = [int(config.seed) + e for e in range(self.E)]
        self.engines = []
        for g in range(self.groups):
            lo, hi = g * self.per_group, (g + 1) * self.per_group
            eng = BatchedMCTS(config, self.per_group, device=device, seeds=seeds[lo:hi], group_width=group_width)
            eng.configure_fused_fc(model, flat)
            eng.stream = torch.cuda.Stream(device=eng.device)
            self.engines.append(eng)
        self._in_flight = [False] * self.groups

    def slice(self, g):
        return slice(g * self.per_group, (g + 1) * self.per_group)

    def begin(self, g, observations, legal, to_play, add_exploration_noise=True, num_legal=None):
        """Queue group g's next search (observations: that group's resident [E/n, obs] tensor)."""
        eng = self.engines[g]
        eng.stream.wait_stream(torch.cuda.current_stream(eng.device))
        eng.search_fused_begin(observations, legal, to_play, add_exploration_noise, num_legal)
        self._in_flight[g] = True

    def finish(self, g):
        """Wait for group g's search; returns its stats dict (valid until its next begin)."""
        self._in_flight[g] = False
        return self.engines[g].readout()

    def close(self):
        for eng in self.engines:
            eng.close()
