"""ctypes binding of oracle/libmz_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
It also restates, in numpy, the two host-side GameHistory helpers of the reference
(self_play.py:497-548) that have no arithmetic worth putting in C.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_int_p = ctypes.POINTER(ctypes.c_int)
c_double_p = ctypes.POINTER(ctypes.c_double)
c_float_p = ctypes.POINTER(ctypes.c_float)
c_u32_p = ctypes.POINTER(ctypes.c_uint32)


class OracleConfig(ctypes.Structure):
    _fields_ = [("A", ctypes.c_int), ("S", ctypes.c_int), ("n_players", ctypes.c_int),
                ("support_size", ctypes.c_int), ("H", ctypes.c_int),
                ("discount", ctypes.c_double), ("pb_c_base", ctypes.c_double),
                ("pb_c_init", ctypes.c_double), ("dirichlet_alpha", ctypes.c_double),
                ("exploration_fraction", ctypes.c_double)]


class OracleLog(ctypes.Structure):
    _fields_ = [("sim_depth", c_int_p), ("sim_actions", c_int_p), ("sim_ties", c_int_p),
                ("sim_value", c_double_p), ("sim_reward", c_double_p), ("sim_priors", c_double_p)]


RECURRENT_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, c_float_p, ctypes.c_int, c_float_p,
                                c_float_p, c_float_p, c_float_p)


def build(force=False):
    so = os.path.join(_HERE, "libmz_oracle.so")
    src = os.path.join(_HERE, "mz_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libmz_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.oracle_rng_sizeof.restype = ctypes.c_size_t
        L.oracle_rng_u32.restype = ctypes.c_uint32
        L.oracle_rng_double.restype = ctypes.c_double
        L.oracle_rng_below.restype = ctypes.c_uint32
        L.oracle_rng_below.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.oracle_rng_gamma.restype = ctypes.c_double
        L.oracle_rng_gamma.argtypes = [ctypes.c_void_p, ctypes.c_double]
        L.oracle_rng_dirichlet.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int, c_double_p]
        L.oracle_rng_choice_p.argtypes = [ctypes.c_void_p, c_double_p, ctypes.c_int]
        L.oracle_rng_words.restype = ctypes.c_uint64
        L.oracle_rng_seed.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.oracle_rng_set_state.argtypes = [ctypes.c_void_p, c_u32_p, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_double]
        L.oracle_rng_get_state.argtypes = [ctypes.c_void_p, c_u32_p, c_int_p, c_int_p, c_double_p]
        L.oracle_support_to_scalar.restype = ctypes.c_float
        L.oracle_support_to_scalar.argtypes = [c_float_p, ctypes.c_int]
        L.oracle_softmax_f32.argtypes = [c_float_p, ctypes.c_int, c_float_p]
        L.oracle_fc_create.restype = ctypes.c_void_p
        L.oracle_fc_create.argtypes = [c_float_p] + [ctypes.c_int] * 4 + [c_int_p, ctypes.c_int] * 5
        L.oracle_fc_destroy.argtypes = [ctypes.c_void_p]
        L.oracle_fc_initial.argtypes = [ctypes.c_void_p] + [c_float_p] * 5
        L.oracle_fc_recurrent.argtypes = [ctypes.c_void_p, c_float_p, ctypes.c_int] + [c_float_p] * 4
        L.oracle_tree_create.restype = ctypes.c_void_p
        L.oracle_tree_create.argtypes = [ctypes.POINTER(OracleConfig)]
        L.oracle_tree_destroy.argtypes = [ctypes.c_void_p]
        L.oracle_tree_reset.argtypes = [ctypes.c_void_p, ctypes.c_void_p, c_int_p, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_double, c_float_p, c_double_p,
                                        c_float_p, ctypes.c_int, c_double_p]
        L.oracle_tree_simulate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                           ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.POINTER(OracleLog)]
        L.oracle_tree_root_stats.argtypes = [ctypes.c_void_p, c_int_p, c_double_p, c_double_p,
                                             c_double_p, c_double_p, c_int_p, c_int_p, c_double_p,
                                             c_double_p]
        L.oracle_tree_node_stats.argtypes = [ctypes.c_void_p, c_int_p, ctypes.c_int, c_int_p,
                                             c_double_p, c_double_p, c_double_p, c_int_p, c_int_p]
        L.oracle_select_action.argtypes = [ctypes.c_void_p, c_int_p, ctypes.c_int, ctypes.c_double]
        L.oracle_search_statistics.argtypes = [ctypes.c_void_p, c_double_p, c_double_p]
        L.oracle_fc_selfplay_moves.restype = ctypes.c_long
        L.oracle_fc_selfplay_moves.argtypes = [ctypes.POINTER(OracleConfig), ctypes.c_void_p,
                                               ctypes.c_void_p, c_float_p, ctypes.c_int,
                                               ctypes.c_double, c_int_p, c_double_p, c_int_p,
                                               ctypes.POINTER(ctypes.c_long)]
        _LIB = L
    return _LIB


def _p(arr, typ):
    return None if arr is None else arr.ctypes.data_as(typ)


class Rng:
    """numpy legacy RandomState clone (one stream)."""

    def __init__(self, seed=None):
        self._buf = ctypes.create_string_buffer(lib().oracle_rng_sizeof())
        self.ptr = ctypes.cast(self._buf, ctypes.c_void_p)
        if seed is not None:
            self.seed(seed)

    def seed(self, seed):
        lib().oracle_rng_seed(self.ptr, int(seed) & 0xFFFFFFFF)

    def u32(self):
        return lib().oracle_rng_u32(self.ptr)

    def double(self):
        return lib().oracle_rng_double(self.ptr)

    def below(self, n):
        return lib().oracle_rng_below(self.ptr, n)

    def gamma(self, shape):
        return lib().oracle_rng_gamma(self.ptr, shape)

    def dirichlet(self, alpha, k):
        out = np.zeros(k, dtype=np.float64)
        lib().oracle_rng_dirichlet(self.ptr, alpha, k, _p(out, c_double_p))
        return out

    def choice_p(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        return lib().oracle_rng_choice_p(self.ptr, _p(p, c_double_p), len(p))

    @property
    def words(self):
        return lib().oracle_rng_words(self.ptr)

    def set_numpy_state(self, state):
        key = np.ascontiguousarray(state[1], dtype=np.uint32)
        lib().oracle_rng_set_state(self.ptr, _p(key, c_u32_p), int(state[2]), int(state[3]),
                                   float(state[4]))

    def get_numpy_state(self):
        key = np.zeros(624, dtype=np.uint32)
        pos, hg, g = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        lib().oracle_rng_get_state(self.ptr, _p(key, c_u32_p), ctypes.byref(pos), ctypes.byref(hg),
                                   ctypes.byref(g))
        return ("MT19937", key, pos.value, hg.value, g.value)


def support_to_scalar(logits, support_size):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    out = np.zeros(logits.shape[0], dtype=np.float32)
    for i in range(logits.shape[0]):
        out[i] = lib().oracle_support_to_scalar(_p(logits[i], c_float_p), support_size)
    return out


def softmax_f32(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros_like(x)
    lib().oracle_softmax_f32(_p(x, c_float_p), x.size, _p(out, c_float_p))
    return out


def make_config(A, S, n_players, discount, pb_c_base, pb_c_init, alpha, frac, support_size, H=0):
    return OracleConfig(A=A, S=S, n_players=n_players, support_size=support_size, H=H,
                        discount=discount, pb_c_base=pb_c_base, pb_c_init=pb_c_init,
                        dirichlet_alpha=alpha, exploration_fraction=frac)


def config_from_fixture(fx, H=0):
    return make_config(int(fx["cfg_A"]), int(fx["cfg_S"]), int(fx["cfg_players"]),
                       float(fx["cfg_discount"]), float(fx["cfg_pb_c_base"]),
                       float(fx["cfg_pb_c_init"]), float(fx["cfg_alpha"]), float(fx["cfg_frac"]),
                       int(fx["cfg_support"]), H)


def config_from_muzero(config, H=0):
    return make_config(len(config.action_space), config.num_simulations, len(config.players),
                       float(config.discount), float(config.pb_c_base), float(config.pb_c_init),
                       float(config.root_dirichlet_alpha), float(config.root_exploration_fraction),
                       int(config.support_size), H)


class FcNet:
    """C restatement of MuZeroFullyConnectedNetwork's inference half."""

    ORDER = ["representation_network", "dynamics_encoded_state_network", "dynamics_reward_network",
             "prediction_policy_network", "prediction_value_network"]

    def __init__(self, weights, obs_size, enc, A, support_size, repr_h, dyn_h, rew_h, pol_h, val_h):
        flat = []
        for net in self.ORDER:
            keys = sorted((k for k in weights if k.startswith(net + ".")),
                          key=lambda k: (int(k.split(".")[-2]), k.endswith("bias")))
            for k in keys:
                flat.append(np.asarray(weights[k], dtype=np.float32).reshape(-1))
        self.flat = np.ascontiguousarray(np.concatenate(flat))
        self.obs_size, self.enc, self.A, self.F = obs_size, enc, A, 2 * support_size + 1

        def arr(h):
            return np.ascontiguousarray(h if len(h) else [0], dtype=np.int32), len(h)

        args = []
        self._keep = []
        for h in (repr_h, dyn_h, rew_h, pol_h, val_h):
            a, n = arr(h)
            self._keep.append(a)
            args += [_p(a, c_int_p), n]
        self.ptr = lib().oracle_fc_create(_p(self.flat, c_float_p), obs_size, enc, A, support_size,
                                          *args)

    @classmethod
    def from_config(cls, config, weights):
        obs = config.observation_shape
        obs_size = (obs[0] * obs[1] * obs[2] * (config.stacked_observations + 1)
                    + config.stacked_observations * obs[1] * obs[2])
        return cls(weights, obs_size, config.encoding_size, len(config.action_space),
                   config.support_size, config.fc_representation_layers, config.fc_dynamics_layers,
                   config.fc_reward_layers, config.fc_policy_layers, config.fc_value_layers)

    def __del__(self):
        try:
            lib().oracle_fc_destroy(self.ptr)
        except Exception:
            pass

    def initial(self, obs):
        obs = np.ascontiguousarray(obs, dtype=np.float32).reshape(-1)
        v, r = np.zeros(self.F, np.float32), np.zeros(self.F, np.float32)
        p, h = np.zeros(self.A, np.float32), np.zeros(self.enc, np.float32)
        lib().oracle_fc_initial(self.ptr, _p(obs, c_float_p), _p(v, c_float_p), _p(r, c_float_p),
                                _p(p, c_float_p), _p(h, c_float_p))
        return v, r, p, h

    def recurrent(self, hidden, action):
        hidden = np.ascontiguousarray(hidden, dtype=np.float32).reshape(-1)
        v, r = np.zeros(self.F, np.float32), np.zeros(self.F, np.float32)
        p, h = np.zeros(self.A, np.float32), np.zeros(self.enc, np.float32)
        lib().oracle_fc_recurrent(self.ptr, _p(hidden, c_float_p), int(action), _p(v, c_float_p),
                                  _p(r, c_float_p), _p(p, c_float_p), _p(h, c_float_p))
        return v, r, p, h


class Tree:
    """One search tree (reference MCTS.run semantics)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.ptr = lib().oracle_tree_create(ctypes.byref(cfg))
        self.n_legal = 0
        S, A, D = cfg.S, cfg.A, cfg.S + 1
        self.sim_depth = np.zeros(S, np.int32)
        self.sim_actions = np.full((S, D), -1, np.int32)
        self.sim_ties = np.zeros((S, D), np.int32)
        self.sim_value = np.zeros(S, np.float64)
        self.sim_reward = np.zeros(S, np.float64)
        self.sim_priors = np.zeros((S, A), np.float64)
        self._log = OracleLog(_p(self.sim_depth, c_int_p), _p(self.sim_actions, c_int_p),
                              _p(self.sim_ties, c_int_p), _p(self.sim_value, c_double_p),
                              _p(self.sim_reward, c_double_p), _p(self.sim_priors, c_double_p))

    def __del__(self):
        try:
            lib().oracle_tree_destroy(self.ptr)
        except Exception:
            pass

    def reset(self, rng, legal, to_play, root_reward, root_policy_logits=None, root_priors=None,
              root_hidden=None, add_noise=True):
        legal = np.ascontiguousarray(legal, dtype=np.int32)
        self.n_legal = len(legal)
        self.sim_actions[:] = -1
        self.sim_ties[:] = 0
        pl = None if root_policy_logits is None else np.ascontiguousarray(root_policy_logits, np.float32)
        pr = None if root_priors is None else np.ascontiguousarray(root_priors, np.float64)
        hid = None if root_hidden is None else np.ascontiguousarray(root_hidden, np.float32).reshape(-1)
        noise = np.zeros(max(self.n_legal, 1), np.float64)
        rc = lib().oracle_tree_reset(self.ptr, rng.ptr, _p(legal, c_int_p), self.n_legal, int(to_play),
                                     float(root_reward), _p(pl, c_float_p), _p(pr, c_double_p),
                                     _p(hid, c_float_p), int(bool(add_noise)), _p(noise, c_double_p))
        if rc == -1:
            raise AssertionError(f"Legal actions should not be an empty array. Got {list(legal)}.")
        if rc == -2:
            raise AssertionError("Legal actions should be a subset of the action space.")
        return noise

    def simulate(self, rng, first=0, n=None, value=None, reward=None, priors=None, callback=None):
        n = self.cfg.S - first if n is None else n
        if callback is not None:
            cb = RECURRENT_CB(callback)
            lib().oracle_tree_simulate(self.ptr, rng.ptr, first, n, None, None, None,
                                       ctypes.cast(cb, ctypes.c_void_p), None, ctypes.byref(self._log))
        else:
            v = np.ascontiguousarray(value, np.float64)
            r = np.ascontiguousarray(reward, np.float64)
            p = np.ascontiguousarray(priors, np.float64)
            lib().oracle_tree_simulate(self.ptr, rng.ptr, first, n, _p(v, c_double_p),
                                       _p(r, c_double_p), _p(p, c_double_p), None, None,
                                       ctypes.byref(self._log))

    def root_stats(self):
        n = self.n_legal
        visits = np.zeros(n, np.int32)
        vs, pr, rw = np.zeros(n), np.zeros(n), np.zeros(n)
        rvs, mn, mx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        rv, md = ctypes.c_int(), ctypes.c_int()
        lib().oracle_tree_root_stats(self.ptr, _p(visits, c_int_p), _p(vs, c_double_p),
                                     _p(pr, c_double_p), _p(rw, c_double_p), ctypes.byref(rvs),
                                     ctypes.byref(rv), ctypes.byref(md), ctypes.byref(mn),
                                     ctypes.byref(mx))
        return dict(visits=visits, child_value_sum=vs, child_prior=pr, child_reward=rw,
                    root_value_sum=rvs.value, root_visit=rv.value, max_tree_depth=md.value,
                    mms_min=mn.value, mms_max=mx.value)

    def node_stats(self, actions):
        a = np.ascontiguousarray(actions if len(actions) else [0], dtype=np.int32)
        visit, tp, nc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        vs, pr, rw = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        rc = lib().oracle_tree_node_stats(self.ptr, _p(a, c_int_p), len(actions), ctypes.byref(visit),
                                          ctypes.byref(vs), ctypes.byref(pr), ctypes.byref(rw),
                                          ctypes.byref(tp), ctypes.byref(nc))
        if rc != 0:
            return None
        return dict(visit=visit.value, value_sum=vs.value, prior=pr.value, reward=rw.value,
                    to_play=tp.value, n_children=nc.value)

    def search_statistics(self):
        cv = np.zeros(self.cfg.A)
        rv = ctypes.c_double()
        lib().oracle_search_statistics(self.ptr, _p(cv, c_double_p), ctypes.byref(rv))
        return cv, rv.value


def select_action(rng, visits, temperature):
    """Returns the child SLOT chosen (reference SelfPlay.select_action)."""
    v = np.ascontiguousarray(visits, dtype=np.int32)
    t = -1.0 if temperature == float("inf") else float(temperature)
    return lib().oracle_select_action(rng.ptr, _p(v, c_int_p), len(v), t)


def fc_selfplay_moves(cfg, net, rng, observations, temperature=1.0):
    """cpu_baseline leg: one MCTS.run + select_action per observation, single thread."""
    obs = np.ascontiguousarray(observations, dtype=np.float32).reshape(len(observations), -1)
    n = obs.shape[0]
    visits = np.zeros((n, cfg.A), np.int32)
    rootv = np.zeros(n)
    actions = np.zeros(n, np.int32)
    depth = ctypes.c_long(0)
    sims = lib().oracle_fc_selfplay_moves(ctypes.byref(cfg), net.ptr, rng.ptr, _p(obs, c_float_p), n,
                                          float(temperature), _p(visits, c_int_p),
                                          _p(rootv, c_double_p), _p(actions, c_int_p),
                                          ctypes.byref(depth))
    return dict(sims=sims, visits=visits, root_value=rootv, actions=actions, depth_sum=depth.value)


# ---------------------------------------------------------------------------------------
# GameHistory host helpers (self_play.py:497-548), numpy restatement
# ---------------------------------------------------------------------------------------
def store_search_statistics(visits_by_action, action_space):
    """visits_by_action: {action: visit_count} of the root's children -> policy target."""
    total = sum(visits_by_action.values())
    return [visits_by_action[a] / total if a in visits_by_action else 0 for a in action_space]


def get_stacked_observations(observation_history, action_history, index, num_stacked):
    index = index % len(observation_history)
    stacked = np.array(observation_history[index]).copy()
    for past in reversed(range(index - num_stacked, index)):
        if 0 <= past:
            prev = np.concatenate((observation_history[past],
                                   [np.ones_like(stacked[0]) * action_history[past + 1]]))
        else:
            prev = np.concatenate((np.zeros_like(observation_history[index]),
                                   [np.zeros_like(stacked[0])]))
        stacked = np.concatenate((stacked, prev))
    return stacked
