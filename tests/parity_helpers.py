"""Shared helpers of the parity tests (and of __graft_entry__.smoke / bench.py's checker legs).

"Injected mode": per-simulation network outputs (value, reward, priors) come from a fixture or a
seeded generator instead of a network, so the tree arithmetic of the HIP engine can be compared
BIT-EXACTLY with the oracle / the reference's recorded traces.
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for _p in (ROOT, GOLDEN, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def fixture_config(fx, base_config=None, H=0):
    """A MuZeroConfig-like attribute bag carrying the search constants of a trace fixture."""
    cfg = base_config if base_config is not None else types.SimpleNamespace()
    cfg.action_space = list(range(int(fx["cfg_A"])))
    cfg.players = list(range(int(fx["cfg_players"])))
    cfg.num_simulations = int(fx["cfg_S"])
    d = float(fx["cfg_discount"])
    cfg.discount = int(d) if d == int(d) else d
    cfg.pb_c_base = float(fx["cfg_pb_c_base"])
    cfg.pb_c_init = float(fx["cfg_pb_c_init"])
    cfg.root_dirichlet_alpha = float(fx["cfg_alpha"])
    cfg.root_exploration_fraction = float(fx["cfg_frac"])
    cfg.support_size = int(fx["cfg_support"])
    if base_config is None:
        cfg.seed = 0
        cfg.network = "fullyconnected"
        cfg.encoding_size = max(H, 4)
        cfg.observation_shape = (1, 1, 4)
        cfg.downsample = False
    return cfg


def make_search_config(A, S, players, discount, alpha=0.25, frac=0.25, support=10, pb_c_base=19652,
                       pb_c_init=1.25, H=4):
    return types.SimpleNamespace(
        action_space=list(range(A)), players=list(range(players)), num_simulations=S,
        discount=discount, pb_c_base=pb_c_base, pb_c_init=pb_c_init, root_dirichlet_alpha=alpha,
        root_exploration_fraction=frac, support_size=support, seed=0, network="fullyconnected",
        encoding_size=H, observation_shape=(1, 1, 4), downsample=False)


# ---- injected streams --------------------------------------------------------------------------
def streams_from_fixture(fx, idx):
    """dict of per-tree inputs for `idx` traces of a golden trace file."""
    idx = list(idx)
    return dict(
        seeds=[int(fx["seed"][i]) for i in idx],
        legal=[fx["legal"][i][: int(fx["n_legal"][i])].tolist() for i in idx],
        to_play=[int(fx["to_play"][i]) for i in idx],
        root_reward=np.array([fx["root_reward"][i] for i in idx], dtype=np.float64),
        root_priors=np.stack([fx["root_priors"][i] for i in idx]).astype(np.float64),
        value=np.stack([fx["sim_value"][i] for i in idx]).astype(np.float64),      # [T,S]
        reward=np.stack([fx["sim_reward"][i] for i in idx]).astype(np.float64),    # [T,S]
        priors=np.stack([fx["sim_priors"][i] for i in idx]).astype(np.float64),    # [T,S,A]
    )


def random_streams(T, A, S, seed, n_players=1, ties=False, min_legal=1):
    """Seeded synthetic injected streams for configurations no golden trace covers.  Values and
    rewards are fp32-representable like the reference's `.item()` results."""
    rs = np.random.RandomState(seed)
    legal = []
    for _ in range(T):
        n = int(rs.randint(min_legal, A + 1))
        legal.append(sorted(rs.choice(A, size=n, replace=False).tolist()))

    def f32(x):
        return np.asarray(x, dtype=np.float32).astype(np.float64)

    def priors(shape):
        p = rs.dirichlet([0.6] * A, size=shape).astype(np.float32)
        if ties:
            p = np.round(p * 4) / 4 + np.float32(0.125)      # many exactly-equal priors
            p = (p / p.sum(axis=-1, keepdims=True)).astype(np.float32)
        return p.astype(np.float64)

    root_priors = np.zeros((T, A))
    for t in range(T):
        n = len(legal[t])
        p = rs.dirichlet([0.8] * n).astype(np.float32)
        root_priors[t, :n] = p
    scale = 3.0 if n_players == 2 else 30.0
    return dict(
        seeds=[int(s) for s in rs.randint(0, 2**31 - 1, size=T)],
        legal=legal,
        to_play=[int(p) for p in rs.randint(0, n_players, size=T)],
        root_reward=np.zeros(T),
        root_priors=root_priors,
        value=f32(scale * rs.standard_normal((T, S))),
        reward=f32(rs.standard_normal((T, S)) * (rs.random_sample((T, S)) < 0.5)),
        priors=priors((T, S)),
    )


def run_injected_on_oracle(oracle, cfg_or_fx, streams=None, idx=None, temperature=1.0):
    """Replay injected streams through the C oracle, one tree at a time."""
    if streams is None:
        fx = cfg_or_fx
        streams = streams_from_fixture(fx, idx)
        ocfg = oracle.config_from_fixture(fx)
    else:
        ocfg = oracle.config_from_muzero(cfg_or_fx)
    T, A, S = len(streams["seeds"]), ocfg.A, ocfg.S
    out = _result_arrays(T, A, S)
    temps = np.broadcast_to(np.asarray(temperature, dtype=np.float64), (T,))
    for t in range(T):
        rng = oracle.Rng(streams["seeds"][t])
        tree = oracle.Tree(ocfg)
        n = len(streams["legal"][t])
        noise = tree.reset(rng, streams["legal"][t], streams["to_play"][t], float(streams["root_reward"][t]),
                           root_priors=streams["root_priors"][t][:n], add_noise=True)
        tree.simulate(rng, value=streams["value"][t], reward=streams["reward"][t], priors=streams["priors"][t])
        st = tree.root_stats()
        out["noise"][t, :n] = noise[:n]
        out["visits"][t, :n] = st["visits"]
        out["child_value_sum"][t, :n] = st["child_value_sum"]
        out["child_prior"][t, :n] = st["child_prior"]
        out["child_reward"][t, :n] = st["child_reward"]
        out["root_value_sum"][t] = st["root_value_sum"]
        out["root_visits"][t] = st["root_visit"]
        out["max_tree_depth"][t] = st["max_tree_depth"]
        out["min_max"][t] = (st["mms_min"], st["mms_max"])
        out["sim_depth"][t] = tree.sim_depth
        out["sim_actions"][t] = tree.sim_actions[:, :S]
        out["sim_ties"][t] = tree.sim_ties[:, :S]
        words_before = rng.words
        cv, rv = tree.search_statistics()
        out["child_visits_target"][t] = cv
        out["root_value_target"][t] = rv
        slot = oracle.select_action(rng, st["visits"], float(temps[t]))
        out["action"][t] = streams["legal"][t][slot]
        out["rng_words_run"][t] = words_before
        out["rng_words_total"][t] = rng.words
    return out


def _result_arrays(T, A, S):
    return dict(
        noise=np.zeros((T, A)), visits=np.zeros((T, A), np.int32), child_value_sum=np.zeros((T, A)),
        child_prior=np.zeros((T, A)), child_reward=np.zeros((T, A)), root_value_sum=np.zeros(T),
        root_visits=np.zeros(T, np.int32), max_tree_depth=np.zeros(T, np.int32), min_max=np.zeros((T, 2)),
        sim_depth=np.zeros((T, S), np.int32), sim_actions=np.full((T, S, S), -1, np.int32),
        sim_ties=np.zeros((T, S, S), np.int32), child_visits_target=np.zeros((T, A)),
        root_value_target=np.zeros(T), action=np.zeros(T, np.int32), rng_words_run=np.zeros(T, np.int64),
        rng_words_total=np.zeros(T, np.int64))


def run_injected_on_engine(engine_mod, config, fx_or_streams, idx=None, device="cuda", temperature=1.0,
                           record_paths=True, repeat=1, engine=None, group_width=0, fused_step=False, select_queue=0,
                           device_noise=False):
    """Drive the HIP engine through the C ABI with injected streams.

    `repeat` tiles the T trees `repeat` times (env e replays stream e % T), which exercises batching:
    every copy must come out identical.  Returns arrays for the first T envs plus `all_equal`."""
    import torch
    if idx is not None:
        streams = streams_from_fixture(fx_or_streams, idx)
        config = fixture_config(fx_or_streams, config)
    else:
        streams = fx_or_streams
    T = len(streams["seeds"])
    E = T * repeat
    A, S = len(config.action_space), config.num_simulations
    own = engine is None
    if own:
        engine = engine_mod.BatchedMCTS(config, E, device=device, seeds=streams["seeds"] * repeat, group_width=group_width)
    else:
        engine.seed(streams["seeds"] * repeat)
    if record_paths:
        engine.set_debug_ties(True)
    if device_noise:
        engine.set_device_noise(True)            # the GPU draws the Dirichlet rows (they reach the host at readout)
    if select_queue:
        engine.set_select_queue(select_queue)   # trees per wavefront of `select` (0 = one descent per lane group)

    def tile(a):
        return np.concatenate([a] * repeat, axis=0)

    engine.begin_search(streams["legal"] * repeat, streams["to_play"] * repeat, True)
    noise = engine.noise.copy()
    engine.expand_roots_injected(tile(streams["root_reward"]), tile(streams["root_priors"]))
    out = _result_arrays(E, A, S)
    value, reward, priors = tile(streams["value"]), tile(streams["reward"]), tile(streams["priors"])
    for s in range(S):
        if s == 0 or not fused_step:
            engine.select(gather=False)
        if record_paths:
            depth, actions, ties = engine.last_paths(with_ties=True)
            out["sim_depth"][:, s] = depth
            out["sim_actions"][:, s, :] = actions
            out["sim_ties"][:, s, :] = ties
        if fused_step and s + 1 < S:     # expand_backup(s) + select(s + 1) in one launch
            engine.expand_backup_select_injected(value[:, s], reward[:, s], priors[:, s, :])
        else:
            engine.expand_backup_injected(value[:, s], reward[:, s], priors[:, s, :])
    st = engine.readout()
    out["noise"][:] = engine.noise if device_noise else noise
    for key in ("visits", "child_value_sum", "child_prior", "child_reward", "root_value_sum", "root_visits",
                "max_tree_depth", "min_max"):
        out[key][:] = st[key]
    out["tie_break_words"] = st["tie_break_words"].copy()
    out["depth_sum"] = st["depth_sum"].copy()
    cv, rv = engine.search_statistics()
    out["child_visits_target"][:] = cv
    out["root_value_target"][:] = rv
    actions, _ = engine.sample_actions(np.broadcast_to(np.asarray(temperature, dtype=np.float64), (T,)).tolist() * repeat)
    out["action"][:] = actions
    torch.cuda.synchronize()
    all_equal = True
    if repeat > 1:
        for key, arr in out.items():
            base = arr[:T]
            for r in range(1, repeat):
                if not np.array_equal(base, arr[r * T:(r + 1) * T]):
                    all_equal = False
    result = {k: v[:T] for k, v in out.items()}
    result["all_equal"] = all_equal
    if own:
        engine.close()
    return result


# ---- models ---------------------------------------------------------------------------------------
def cartpole_model_and_weights(models_mod, config, device="cpu"):
    """The reference's trained CartPole network (weights fixture) on `device`."""
    import torch
    w = load_golden("cartpole_weights")
    weights = {k: torch.from_numpy(w[k]) for k in w.files}
    model = models_mod.MuZeroNetwork(config)
    model.set_weights(weights)
    model.to(device)
    model.eval()
    return model, weights


def synthetic_model(models_mod, config, device="cpu", seed=0):
    """Network with the deterministic synthetic weights of tests/golden/synth.py."""
    import torch
    from synth import synthetic_state_dict
    model = models_mod.MuZeroNetwork(config)
    sd = synthetic_state_dict(model.state_dict(), seed)
    weights = {k: torch.from_numpy(v) for k, v in sd.items()}
    model.set_weights(weights)
    model.to(device)
    model.eval()
    return model, weights


# ---- the value transform seen as an error amplifier (reference models.py:641-662) ----------------------
def categorical_mean(logits, support_size):
    """x = sum(softmax(logits) * [-s .. s]) in float64 (the quantity the inverse transform is applied to)."""
    logits = np.asarray(logits, dtype=np.float64)
    p = np.exp(logits - logits.max(axis=-1, keepdims=True))
    p /= p.sum(axis=-1, keepdims=True)
    return (p * np.arange(-support_size, support_size + 1, dtype=np.float64)).sum(axis=-1)


def categorical_mean_bound(logit_deviation, support_size):
    """|x' - x| for logits that differ by at most `logit_deviation` per entry: the soft-max moves by
    ||p' - p||_1 <= 2 * ||l' - l||_inf (to first order; 1 % margin for the rest) and |x' - x| <= s * ||p' - p||_1."""
    return 2.02 * support_size * logit_deviation


def value_transform_bound(value, delta_x, eps=0.001):
    """Bound on |support_to_scalar(l') - support_to_scalar(l)| for decoded value `value` (either of the two) when the
    categorical means differ by at most delta_x, both decodes evaluated in float32 in the reference's operation order:

        v = sign(x) * (z**2 - 1),   z = (sqrt(1 + 4 eps (|x| + 1 + eps)) - 1) / (2 eps)       (models.py:655-660)

    * analytic part: dv/dx = 2 z / u with u = 1 + 2 eps z (= 2 at x = 0, growing like 2 sqrt(|v| + 1)): kappa * delta_x;
    * granularity of the float32 evaluation, by forward error analysis of the seven operations (half an ulp each; the
      soft-max and the weighted sum that produce x are charged 8 ulps).  What dominates: w = 1 + 4 eps (...) lies in
      [1, 2) for |x| < 249, where float32 numbers are 2**-23 apart, and so does u = sqrt(w); u - 1 is exact, so
      z = (u - 1) / (2 eps) lives on a lattice of spacing 2**-23 / (2 eps) = 6.0e-5 and v = z**2 - 1 moves in steps of
      2 z * 6.0e-5 = 1.2e-4 * sqrt(|v| + 1).  The reference's own output cannot resolve anything finer
      (tests/test_value_bound.py shows fixture G1 sitting on that lattice); each of the two evaluations carries this error.
    Returns kappa(|x| + delta_x) * delta_x + 2 * (rounding error of one evaluation)."""
    f32 = np.float32

    def ulp(a):
        return np.spacing(np.asarray(a, dtype=np.float64).astype(f32)).astype(np.float64)

    value = np.abs(np.asarray(value, dtype=np.float64))
    delta_x = np.asarray(delta_x, dtype=np.float64)
    z = np.sqrt(value + 1.0) + delta_x                     # (z grows by at most dz/dx * delta_x <= delta_x)
    x = z - 1.0 + eps * (z * z - 1.0)                      # the forward transform h(v), models.py:665-671
    u = 1.0 + 2.0 * eps * z
    kappa = 2.0 * z / u
    e_x = 8.0 * ulp(np.maximum(x, 1.0))
    t = x + 1.0 + eps
    e_t = e_x + ulp(t)                                     # two additions
    e_m = 4.0 * eps * e_t + 0.5 * ulp(4.0 * eps * t)
    e_w = e_m + 0.5 * ulp(u * u)
    e_u = e_w / (2.0 * u) + 0.5 * ulp(u)
    e_z = e_u / (2.0 * eps) + 0.5 * ulp(z)                 # (u - 1 is exact)
    e_s = 2.0 * z * e_z + 0.5 * ulp(z * z)
    e_v = e_s + 0.5 * ulp(np.maximum(z * z - 1.0, 1e-30))
    return 1.01 * kappa * delta_x + 2.0 * e_v
