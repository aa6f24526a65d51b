"""Parity tests proper: the HIP engine, called through the C ABI, against the oracle and the golden
fixtures recorded from the reference.  Needs an MI355X (`pytest -m gpu`).

  injected mode  network outputs replayed -> every integer and every fp64 statistic BIT-EXACT
  native mode    PyTorch-ROCm inference   -> logits / priors within 1e-5, decoded values within 3e-5
                 relative (fp32 conditioning of the inverse value transform, see
                 tests/test_oracle_mcts.py), policy targets identical wherever paths are identical
"""
import importlib
import os

import numpy as np
import pytest
import torch

from parity_helpers import (cartpole_model_and_weights, fixture_config, load_golden, make_search_config,
                            random_streams, run_injected_on_engine, run_injected_on_oracle,
                            streams_from_fixture, synthetic_model)

pytestmark = pytest.mark.gpu

TRACE_FILES = ["g4_cartpole_traces", "g5_tictactoe_traces", "g5_connect4_traces", "g5_cartpole_ties_traces",
               "g5_atari84_traces"]
EXACT_KEYS = ["noise", "visits", "child_value_sum", "child_prior", "child_reward", "root_value_sum",
              "root_visits", "max_tree_depth", "min_max", "sim_depth", "sim_actions", "sim_ties",
              "child_visits_target", "root_value_target", "action"]


@pytest.fixture(scope="module")
def eng(pkg):
    importlib.import_module("muzero-hypermodel_amd.build").build_native()
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return importlib.import_module("muzero-hypermodel_amd.engine")


@pytest.fixture(scope="module")
def models_mod(pkg):
    return importlib.import_module("muzero-hypermodel_amd.models")


def games(name):
    return importlib.import_module(f"muzero-hypermodel_amd.games.{name}")


def assert_exact(got, want, keys=EXACT_KEYS, where=""):
    for key in keys:
        assert np.array_equal(got[key], want[key]), f"{where}{key} differs"


# ---- injected mode: bit-exact ---------------------------------------------------------------------
@pytest.mark.parametrize("name", TRACE_FILES)
def test_injected_traces_bit_exact_vs_reference_and_oracle(eng, oracle, name):
    fx = load_golden(name)
    T = len(fx["seed"])
    idx = list(range(T))
    temps = fx["temperature"].tolist()
    got = run_injected_on_engine(eng, None, fx, idx, temperature=temps)
    want = run_injected_on_oracle(oracle, fx, idx=idx, temperature=temps)
    assert_exact(got, want, where=f"{name} vs oracle: ")
    # and directly against what the reference recorded
    A, S = int(fx["cfg_A"]), int(fx["cfg_S"])
    for key in ("noise", "visits", "child_value_sum", "child_prior", "child_reward"):
        assert np.array_equal(got[key], fx[key]), key
    assert np.array_equal(got["root_value_sum"], fx["root_value_sum"])
    assert np.array_equal(got["max_tree_depth"], fx["max_tree_depth"])
    assert np.array_equal(got["min_max"][:, 0], fx["mms_min"]) and np.array_equal(got["min_max"][:, 1], fx["mms_max"])
    assert np.array_equal(got["sim_depth"], fx["sim_depth"])
    assert np.array_equal(got["sim_actions"], fx["sim_actions"][:, :, :S])
    assert np.array_equal(got["sim_ties"], fx["sim_ties"][:, :, :S])
    assert np.array_equal(got["child_visits_target"], fx["child_visits_target"])
    assert np.array_equal(got["root_value_target"], fx["root_value_target"])
    assert np.array_equal(got["action"], fx["action_T"])
    # RNG stream accounting: Dirichlet words (host) + tie-break words (device) == reference's count
    dirichlet_words = want["rng_words_run"] - got["tie_break_words"]
    assert (dirichlet_words > 0).all()
    assert np.array_equal(got["depth_sum"], fx["sim_depth"].sum(axis=1))


@pytest.mark.parametrize("name", TRACE_FILES)
def test_injected_traces_fused_step(eng, oracle, name):
    """mzmcts_expand_backup_select (expand + backup of a simulation and the descent of the next in one launch): the same
    bits as the three-step loop and the reference's traces, paths and tie lists included."""
    fx = load_golden(name)
    idx = list(range(len(fx["seed"])))
    temps = fx["temperature"].tolist()
    got = run_injected_on_engine(eng, None, fx, idx, temperature=temps, fused_step=True)
    want = run_injected_on_oracle(oracle, fx, idx=idx, temperature=temps)
    assert_exact(got, want, where=f"{name} (fused step) vs oracle: ")
    S = int(fx["cfg_S"])
    for key in ("noise", "visits", "child_value_sum", "child_prior", "child_reward"):
        assert np.array_equal(got[key], fx[key]), key
    assert np.array_equal(got["sim_actions"], fx["sim_actions"][:, :, :S])
    assert np.array_equal(got["sim_ties"], fx["sim_ties"][:, :, :S])
    assert np.array_equal(got["depth_sum"], fx["sim_depth"].sum(axis=1))


@pytest.mark.parametrize("name", ["g4_cartpole_traces", "g5_cartpole_ties_traces"])
def test_injected_traces_one_lane_per_tree(eng, oracle, name):
    """group_width = 1 (two-action games): one lane owns a tree and loops over its children -- twice the trees per
    wavefront for the HBM-scale lock-step kernels.  Same bits as the reference's traces."""
    fx = load_golden(name)
    idx = list(range(len(fx["seed"])))
    temps = fx["temperature"].tolist()
    got = run_injected_on_engine(eng, None, fx, idx, temperature=temps, group_width=1)
    want = run_injected_on_oracle(oracle, fx, idx=idx, temperature=temps)
    assert_exact(got, want, where=f"{name} (one lane per tree) vs oracle: ")
    for key in ("noise", "visits", "child_value_sum", "child_prior", "child_reward"):
        assert np.array_equal(got[key], fx[key]), key
    assert np.array_equal(got["sim_actions"], fx["sim_actions"][:, :, : int(fx["cfg_S"])])


@pytest.mark.parametrize("name", TRACE_FILES)
def test_injected_traces_two_children_per_lane(eng, oracle, name):
    """group_width = pow2(A) / 2 (4 or 8 lanes, two children per lane) for action counts just above a power of two
    (TicTacToe: 9 actions on 8 lanes instead of 16): half the wavefronts in the tree kernels.  Every trace tiled 3x
    (ragged wavefronts): the same bits as the reference's traces -- paths, tie lists, RNG words."""
    fx = load_golden(name)
    A = len(fixture_config(fx, None).action_space)
    full = 1
    while full < A:
        full *= 2
    if full // 2 not in (4, 8) or A <= full // 2:
        pytest.skip(f"{A} actions: no halved lane group")
    idx = list(range(len(fx["seed"])))
    temps = fx["temperature"].tolist()
    got = run_injected_on_engine(eng, None, fx, idx, temperature=temps, repeat=3, group_width=full // 2)
    want = run_injected_on_oracle(oracle, fx, idx=idx, temperature=temps)
    assert got["all_equal"]
    assert_exact(got, want, where=f"{name} (two children per lane) vs oracle: ")
    for key in ("noise", "visits", "child_value_sum", "child_prior", "child_reward"):
        assert np.array_equal(got[key], fx[key]), key
    assert np.array_equal(got["sim_actions"], fx["sim_actions"][:, :, : int(fx["cfg_S"])])
    assert np.array_equal(got["sim_ties"], fx["sim_ties"][:, :, : int(fx["cfg_S"])])


@pytest.mark.parametrize("name", TRACE_FILES)
@pytest.mark.parametrize("queue", [40, 256])
def test_injected_traces_select_queue(eng, oracle, name, queue):
    """select with a wavefront-local queue of trees (mzmcts_set_select_queue): lane groups whose descent ended pick
    up the next tree of their wavefront.  Every trace tiled 5x, `queue` trees per wavefront (40: a
    ragged refill and a ragged last wavefront; 256: several refills per lane group): the same bits as the
    reference's traces -- paths, tie lists and RNG words included -- and every copy identical."""
    fx = load_golden(name)
    idx = list(range(len(fx["seed"])))
    temps = fx["temperature"].tolist()
    got = run_injected_on_engine(eng, None, fx, idx, temperature=temps, repeat=5, select_queue=queue)
    want = run_injected_on_oracle(oracle, fx, idx=idx, temperature=temps)
    assert got["all_equal"]
    assert_exact(got, want, where=f"{name} (select queue {queue}) vs oracle: ")
    S = int(fx["cfg_S"])
    for key in ("noise", "visits", "child_value_sum", "child_prior", "child_reward"):
        assert np.array_equal(got[key], fx[key]), key
    assert np.array_equal(got["sim_depth"], fx["sim_depth"])
    assert np.array_equal(got["sim_actions"], fx["sim_actions"][:, :, :S])
    assert np.array_equal(got["sim_ties"], fx["sim_ties"][:, :, :S])
    assert np.array_equal(got["depth_sum"], fx["sim_depth"].sum(axis=1))


def test_injected_batching_independence(eng):
    """4096 trees = the 32 CartPole traces tiled 128x: every copy must be identical to the first."""
    fx = load_golden("g4_cartpole_traces")
    got = run_injected_on_engine(eng, None, fx, list(range(32)), record_paths=False, repeat=128)
    assert got["all_equal"]
    assert np.array_equal(got["visits"], fx["visits"])
    assert np.array_equal(got["root_value_sum"], fx["root_value_sum"])


@pytest.mark.parametrize("A,S,players,discount,ties", [
    (1, 12, 1, 0.997, False), (2, 50, 1, 0.997, True), (3, 40, 2, 1, False), (4, 50, 1, 0.997, False),
    (7, 200, 2, 1, False), (9, 25, 2, 1, True), (16, 30, 1, 0.9, False), (33, 40, 2, 0.95, True),
    (64, 40, 1, 0.997, False), (65, 30, 2, 1, False), (121, 60, 2, 1, True), (256, 20, 1, 0.997, False)])
def test_injected_random_streams_vs_oracle(eng, oracle, A, S, players, discount, ties):
    """Configurations no golden trace covers (all lane-group widths, the A > 64 chunked path, long
    searches, exact-tie storms): engine vs oracle, bit-exact."""
    cfg = make_search_config(A, S, players, discount)
    T = 24
    streams = random_streams(T, A, S, seed=1000 + A, n_players=players, ties=ties)
    temps = [[1.0, 0.5, 0.25, 0.0][t % 4] for t in range(T)]
    got = run_injected_on_engine(eng, cfg, streams, temperature=temps)
    want = run_injected_on_oracle(oracle, cfg, streams, temperature=temps)
    assert_exact(got, want, where=f"A={A} S={S}: ")
    if ties:
        assert got["tie_break_words"].sum() > T       # the device RNG really was exercised


def test_mt19937_twist_on_device(eng, oracle):
    """All-equal priors force a tie-break at every level: > 624 words per search, so the device
    generator regenerates its state mid-search; the host mirror must stay in step for the next move."""
    A, S, T = 2, 400, 6
    cfg = make_search_config(A, S, 1, 1.0)
    streams = random_streams(T, A, S, seed=5, n_players=1)
    streams["priors"][:] = 0.5
    streams["root_priors"][:] = 0.0
    for t in range(T):
        streams["legal"][t] = [0, 1]
        streams["root_priors"][t] = [0.5, 0.5]
    streams["value"][:] = 0.0
    streams["reward"][:] = 0.0
    engine = eng.BatchedMCTS(cfg, T, seeds=streams["seeds"])
    got1 = run_injected_on_engine(eng, cfg, streams, engine=engine, record_paths=False)
    assert (got1["tie_break_words"] > 624).all()
    engine.close()
    want1 = run_injected_on_oracle(oracle, cfg, streams)
    assert_exact(got1, want1, keys=["noise", "visits", "root_value_sum", "action", "max_tree_depth"])


def test_two_consecutive_moves_keep_rng_in_step(eng, oracle):
    """Move 2 starts from the RNG state move 1 left behind (Dirichlet + tie-breaks + action sample)."""
    fx = load_golden("g5_cartpole_ties_traces")
    idx = list(range(8))
    streams = streams_from_fixture(fx, idx)
    cfg = fixture_config(fx)
    engine = eng.BatchedMCTS(cfg, len(idx), seeds=streams["seeds"])
    S = cfg.num_simulations

    def one_move():
        engine.begin_search(streams["legal"], streams["to_play"], True)
        noise = engine.noise.copy()
        engine.expand_roots_injected(streams["root_reward"], streams["root_priors"])
        for s in range(S):
            engine.select(gather=False)
            engine.expand_backup_injected(streams["value"][:, s], streams["reward"][:, s], streams["priors"][:, s])
        st = {k: v.copy() for k, v in engine.readout().items()}
        actions, _ = engine.sample_actions(1.0)
        return noise, st, actions

    moves = [one_move(), one_move()]
    engine.close()
    ocfg = oracle.config_from_fixture(fx)
    for t in range(len(idx)):
        rng = oracle.Rng(streams["seeds"][t])
        for noise, st, actions in moves:
            tree = oracle.Tree(ocfg)
            n = len(streams["legal"][t])
            want_noise = tree.reset(rng, streams["legal"][t], streams["to_play"][t], float(streams["root_reward"][t]),
                                    root_priors=streams["root_priors"][t][:n])
            tree.simulate(rng, value=streams["value"][t], reward=streams["reward"][t], priors=streams["priors"][t])
            ost = tree.root_stats()
            assert np.array_equal(noise[t, :n], want_noise[:n])
            assert np.array_equal(st["visits"][t, :n], ost["visits"])
            assert st["root_value_sum"][t] == ost["root_value_sum"]
            slot = oracle.select_action(rng, ost["visits"], 1.0)
            assert actions[t] == streams["legal"][t][slot]


def test_inactive_envs_are_left_alone(eng, oracle):
    fx = load_golden("g5_tictactoe_traces")
    idx = list(range(12))
    streams = streams_from_fixture(fx, idx)
    cfg = fixture_config(fx)
    off = {1, 5, 6}
    for t in off:
        streams["legal"][t] = []
    engine = eng.BatchedMCTS(cfg, len(idx), seeds=streams["seeds"])
    got = run_injected_on_engine(eng, cfg, streams, engine=engine, temperature=fx["temperature"][:12].tolist())
    states_off = {t: engine.get_rng_state(t) for t in off}
    engine.close()
    for t in range(len(idx)):
        if t in off:
            assert got["visits"][t].sum() == 0 and got["action"][t] == -1 and got["root_visits"][t] == 0
            fresh = oracle.Rng(streams["seeds"][t]).get_numpy_state()
            assert np.array_equal(states_off[t][1], fresh[1]) and states_off[t][2] == fresh[2]
        else:
            assert np.array_equal(got["visits"][t], fx["visits"][t])
            assert got["action"][t] == fx["action_T"][t]


def test_plugin_contract_errors(eng):
    cfg = make_search_config(9, 5, 2, 1)
    engine = eng.BatchedMCTS(cfg, 2)
    with pytest.raises(AssertionError, match="subset of the action space"):
        engine.begin_search([[0, 9], [1]], [0, 0], True)
    with pytest.raises(AssertionError, match="subset of the action space"):
        engine.begin_search([list(range(10)), [1]], [0, 0], True)
    with pytest.raises(RuntimeError, match="before begin_search|before expand_roots"):
        engine.select()
    engine.close()
    with pytest.raises(NotImplementedError, match="More than two player"):
        eng.BatchedMCTS(make_search_config(2, 5, 3, 1), 2)


# ---- device decode kernels vs the reference's torch outputs -----------------------------------------
def test_support_to_scalar_and_softmax_kernels(eng):
    fx = load_golden("g1_support_to_scalar")
    E = 64
    cfg = make_search_config(2, 3, 1, 0.997, support=10)
    engine = eng.BatchedMCTS(cfg, E)
    logits = torch.from_numpy(fx["logits21"]).cuda()
    policy = torch.from_numpy(np.random.RandomState(0).standard_normal((E, 2)).astype(np.float32)).cuda()
    engine.begin_search([[0, 1]] * E, [0] * E, False)
    engine.expand_roots(logits, None, policy, torch.zeros(E, 4, device="cuda"))
    st = engine.readout()
    np.testing.assert_allclose(st["root_predicted_value"], fx["out21"][:, 0].astype(np.float64), rtol=1e-5, atol=1e-5)
    want_priors = torch.softmax(policy.cpu(), dim=1).numpy().astype(np.float64)
    np.testing.assert_allclose(st["child_prior"], want_priors, rtol=0, atol=1e-6)
    engine.close()
    # F = 601 (atari.py support 300)
    cfg = make_search_config(4, 3, 1, 0.997, support=300)
    engine = eng.BatchedMCTS(cfg, 8)
    engine.begin_search([[0, 1, 2, 3]] * 8, [0] * 8, False)
    engine.expand_roots(torch.from_numpy(fx["logits601"]).cuda(), None, torch.zeros(8, 4, device="cuda"),
                        torch.zeros(8, 4, device="cuda"))
    st = engine.readout()
    np.testing.assert_allclose(st["root_predicted_value"], fx["out601"][:, 0].astype(np.float64), rtol=1e-4, atol=2e-4)
    engine.close()


# ---- native mode: PyTorch-ROCm inference + HIP tree kernels -------------------------------------------
PARITY_REPORT = {}


def _write_parity_report():
    """Worst deviations per config, for profiles/ (gpurun_out/ is merged back from the GPU box)."""
    import json
    import os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_report.json"), "w") as f:
        json.dump(PARITY_REPORT, f, indent=1, sort_keys=True)


def native_vs_fixture(eng, model, config, fx, idx, expected_divergent=(), logit_tol=1e-5, name=None):
    """Native-mode search (PyTorch-ROCm / HIP inference + HIP tree kernels) of recorded reference traces.

    Every trace must walk the reference's paths simulation for simulation -- except the traces listed in
    `expected_divergent`, whose fp32 network outputs flip a UCB near-tie on this hardware (measured, MI355X);
    a trace outside that set that diverges is a regression.  On identical paths every integer statistic, the
    policy target and the sampled action are exact.

    Floating point, as a chain of derived bounds (north star: logits / targets "within 1e-5"):
      1. a network evaluation on the reference's own input gives value / reward / policy LOGITS within `logit_tol` = 1e-5
         of the reference's: fixtures G2 / G3 (test_gpu_network_outputs_vs_reference_fixtures), and here the root's and
         every depth-1 leaf's.  Deeper leaves inherit the drift of their chain of hidden states: the dynamics network
         and the per-plane min-max rescale amplify an input difference (for these synthetic weights even this package's
         torch-CPU modules replaying the reference's paths are 3e-5 off by depth 4, tests/test_chain_sensitivity.py);
         that drift is measured, reported per depth and carried through steps 2-4 as it is;
      2. the categorical means then differ by at most 2 s |dl| (parity_helpers.categorical_mean_bound; asserted on the
         measured means);
      3. a decoded leaf value / reward differs by at most kappa(x) |dx| + the granularity of the reference's own float32
         transform (parity_helpers.value_transform_bound: 1.2e-4 sqrt(|v| + 1) per step) -- asserted per leaf;
      4. the root value target is an average of discounted sums of those leaves: it differs by at most the worst leaf
         bound plus depth x the worst reward bound -- asserted per trace.
    So the 8.5e-5 seen on TicTacToe value targets IS the logit bar (in fact a 1e-6 logit deviation) seen through the
    transform: less than one lattice step of the reference's float32 arithmetic."""
    from parity_helpers import categorical_mean, categorical_mean_bound, value_transform_bound
    T = len(idx)
    engine = eng.BatchedMCTS(config, T, seeds=[int(fx["seed"][i]) for i in idx])
    engine.set_debug_ties(True)
    obs = np.stack([fx["obs"][i] for i in idx])
    legal = [fx["legal"][i][: int(fx["n_legal"][i])].tolist() for i in idx]
    to_play = [int(fx["to_play"][i]) for i in idx]
    S, support = config.num_simulations, int(config.support_size)
    paths = np.full((T, S, S), -1, np.int32)
    leaf_logits = []
    with torch.no_grad():
        value, reward, policy, hidden = model.initial_inference(torch.from_numpy(obs).cuda())
        want_logits = np.stack([fx["root_policy_logits"][i] for i in idx])
        worst_logit = float(np.abs(policy.cpu().numpy() - want_logits).max())
        np.testing.assert_allclose(policy.cpu().numpy(), want_logits, rtol=0, atol=logit_tol)
        root_value_logits = value.cpu().numpy()
        np.testing.assert_allclose(root_value_logits, np.stack([fx["root_value_logits"][i] for i in idx]), rtol=0, atol=logit_tol)
        engine.begin_search(legal, to_play, True)
        engine.expand_roots(value, reward.contiguous(), policy, hidden)
        for s in range(S):
            engine._select_for(model)
            value, reward, policy, _ = engine._infer(model)
            leaf_logits.append((value.clone(), reward.clone(), policy.clone()))
            engine.expand_backup(value, reward, policy, None)
            _, actions, _ = engine.last_paths()
            paths[:, s] = actions
    st = engine.readout()
    cv, rv = engine.search_statistics()
    temps = [float(fx["temperature"][i]) for i in idx]
    actions, _ = engine.sample_actions(temps)
    engine.close()
    got_v = torch.stack([x[0] for x in leaf_logits], dim=1).cpu()             # [T, S, F]
    got_r = torch.stack([x[1] for x in leaf_logits], dim=1).cpu()
    got_p = torch.stack([x[2] for x in leaf_logits], dim=1).cpu().numpy()     # [T, S, A]
    models_mod = importlib.import_module("muzero-hypermodel_amd.models")

    def decode(logits):                                                        # [n, F] float32 -> [n] (torch CPU, reference order)
        return models_mod.support_to_scalar(logits, support)[:, 0].double().numpy()

    by_depth = {}
    single_step = 0.0          # worst logit deviation of an evaluation with no chain behind it (root, depth-1 leaves)
    same, divergent = 0, {}
    worst = dict(target_abs=0.0, target_rel=0.0, target_over_bound=0.0, pred_abs=0.0, pred_over_bound=0.0, leaf_abs=0.0,
                 leaf_over_bound=0.0, leaf_logit=0.0, prior=0.0, bound_at_bar=0.0, bound_at_measured=0.0)
    for t, i in enumerate(idx):
        n = int(fx["n_legal"][i])
        assert st["visits"][t].sum() == S and st["root_visits"][t] == S
        # root prediction: chain steps 2-3 on the root's value logits
        ref_l = fx["root_value_logits"][i][None]
        dx = abs(float(categorical_mean(root_value_logits[t][None], support)[0] - categorical_mean(ref_l, support)[0]))
        dl = float(np.abs(root_value_logits[t] - ref_l[0]).max())
        assert dx <= categorical_mean_bound(dl, support)
        bound = float(value_transform_bound(fx["root_predicted_value"][i], dx))
        dev = abs(st["root_predicted_value"][t] - fx["root_predicted_value"][i])
        worst["pred_abs"], worst["pred_over_bound"] = max(worst["pred_abs"], dev), max(worst["pred_over_bound"], dev / bound)
        assert dev <= bound, (i, dev, bound)
        assert np.array_equal(engine.noise[t, :n], fx["noise"][i][:n])          # host RNG: exact
        if not np.array_equal(paths[t], fx["sim_actions"][i][:, :S]):
            first = int(np.nonzero((paths[t] != fx["sim_actions"][i][:, :S]).any(axis=1))[0][0])
            divergent[int(i)] = first
            continue
        same += 1
        assert np.array_equal(st["visits"][t], fx["visits"][i])
        assert np.array_equal(cv[t], fx["child_visits_target"][i])              # policy target: exact
        # 1. every leaf's logits
        dl_v = np.abs(got_v[t].numpy() - fx["sim_value_logits"][i]).max(axis=1)           # [S]
        dl_r = np.abs(got_r[t].numpy() - fx["sim_reward_logits"][i]).max(axis=1)
        dl_p = np.abs(got_p[t] - fx["sim_policy_logits"][i]).max(axis=1)
        leaf_logit = float(max(dl_v.max(), dl_r.max(), dl_p.max()))
        worst["leaf_logit"] = max(worst["leaf_logit"], leaf_logit)
        shallow = fx["sim_depth"][i] <= 1            # a leaf one step below the root: no chain behind it yet
        if shallow.any():
            single_step = max(single_step, float(max(dl_v[shallow].max(), dl_r[shallow].max(), dl_p[shallow].max())))
            assert single_step <= logit_tol, i
        for d in range(1, int(fx["sim_depth"][i].max()) + 1):
            at = fx["sim_depth"][i] == d
            if at.any():
                by_depth[d] = max(by_depth.get(d, 0.0), float(max(dl_v[at].max(), dl_r[at].max(), dl_p[at].max())))
        # 2. categorical means, 3. decoded leaves
        leaf_bounds = {}
        for kind, got, ref_logits, ref_scalar, dl_k in (("value", got_v[t], fx["sim_value_logits"][i], fx["sim_value"][i], dl_v),
                                                       ("reward", got_r[t], fx["sim_reward_logits"][i], fx["sim_reward"][i], dl_r)):
            dx = np.abs(categorical_mean(got.numpy(), support) - categorical_mean(ref_logits, support))
            assert (dx <= categorical_mean_bound(dl_k, support) + 1e-12).all()
            bounds = value_transform_bound(ref_scalar, dx)
            devs = np.abs(decode(got) - ref_scalar.astype(np.float64))
            assert (devs <= bounds).all(), (i, kind, float((devs / bounds).max()))
            leaf_bounds[kind] = float(bounds.max())
            if kind == "value":
                worst["leaf_abs"] = max(worst["leaf_abs"], float(devs.max()))
                worst["leaf_over_bound"] = max(worst["leaf_over_bound"], float((devs / bounds).max()))
        # 4. the root value target
        depth = int(fx["sim_depth"][i].max())
        target_bound = leaf_bounds["value"] + depth * leaf_bounds["reward"]
        dev = abs(rv[t] - fx["root_value_target"][i])
        worst["target_abs"] = max(worst["target_abs"], dev)
        worst["target_rel"] = max(worst["target_rel"], dev / max(1.0, abs(fx["root_value_target"][i])))
        worst["target_over_bound"] = max(worst["target_over_bound"], dev / target_bound)
        assert dev <= target_bound, (i, dev, target_bound)
        # what the same chain allows at the north star's logit bar, and at the deviation measured on this trace
        vmax, rmax = float(np.abs(fx["sim_value"][i]).max()), float(np.abs(fx["sim_reward"][i]).max())
        for key, dlv in (("bound_at_bar", logit_tol), ("bound_at_measured", single_step)):
            b = float(value_transform_bound(vmax, categorical_mean_bound(dlv, support))
                      + depth * value_transform_bound(rmax, categorical_mean_bound(dlv, support)))
            worst[key] = max(worst[key], b)
        worst["prior"] = max(worst["prior"], float(np.abs(st["child_prior"][t, :n] - fx["child_prior"][i][:n]).max()))
        np.testing.assert_allclose(st["child_prior"][t, :n], fx["child_prior"][i][:n], rtol=0, atol=logit_tol)
        np.testing.assert_allclose(st["child_value_sum"][t, :n], fx["child_value_sum"][i][:n], rtol=0, atol=target_bound * S)
        assert actions[t] == fx["action_T"][i]
    print(f"identical-path rate {same}/{T}; worst root-value deviation {worst['target_abs']:.2e} abs = "
          f"{worst['target_over_bound']:.2f} of its derived bound; worst logit deviation per evaluation {single_step:.2e}, "
          f"inside the search {worst['leaf_logit']:.2e}; "
          f"divergent traces (trace: first differing simulation) {divergent}")
    if name:
        PARITY_REPORT[name] = dict(
            traces=T, identical_paths=same, divergent_first_simulation=divergent, logit_bar=logit_tol,
            worst_root_policy_logit_abs=worst_logit, worst_depth1_leaf_logit_abs=single_step,
            worst_in_search_leaf_logit_abs=worst["leaf_logit"],
            in_search_leaf_logit_abs_by_depth={str(d): v for d, v in sorted(by_depth.items())}, worst_prior_abs=worst["prior"],
            worst_root_predicted_value_abs=worst["pred_abs"], worst_root_predicted_value_over_derived_bound=worst["pred_over_bound"],
            worst_leaf_value_abs=worst["leaf_abs"], worst_leaf_value_over_derived_bound=worst["leaf_over_bound"],
            worst_value_target_abs=worst["target_abs"], worst_value_target_rel=worst["target_rel"],
            worst_value_target_over_derived_bound=worst["target_over_bound"],
            value_target_bound_from_measured_depth1_logit_deviation=worst["bound_at_measured"],
            value_target_bound_from_the_1e5_logit_bar=worst["bound_at_bar"])
        _write_parity_report()
    unexpected = sorted(set(divergent) - set(expected_divergent))
    assert not unexpected, f"traces {unexpected} left the reference's paths (identical-path rate {same}/{T})"
    return same / T


# Bars per config (BASELINE.md "parity bars"): every network logit of a recorded reference search within the north star's
# 1e-5 (worst measured 1.5e-6); decoded values, rewards and value targets within the bound DERIVED from the logit deviation
# through the reference's float32 transform (native_vs_fixture steps 2-4; round 2 held them to an empirical 1.5e-4).
# The play_game fixtures (G6) hold no logits, so their root values get the same chain evaluated AT the logit bar for the
# largest decoded magnitudes these synthetic networks produce (|v|, |r| <= 4, paths of <= 9 moves): 1.4e-2 absolute --
# rigorous and loose; measured deviations are printed by the tests (2e-5) and the traces above carry the tight statement.
def _g6_value_bound():
    from parity_helpers import categorical_mean_bound, value_transform_bound
    return float(value_transform_bound(4.0, categorical_mean_bound(1e-5, 10)) * (1 + 9))


RESNET_TOL = dict(value_tol=_g6_value_bound(), logit_tol=1e-5)
# traces whose search leaves the reference's path on MI355X because an fp32-rounding-sized difference of the
# network outputs flips a UCB near-tie (trace index: see the report); everything else must match move for move
# connect4 (default path: the split-precision tower kernel): traces 23 and 29 leave the reference's path at simulation
# 167 / 183 of 200, 28/30 identical.  Which near-ties flip depends on the fp32 rounding of the network path: MIOpen's
# convolution flips 21 and 23, the exact-fp32 MFMA kernel 9, 21 and 23 (MZ_BOARD_CONV_PRECISION=fp32).
EXPECTED_DIVERGENT = {"cartpole": (), "tictactoe": (), "connect4": (23, 29), "atari84": ()}


def test_native_cartpole_vs_reference(eng, models_mod):
    config = games("cartpole").MuZeroConfig()
    model, _ = cartpole_model_and_weights(models_mod, config, "cuda")
    fx = load_golden("g4_cartpole_traces")
    native_vs_fixture(eng, model, config, fx, list(range(len(fx["seed"]))), EXPECTED_DIVERGENT["cartpole"],
                      name="cartpole")


def test_native_tictactoe_vs_reference(eng, models_mod):
    config = games("tictactoe").MuZeroConfig()
    model, _ = synthetic_model(models_mod, config, "cuda")
    fx = load_golden("g5_tictactoe_traces")
    native_vs_fixture(eng, model, config, fx, list(range(len(fx["seed"]))), EXPECTED_DIVERGENT["tictactoe"],
                      name="tictactoe", logit_tol=RESNET_TOL["logit_tol"])


def test_native_connect4_vs_reference(eng, models_mod):
    config = games("connect4").MuZeroConfig()
    model, _ = synthetic_model(models_mod, config, "cuda")
    fx = load_golden("g5_connect4_traces")
    native_vs_fixture(eng, model, config, fx, list(range(len(fx["seed"]))), EXPECTED_DIVERGENT["connect4"],
                      name="connect4", logit_tol=RESNET_TOL["logit_tol"])


def atari84_traces():
    fx = dict(load_golden("g5_atari84_traces"))
    fx["obs"] = fx["obs_u8"].astype(np.float32) / np.float32(255)      # 8-bit frames, as recorded
    return fx


def test_native_atari84_vs_reference(eng, models_mod):
    """BASELINE config #5 end to end: DownsampleCNN representation of 4 x 84 x 84 frames, A = 4, 50 simulations."""
    config = games("breakout").atari84_config()
    model, _ = synthetic_model(models_mod, config, "cuda")
    fx = atari84_traces()
    native_vs_fixture(eng, model, config, fx, list(range(len(fx["seed"]))), EXPECTED_DIVERGENT["atari84"],
                      name="atari84", logit_tol=RESNET_TOL["logit_tol"])


def test_gpu_network_outputs_vs_reference_fixtures(models_mod):
    for name, loader in (("cartpole", "fc"), ("tictactoe", "res"), ("connect4", "res"), ("atari84", "res")):
        if name == "atari84":
            config = games("breakout").atari84_config()
        else:
            config = games(name).MuZeroConfig()
        if loader == "fc":
            model, _ = cartpole_model_and_weights(models_mod, config, "cuda")
            fx = load_golden("g2_fc_inference")
        else:
            model, _ = synthetic_model(models_mod, config, "cuda")
            fx = load_golden(f"g3_{name}_inference")
        with torch.no_grad():
            v0, r0, p0, h0 = model.initial_inference(torch.from_numpy(fx["obs"]).cuda())
            v1, r1, p1, h1 = model.recurrent_inference(torch.from_numpy(fx["init_hidden"]).cuda(),
                                                       torch.from_numpy(fx["actions"]).cuda())
        for got, key in ((v0, "init_value"), (p0, "init_policy"), (h0, "init_hidden"), (v1, "rec_value"),
                         (r1, "rec_reward"), (p1, "rec_policy"), (h1, "rec_hidden")):
            # the north star's bar on every network output: logits within 1e-5 of the reference's (measured worst over the
            # four networks 1.4e-6).  Hidden states are (x - min) / (max - min) per board plane: an error e in x, min and
            # max shows as up to 3 e / span, so they get 2e-5 (measured worst 7.3e-6, TicTacToe's recurrent state)
            tol = 2e-5 if key.endswith("hidden") and loader != "fc" else 1e-5
            np.testing.assert_allclose(got.cpu().numpy(), fx[key], rtol=tol, atol=tol, err_msg=f"{name}:{key}")


def test_graph_replay_equals_eager(eng, models_mod):
    """The hipGraph-captured simulation loop must give bit-identical trees to eager launches."""
    config = games("cartpole").MuZeroConfig()
    model, _ = cartpole_model_and_weights(models_mod, config, "cuda")
    E = 256
    rs = np.random.RandomState(3)
    obs = [rs.uniform(-0.05, 0.05, (E, 1, 1, 4)).astype(np.float32) for _ in range(4)]
    results = {}
    for mode in (False, True):
        engine = eng.BatchedMCTS(config, E, use_graph=mode)
        out = []
        for o in obs:
            st = engine.search(model, o, [[0, 1]] * E, [0] * E, True)
            acts, _ = engine.sample_actions(1.0)
            out.append((st["visits"].copy(), st["root_value_sum"].copy(), st["child_value_sum"].copy(), acts.copy()))
        assert (engine._graph is not None) == mode
        results[mode] = out
        engine.close()
    for a, b in zip(results[False], results[True]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_full_size_invariants_cartpole_4096(eng, models_mod):
    """BASELINE config #2 (4096 envs x 50 sims): size-independent properties + run-to-run determinism."""
    config = games("cartpole").MuZeroConfig()
    model, _ = cartpole_model_and_weights(models_mod, config, "cuda")
    E, S = 4096, config.num_simulations
    obs = np.random.RandomState(123).uniform(-0.05, 0.05, (E, 1, 1, 4)).astype(np.float32)
    runs = []
    for _ in range(2):
        engine = eng.BatchedMCTS(config, E, use_graph=False)
        st = {k: v.copy() for k, v in engine.search(model, obs, [[0, 1]] * E, [0] * E, True).items()}
        cv, rv = engine.search_statistics()
        tree = engine.export_tree(17)
        engine.close()
        runs.append((st, cv, rv, tree))
    st, cv, rv, tree = runs[0]
    assert (st["visits"].sum(axis=1) == S).all() and (st["root_visits"] == S).all()
    assert (st["max_tree_depth"] >= 1).all() and (st["max_tree_depth"] <= S).all()
    assert (st["depth_sum"] >= S).all()
    np.testing.assert_allclose(cv.sum(axis=1), 1.0, rtol=0, atol=1e-12)
    assert (st["min_max"][:, 0] <= st["min_max"][:, 1]).all()
    np.testing.assert_allclose(st["child_prior"].sum(axis=1), 1.0, rtol=0, atol=1e-6)
    # whole-tree consistency of one env: every expanded node's visits = 1 + sum of its children's
    visits, child = tree["visits"], tree["child_node"]
    for k in range(S + 1):
        for a in range(2):
            ck = child[k, a]
            if ck >= 0:
                assert visits[k, a] == 1 + visits[ck].sum()
    assert sorted(int(c) for c in child.reshape(-1) if c >= 0) == list(range(1, S + 1))
    for (a_st, a_cv, a_rv, _), (b_st, b_cv, b_rv, _) in [(runs[0], runs[1])]:
        for key in a_st:
            assert np.array_equal(a_st[key], b_st[key]), key
        assert np.array_equal(a_rv, b_rv)


# ---- the drop-in facade -------------------------------------------------------------------------------
def test_mcts_run_facade_matches_reference_trace(eng, models_mod, pkg):
    sp = importlib.import_module("muzero-hypermodel_amd.self_play")
    config = games("cartpole").MuZeroConfig()
    model, _ = cartpole_model_and_weights(models_mod, config, "cuda")
    fx = load_golden("g4_cartpole_traces")
    mcts = sp.MCTS(config)
    hits = 0
    for i in range(8):
        np.random.seed(int(fx["seed"][i]))
        root, info = mcts.run(model, fx["obs"][i], [0, 1], 0, True)
        assert set(info) == {"max_tree_depth", "root_predicted_value"}
        assert root.visit_count == config.num_simulations and root.hidden_state.shape == (1, 8)
        assert sum(c.visit_count for c in root.children.values()) == config.num_simulations
        assert abs(info["root_predicted_value"] - fx["root_predicted_value"][i]) <= 3e-5 * abs(fx["root_predicted_value"][i])
        action = sp.SelfPlay.select_action(root, float(fx["temperature"][i]))
        gh = sp.GameHistory()
        gh.store_search_statistics(root, config.action_space)
        if [c.visit_count for c in root.children.values()] == fx["visits"][i].tolist():
            hits += 1
            assert info["max_tree_depth"] == fx["max_tree_depth"][i]
            assert [c.prior for c in root.children.values()] == pytest.approx(fx["child_prior"][i].tolist(), abs=1e-6)
            assert gh.child_visits[0] == fx["child_visits_target"][i].tolist()
            assert abs(gh.root_values[0] - fx["root_value_target"][i]) <= 3e-5 * abs(fx["root_value_target"][i])
            assert action == fx["action_T"][i]
        # deep structure: expanded children carry hidden states and children of their own
        for child in root.children.values():
            if child.visit_count > 0:
                assert child.expanded() and child.hidden_state is not None and child.to_play == 0
    mcts.close()
    assert hits == 8, f"MCTS.run facade: {hits}/8 recorded searches reproduced (measured on MI355X: 8/8)"
    with pytest.raises(AssertionError, match="should not be an empty array"):
        sp.MCTS(config).run(model, fx["obs"][0], [], 0, True)


def test_mcts_run_override_root_with(eng, models_mod, pkg):
    """MCTS.run(..., override_root_with=root) (self_play.py:276-278, diagnose_model.py:97-123): a root expanded by hand
    from the network's own outputs must be searched exactly like the root MCTS.run builds itself."""
    sp = importlib.import_module("muzero-hypermodel_amd.self_play")
    ttt = games("tictactoe")
    config = ttt.MuZeroConfig()
    model, _ = synthetic_model(models_mod, config, "cuda")
    fx = load_golden("g5_tictactoe_traces")
    mcts = sp.MCTS(config)
    for i in (0, 3, 7):
        legal = fx["legal"][i][: int(fx["n_legal"][i])].tolist()
        to_play = int(fx["to_play"][i])
        for noise in (False, True):
            np.random.seed(int(fx["seed"][i]))
            plain, info = mcts.run(model, fx["obs"][i], legal, to_play, noise)
            with torch.no_grad():
                v, r, p, h = model.initial_inference(torch.from_numpy(fx["obs"][i]).float().unsqueeze(0).cuda())
            root = sp.Node(0)
            root.expand(legal, to_play, models_mod.support_to_scalar(r, config.support_size).item(), p, h)
            np.random.seed(int(fx["seed"][i]))
            given, info2 = mcts.run(model, None, None, None, noise, override_root_with=root)
            assert info2["root_predicted_value"] is None and info2["max_tree_depth"] == info["max_tree_depth"]
            assert list(given.children) == legal and given.to_play == to_play and given.visit_count == plain.visit_count
            for a in legal:
                assert given.children[a].visit_count == plain.children[a].visit_count
                # (the hand-built root's priors are torch's CPU softmax, the engine's its own fp32 softmax: a couple of ulps apart)
                assert given.children[a].prior == pytest.approx(plain.children[a].prior, abs=2e-7)
                assert given.children[a].value_sum == pytest.approx(plain.children[a].value_sum, rel=1e-6, abs=1e-6)
            if noise:                                      # the recorded reference search, when paths agree
                assert [given.children[a].visit_count for a in legal] == fx["visits"][i][: len(legal)].tolist()
    searched = plain
    with pytest.raises(NotImplementedError, match="freshly expanded"):
        mcts.run(model, None, None, None, False, override_root_with=searched)
    mcts.close()


# Searches of recorded games whose visit counts may leave the reference's by ONE simulation (a UCB near-tie decided the
# other way by the last bits of the network's outputs), the sampled action staying the reference's: (fixture, tower
# precision) -> {run: [move indices]}.  Measured on MI355X; with the exact-fp32 towers every game is exact.
G6_NEAR_TIES = {("g6_connect4_opponents_games", "split"): {2: [5]}}


@pytest.mark.parametrize("fixture,game,precision", [
    ("g6_tictactoe_games", "tictactoe", "split"), ("g6_connect4_games", "connect4", "split"),
    ("g6_connect4_games", "connect4", "fp32"), ("g6_connect4_opponents_games", "connect4", "split"),
    ("g6_connect4_opponents_games", "connect4", "fp32")])
def test_self_play_games_vs_reference_g6(eng, models_mod, pkg, monkeypatch, fixture, game, precision):
    """SelfPlay.play_game (self_play.py:110-184) with the synthetic weights against trajectories recorded from the
    reference: TicTacToe (seeds 0-3 self-play, expert / random opponents, temperature threshold), Connect4 self-play
    (200 simulations per move through the split-precision tower) and Connect4 test-mode games -- the expert of
    games/connect4.py:306-343 as either player, a random opponent (select_opponent_action, self_play.py:189-221), and
    the temperature threshold.  Every game must be reproduced move for move, with the policy targets bit for bit, the
    value targets within the residual networks' bar, and the global RNG left where the reference left it.  Connect4's
    64-channel towers run in both forms: `split` (two fp16 halves per operand, the default) and `fp32` (exact-fp32 MFMA);
    the searches listed in G6_NEAR_TIES may move one simulation between two children."""
    monkeypatch.setenv("MZ_BOARD_CONV_PRECISION", precision)
    near_ties = G6_NEAR_TIES.get((fixture, precision), {})
    sp = importlib.import_module("muzero-hypermodel_amd.self_play")
    mod = games(game)
    config = mod.MuZeroConfig()
    A = len(config.action_space)
    _, weights = synthetic_model(models_mod, config, "cpu")
    fx = load_golden(fixture)
    full, total, worst_rv = 0, int(fx["n_runs"]), 0.0
    for i in range(total):
        seed, temp, thr, opp, mzp = fx[f"run{i}_args"]
        opponent = {0: "self", 1: "expert", 2: "random"}[int(opp)]
        actor = sp.SelfPlay({"weights": weights}, mod.Game, config, int(seed))
        gh = actor.play_game(float(temp), None if thr < 0 else int(thr), False, opponent, int(mzp))
        actor.close_game()
        ref_actions = fx[f"run{i}_actions"].tolist()
        # the first move depends only on the root search: it must agree
        assert gh.action_history[0] == 0 and gh.to_play_history[0] == 0
        n = min(len(ref_actions), len(gh.action_history))
        agree = 0
        while agree < n and gh.action_history[agree] == ref_actions[agree]:
            agree += 1
        assert agree >= 2, f"run {i}: diverged at the very first move"
        if agree == len(ref_actions) == len(gh.action_history):
            full += 1
            assert gh.reward_history == fx[f"run{i}_rewards"].tolist()
            assert gh.to_play_history == fx[f"run{i}_to_play"].tolist()
            got_cv = np.array(gh.child_visits, dtype=np.float64).reshape(-1, A)
            ref_cv = fx[f"run{i}_child_visits"]
            got_rv = np.array([np.nan if v is None else v for v in gh.root_values])
            ref_rv = fx[f"run{i}_root_values"].copy()
            flipped = [m for m in range(len(got_cv)) if not np.array_equal(got_cv[m], ref_cv[m])]
            assert flipped == [m for m in near_ties.get(i, []) if m in flipped], (i, flipped)     # only listed searches
            searched = np.flatnonzero(~np.isnan(ref_rv))          # (opponent moves store no statistics: root value None)
            for m in flipped:
                # one simulation went to another child: two policy-target entries move by 1 / S, the root value by one
                # leaf evaluation out of S
                delta = np.abs(got_cv[m] - ref_cv[m]) * config.num_simulations
                assert sorted(np.round(delta).tolist())[-3:] == [0.0, 1.0, 1.0] and abs(delta.sum() - 2.0) < 1e-9
                at = searched[m]
                assert abs(got_rv[at] - ref_rv[at]) <= 4.0 / config.num_simulations        # (|leaf values| stay below 2 here)
                got_rv[at] = ref_rv[at]
            np.testing.assert_allclose(got_rv, ref_rv, rtol=0, atol=RESNET_TOL["value_tol"], equal_nan=True)
            worst_rv = max(worst_rv, float(np.nanmax(np.abs(got_rv - ref_rv))))
            assert np.array_equal(np.array(gh.observation_history, dtype=np.float32), fx[f"run{i}_observations"])
            assert int(np.random.randint(0, 2**31 - 1)) == int(fx[f"run{i}_rng_next_word"])
        else:
            print(f"{fixture} run {i} ({opponent}): left the reference's game at move {agree} of {len(ref_actions) - 1}")
    print(f"{fixture} ({precision}): games reproduced move for move: {full}/{total}; worst root-value deviation {worst_rv:.2e}")
    assert full == total, f"only {full}/{total} games reproduced move for move (measured on MI355X: all of them)"


def test_batched_self_play_matches_single_env_actor(eng, models_mod, pkg):
    """E lock-step envs: env e must play exactly the game reference worker `seed + e` plays."""
    sp = importlib.import_module("muzero-hypermodel_amd.self_play")
    ttt = games("tictactoe")
    config = ttt.MuZeroConfig()
    _, weights = synthetic_model(models_mod, config, "cpu")
    E = 4
    finished = {}
    batched = sp.BatchedSelfPlay({"weights": weights}, ttt.Game, config, 0, E, use_graph=False)
    while len(finished) < E:
        batched.step(1.0, None, on_game=lambda e, gh: finished.setdefault(e, gh))
    batched.close()
    fx = load_golden("g6_tictactoe_games")
    for e in range(E):
        single = sp.SelfPlay({"weights": weights}, ttt.Game, config, e)
        gh = single.play_game(1.0, None, False, "self", 0)
        single.close_game()
        assert finished[e].action_history == gh.action_history
        assert finished[e].reward_history == gh.reward_history
        assert np.array_equal(np.array(finished[e].child_visits, dtype=float), np.array(gh.child_visits, dtype=float))
        # network numerics are not batch-size invariant (different MIOpen / GEMM kernels at batch 4 and
        # batch 1), so values agree to the ResNet tolerance while every integer statistic is identical
        np.testing.assert_allclose(finished[e].root_values, gh.root_values, rtol=0, atol=RESNET_TOL["value_tol"])


def test_batched_self_play_cartpole_fused(eng, models_mod, pkg):
    """BatchedSelfPlay on the CartPole plugin: FC network => the fused kernel; envs restart when done."""
    sp = importlib.import_module("muzero-hypermodel_amd.self_play")
    cp = games("cartpole")
    config = cp.MuZeroConfig()
    config.max_moves = 30
    _, weights = cartpole_model_and_weights(models_mod, config, "cpu")
    E = 64
    done = []
    actor = sp.BatchedSelfPlay({"weights": weights}, cp.Game, config, 0, E)
    assert actor.engine._fc_model is actor.model            # fused path configured
    for _ in range(35):
        actor.step(1.0, None, on_game=lambda e, gh: done.append((e, gh)))
    actor.close()
    assert actor.moves_played == 35 * E and len(done) >= E   # every env finished at least one game
    for e, gh in done[:8]:
        n = len(gh.action_history)
        assert 2 <= n <= config.max_moves + 1
        assert len(gh.child_visits) == n - 1 == len(gh.root_values)
        assert all(abs(sum(cv) - 1.0) < 1e-12 for cv in gh.child_visits)
        assert np.asarray(gh.observation_history[0]).shape == config.observation_shape
        assert set(gh.action_history[1:]) <= {0, 1} and all(r == 1.0 for r in gh.reward_history[1:])


@pytest.mark.parametrize("use_graph", [False, True])
def test_pipelined_lockstep_groups_equal_one_engine(eng, models_mod, use_graph):
    """engine.PipelinedLockstep: the envs as two lock-step engines on streams of their own (one group's host work under
    the other's kernels, network replica per group) play exactly what one engine of all envs plays -- noise, visit counts,
    value sums, sampled actions -- over two consecutive moves (TicTacToe residual network, masked roots)."""
    import importlib
    from parity_helpers import synthetic_model
    config = importlib.import_module("muzero-hypermodel_amd.games.tictactoe").MuZeroConfig()
    model, _ = synthetic_model(models_mod, config, "cuda")
    E = 64
    rs = np.random.RandomState(9)
    obs = torch.from_numpy(rs.randint(-1, 2, (E, 3, 3, 3)).astype(np.float32)).cuda()
    legal = np.zeros((E, 9), np.int32)
    num_legal = rs.randint(1, 10, E).astype(np.int32)
    for e in range(E):
        legal[e, :num_legal[e]] = np.sort(rs.permutation(9)[:num_legal[e]])
    to_play = rs.randint(0, 2, E).astype(np.int32)
    seeds = list(range(500, 500 + E))

    single = eng.BatchedMCTS(config, E, seeds=seeds)
    want = []
    for _ in range(4):       # (with use_graph: an eager move, the move that captures the simulation loop, two replays)
        st = single.search(model, obs, legal, to_play, True, num_legal=num_legal)
        actions, _ = single.sample_actions(1.0)
        want.append(({k: v.copy() for k, v in st.items()}, single.noise.copy(), actions))
    single.close()

    pipe = eng.PipelinedLockstep(config, E, model, groups=2, seeds=seeds, use_graph=use_graph)
    got = []
    for _ in range(4):
        for g in range(2):
            sl = pipe.slice(g)
            pipe.begin(g, obs[sl].contiguous(), legal[sl], to_play[sl], True, num_legal=num_legal[sl])
        parts = []
        for g in range(2):
            st = pipe.finish(g)
            actions, _ = pipe.engines[g].sample_actions(1.0)
            parts.append(({k: v.copy() for k, v in st.items()}, pipe.engines[g].noise.copy(), actions))
        got.append(parts)
    pipe.close()
    for (st, noise, actions), parts in zip(want, got):
        assert np.array_equal(noise, np.concatenate([p[1] for p in parts]))
        assert np.array_equal(actions, np.concatenate([p[2] for p in parts]))
        for key in ("visits", "child_value_sum", "child_prior", "root_value_sum", "max_tree_depth", "min_max"):
            assert np.array_equal(st[key], np.concatenate([p[0][key] for p in parts])), key
        assert (st["visits"].sum(axis=1) == config.num_simulations).all()


def test_many_env_actor_plays_expert_and_random_opponents(eng, models_mod, pkg):
    """Test-mode games against an opponent with E envs in lock step (reference self_play.py:65-90, 189-221: MuZero
    searches only on muzero_player's turns; the other moves come from the plugin's expert_agent() or from
    numpy.random.choice on the worker's own stream; no search statistics are stored for them).  With one env the actor
    replays the reference's recorded TicTacToe games (fixture G6 runs 4 and 5: expert opponent / random opponent with
    MuZero as second player); with several envs every env plays the game the single-env facade plays with its seed; and
    ManyEnvLoop.continuous_self_play(test_mode=True) reports the same metrics as SelfPlay.continuous_self_play."""
    sp = importlib.import_module("muzero-hypermodel_amd.self_play")
    ttt = games("tictactoe")
    config = ttt.MuZeroConfig()
    _, weights = synthetic_model(models_mod, config, "cpu")
    fx = load_golden("g6_tictactoe_games")

    def play(seed, E, opponent, mzp, temperature=0):
        finished = {}
        actor = sp.BatchedSelfPlay({"weights": weights}, ttt.Game, config, seed, E, use_graph=False)
        while len(finished) < E:
            actor.step(temperature, None, on_game=lambda e, gh: finished.setdefault(e, gh), opponent=opponent, muzero_player=mzp)
        actor.close()
        return finished

    for run in (4, 5):
        seed, temp, thr, opp, mzp = fx[f"run{run}_args"]
        opponent = {1: "expert", 2: "random"}[int(opp)]
        gh = play(int(seed), 1, opponent, int(mzp), float(temp))[0]
        assert gh.action_history == fx[f"run{run}_actions"].tolist(), run
        assert gh.reward_history == fx[f"run{run}_rewards"].tolist() and gh.to_play_history == fx[f"run{run}_to_play"].tolist()
        assert np.array_equal(np.array(gh.child_visits, dtype=np.float64).reshape(-1, 9), fx[f"run{run}_child_visits"])
        got_rv = np.array([np.nan if v is None else v for v in gh.root_values])
        assert np.array_equal(np.isnan(got_rv), np.isnan(fx[f"run{run}_root_values"]))      # None exactly on opponent moves
        np.testing.assert_allclose(got_rv, fx[f"run{run}_root_values"], rtol=0, atol=RESNET_TOL["value_tol"], equal_nan=True)

    for opponent, mzp in (("expert", 1), ("random", 0)):
        many = play(20, 3, opponent, mzp)
        for e in range(3):
            single = sp.SelfPlay({"weights": weights}, ttt.Game, config, 20 + e)
            gh = single.play_game(0, None, False, opponent, mzp)
            single.close_game()
            assert many[e].action_history == gh.action_history, (opponent, e)
            assert [v is None for v in many[e].root_values] == [v is None for v in gh.root_values]
            assert np.array_equal(np.array(many[e].child_visits, dtype=float), np.array(gh.child_visits, dtype=float))

    class Storage:
        def __init__(self):
            self.info = {"training_step": 0, "terminate": False, "weights": weights}
            self.metrics = []

        def get_info(self, key):
            return self.info[key]

        def set_info(self, keys, values=None):
            self.metrics.append(dict(keys))
            if "muzero_reward" in keys:
                self.info["training_step"] += 1             # (one test game per "training step": ends the loop)

    cfg = ttt.MuZeroConfig()
    cfg.opponent, cfg.muzero_player, cfg.training_steps = "expert", 0, 2
    stores = []
    for kind in ("facade", "many"):
        store = Storage()
        if kind == "facade":
            actor = sp.SelfPlay({"weights": weights}, ttt.Game, cfg, 30)
        else:
            actor = sp.BatchedSelfPlay({"weights": weights}, ttt.Game, cfg, 30, 1, use_graph=False)
        actor.continuous_self_play(store, None, True)
        stores.append(store.metrics)
    assert len(stores[0]) == 4 and {"muzero_reward", "opponent_reward"} <= set(stores[0][1])
    for a, b in zip(*stores):
        assert set(a) == set(b)
        for k in a:
            assert a[k] == pytest.approx(b[k], abs=RESNET_TOL["value_tol"]), k


@pytest.mark.parametrize("game,use_graph", [("tictactoe", False), ("tictactoe", True), ("connect4", True)])
def test_fused_step_with_towers_gathering_from_the_pool(eng, models_mod, monkeypatch, game, use_graph):
    """MZ_FUSED_STEP=on (expand_backup of a simulation and the descent of the next in one launch) on the path where the
    towers read their input from the hidden-state pool: the same noise, visit counts, value sums, bounds, sampled actions
    and tree as the two-launch loop, over three moves (an eager one, the capturing one, a replay)."""
    import importlib
    from parity_helpers import synthetic_model
    config = importlib.import_module(f"muzero-hypermodel_amd.games.{game}").MuZeroConfig()
    if game == "connect4":
        config.num_simulations = 40
    model, _ = synthetic_model(models_mod, config, "cuda")
    A = len(config.action_space)
    E = 150
    rs = np.random.RandomState(21)
    c, h, w = config.observation_shape
    obs = torch.from_numpy(rs.randint(-1, 2, (E, c, h, w)).astype(np.float32)).cuda()
    legal = np.zeros((E, A), np.int32)
    num_legal = rs.randint(1, A + 1, E).astype(np.int32)
    for e in range(E):
        legal[e, :num_legal[e]] = np.sort(rs.permutation(A)[:num_legal[e]])
    to_play = rs.randint(0, 2, E).astype(np.int32)
    seeds = list(range(900, 900 + E))
    runs = {}
    for mode in ("off", "on"):
        monkeypatch.setenv("MZ_FUSED_STEP", mode)
        engine = eng.BatchedMCTS(config, E, seeds=seeds, use_graph=use_graph)
        assert engine.fused_step == (mode == "on") and engine._pool_path(model)
        moves = []
        for _ in range(3):
            st = engine.search(model, obs, legal, to_play, True, num_legal=num_legal)
            actions, _ = engine.sample_actions(1.0)
            moves.append(({k: v.copy() for k, v in st.items()}, engine.noise.copy(), actions.copy(), engine.export_tree(E - 1)))
        engine.close()
        runs[mode] = moves
    for (a_st, a_noise, a_act, a_tree), (b_st, b_noise, b_act, b_tree) in zip(runs["off"], runs["on"]):
        assert np.array_equal(a_noise, b_noise) and np.array_equal(a_act, b_act)
        for key in a_st:
            assert np.array_equal(a_st[key], b_st[key]), key
        for key in a_tree:
            assert np.array_equal(a_tree[key], b_tree[key]), f"tree {key}"
