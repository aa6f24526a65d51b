"""Exploration noise drawn on the GPU (numpy.random.dirichlet, reference self_play.py:468-477): glibc's log / pow on
the device and the legacy gamma sampler on device-resident MT19937 streams, against the host's libm and against
fixture G7 (vectors recorded from numpy itself next to the reference run)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native(pkg):
    import importlib
    return importlib.import_module("muzero-hypermodel_amd._native")


def test_device_log_pow_return_libm_bits(native):
    lib = native.load()
    rs = np.random.RandomState(11)
    n = 1 << 21
    u = rs.random_sample(n)
    kind = np.arange(n) & 3
    x = np.where(kind == 0, 1.0 - u, np.where(kind == 1, u * 4.0 + 1e-300,
                 np.where(kind == 2, np.ldexp(u + 0.5, rs.randint(-1000, 1000, n)), (1.0 - u) / 0.25)))
    shapes = np.array([0.25, 0.1, 0.3, 0.03, 0.15, 0.5, 0.7, 0.9])[(np.arange(n) >> 2) & 7]
    y = 1.0 / shapes
    bases = np.where(kind == 0, u, np.where(kind == 1, 1.0 - shapes + shapes * -np.log(u * shapes),
                     np.where(kind == 2, np.ldexp(u + 0.5, -rs.randint(0, 60, n)), u * 8.0)))
    bases[::99991] = 0.0
    x[::99989] = np.ldexp(u[::99989] + 0.5, -1060)          # subnormal arguments of log
    # log on x, pow on (bases, y); the second output of each call is not looked at
    P = native.c_f64_p
    got_log, got_pow, unused = np.empty(n), np.empty(n), np.empty(n)
    assert lib.mzmcts_device_libm(native.ptr(x, P), native.ptr(y, P), n, native.ptr(got_log, P), native.ptr(unused, P)) == 0
    assert lib.mzmcts_device_libm(native.ptr(bases, P), native.ptr(y, P), n, native.ptr(unused, P), native.ptr(got_pow, P)) == 0
    want_log = np.array([math.log(v) for v in x])            # Python's math module calls libm (numpy's array log may not)
    want_pow = np.array([math.pow(b, e) for b, e in zip(bases, y)])
    assert np.array_equal(got_log.view(np.uint64), want_log.view(np.uint64))
    assert np.array_equal(got_pow.view(np.uint64), want_pow.view(np.uint64))
    assert (want_pow[bases > 0] < 1e-200).sum() > 1000       # the scaled / subnormal tail of pow was exercised


@pytest.mark.parametrize("alpha,k", [(0.25, 2), (0.1, 9), (0.3, 7), (0.25, 4), (1.0, 3), (0.03, 121)])
def test_device_dirichlet_equals_numpy_fixture(native, golden, alpha, k):
    lib = native.load()
    fx = golden("g7_numpy_rng")
    seeds = np.array([0, 1, 12345, 2**32 - 1], dtype=np.uint32)
    draws = 6
    out = np.zeros((len(seeds), draws, k))
    words = np.zeros(len(seeds), dtype=np.uint32)
    assert lib.mzmcts_device_dirichlet(native.ptr(seeds, native.c_u32_p), len(seeds), alpha, k, draws,
                                       native.ptr(out, native.c_f64_p), native.ptr(words, native.c_u32_p)) == 0
    for i, seed in enumerate(seeds.tolist()):
        assert np.array_equal(out[i], fx[f"seed{seed}_dirichlet_{alpha}_{k}"]), (seed, alpha, k)
        # stream position afterwards: the host clone, advanced by the device's word count, draws the fixture's next doubles
        rng = native.HostRng(seed)
        for _ in range(int(words[i])):
            rng.next_u32()
        nxt = np.array([rng.random_sample(), rng.random_sample()])
        assert np.array_equal(nxt, fx[f"seed{seed}_dirichlet_{alpha}_{k}_next"])


def test_device_dirichlet_many_streams_equal_host_clone(native):
    """4096 streams x 8 draws of dirichlet([0.3] * 7) (Connect4's noise): the device rows and word counts equal the
    host clone's (libm's log / pow), stream by stream."""
    lib = native.load()
    n, k, draws, alpha = 4096, 7, 8, 0.3
    seeds = np.arange(1000, 1000 + n, dtype=np.uint32)
    out = np.zeros((n, draws, k))
    words = np.zeros(n, dtype=np.uint32)
    assert lib.mzmcts_device_dirichlet(native.ptr(seeds, native.c_u32_p), n, alpha, k, draws,
                                       native.ptr(out, native.c_f64_p), native.ptr(words, native.c_u32_p)) == 0
    for i in range(0, n, 37):
        rng = native.HostRng(int(seeds[i]))
        want = np.array([rng.dirichlet(alpha, k) for _ in range(draws)])
        assert np.array_equal(out[i], want), i


# ---- the engine with device-drawn noise ---------------------------------------------------------------------
from parity_helpers import (fixture_config, load_golden, run_injected_on_engine, run_injected_on_oracle,   # noqa: E402
                            streams_from_fixture)

TRACE_FILES = ["g4_cartpole_traces", "g5_tictactoe_traces", "g5_connect4_traces", "g5_cartpole_ties_traces",
               "g5_atari84_traces"]
EXACT_KEYS = ["noise", "visits", "child_value_sum", "child_prior", "child_reward", "root_value_sum", "root_visits",
              "max_tree_depth", "min_max", "sim_depth", "sim_actions", "sim_ties", "child_visits_target",
              "root_value_target", "action"]


@pytest.fixture(scope="module")
def eng(pkg):
    import importlib
    return importlib.import_module("muzero-hypermodel_amd.engine")


@pytest.mark.parametrize("name", TRACE_FILES)
def test_injected_traces_with_device_noise(eng, oracle, name):
    """The reference's recorded searches (masked TicTacToe / Connect4 roots included) with the exploration noise drawn
    by the GPU: the rows are the reference's own (fixture `noise`), and so is everything downstream of them --
    priors, paths, tie lists, visit counts, value sums, the sampled action (which needs the host mirror of the stream
    to have stepped over the device's draw)."""
    fx = load_golden(name)
    idx = list(range(len(fx["seed"])))
    temps = fx["temperature"].tolist()
    got = run_injected_on_engine(eng, None, fx, idx, temperature=temps, device_noise=True)
    want = run_injected_on_oracle(oracle, fx, idx=idx, temperature=temps)
    for key in EXACT_KEYS:
        assert np.array_equal(got[key], want[key]), f"{name}: {key} differs from the oracle"
    for key in ("noise", "visits", "child_value_sum", "child_prior", "child_reward"):
        assert np.array_equal(got[key], fx[key]), key
    assert np.array_equal(got["action"], fx["action_T"])


def test_consecutive_moves_device_noise_equals_host_noise(eng):
    """Three moves in a row on two engines, one drawing the noise on the host mirrors, one on the GPU: same noise,
    same statistics, same sampled actions, and the same stream state afterwards (the mirrors are kept level across
    Dirichlet draw, tie-breaks and action sampling whichever side consumed the words)."""
    fx = load_golden("g5_tictactoe_traces")
    idx = list(range(16))
    streams = streams_from_fixture(fx, idx)
    cfg = fixture_config(fx)
    S = cfg.num_simulations
    engines = [eng.BatchedMCTS(cfg, len(idx), seeds=streams["seeds"]) for _ in range(2)]
    engines[1].set_device_noise(True)
    results = []
    for engine in engines:
        moves = []
        for _ in range(3):
            engine.begin_search(streams["legal"], streams["to_play"], True)
            engine.expand_roots_injected(streams["root_reward"], streams["root_priors"])
            for s in range(S):
                engine.select(gather=False)
                engine.expand_backup_injected(streams["value"][:, s], streams["reward"][:, s], streams["priors"][:, s])
            st = {k: v.copy() for k, v in engine.readout().items()}
            actions, _ = engine.sample_actions(1.0)
            moves.append((engine.noise.copy(), st, actions))
        results.append((moves, [engine.get_rng_state(e) for e in range(len(idx))]))
        engine.close()
    (host_moves, host_rng), (dev_moves, dev_rng) = results
    for (n0, s0, a0), (n1, s1, a1) in zip(host_moves, dev_moves):
        assert np.array_equal(n0, n1)
        for key in s0:
            assert np.array_equal(s0[key], s1[key]), key
        assert np.array_equal(a0, a1)
    for r0, r1 in zip(host_rng, dev_rng):
        assert np.array_equal(r0[1], r1[1]) and r0[2:] == r1[2:]


def test_fused_search_with_device_noise(eng, pkg):
    """The whole-move kernels read device-drawn rows the same way (CartPole FC network, 64 envs)."""
    import importlib
    import torch
    from parity_helpers import cartpole_model_and_weights
    models = importlib.import_module("muzero-hypermodel_amd.models")
    config = importlib.import_module("muzero-hypermodel_amd.games.cartpole").MuZeroConfig()
    model, _ = cartpole_model_and_weights(models, config, "cuda")
    E = 64
    obs = np.random.RandomState(3).uniform(-0.05, 0.05, (E, 1, 1, 4)).astype(np.float32)
    out = []
    for device_noise in (False, True):
        engine = eng.BatchedMCTS(config, E, seeds=list(range(100, 100 + E)), group_width=16)
        engine.configure_fused_fc(model)
        engine.set_device_noise(device_noise)
        stats = []
        for _ in range(2):
            st = engine.search(model, obs, [[0, 1]] * E, [0] * E, True)
            actions, _ = engine.sample_actions(1.0)
            stats.append(({k: v.copy() for k, v in st.items()}, engine.noise.copy(), actions))
        out.append(stats)
        engine.close()
    for (s0, n0, a0), (s1, n1, a1) in zip(*out):
        assert np.array_equal(n0, n1) and (n1.sum(axis=1) > 0.999).all()
        for key in s0:
            assert np.array_equal(s0[key], s1[key]), key
        assert np.array_equal(a0, a1)
