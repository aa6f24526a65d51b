"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mzmcts.h declares,
fails loudly without a GPU, and its host RNG streams are numpy's legacy generator."""
import importlib
import os
import re

import numpy
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native(pkg):
    importlib.import_module("muzero-hypermodel_amd.build").build_native()
    return importlib.import_module("muzero-hypermodel_amd._native")


def declared_symbols():
    names = set()
    for header in ("mzmcts.h", "mzenv.h", "mzreplay.h", "mzhist.h", "mztrain.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(mz(?:mcts|env|replay|hist|train)_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol(native):
    lib = native.load()
    names = declared_symbols()
    assert len(names) >= 35
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/mzmcts.h but not exported"
    # and the Python binding covers exactly the header
    assert sorted(native.PROTOTYPES) == names
    assert lib.mzmcts_abi_version() == native.ABI_VERSION == 2


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_create_fails_loudly_without_gpu(native, pkg):
    import ctypes
    lib = native.load()
    cfg = native.MzConfig(num_envs=4, num_actions=2, num_simulations=5, num_players=1, support_size=10,
                          hidden_floats=8, device=0, discount=0.997, pb_c_base=19652, pb_c_init=1.25,
                          root_dirichlet_alpha=0.25, root_exploration_fraction=0.25, hidden_pool=None)
    handle = ctypes.c_void_p()
    rc = lib.mzmcts_create(ctypes.byref(cfg), ctypes.byref(handle))
    assert rc == native.ERR_HIP and not handle.value
    assert b"no CPU fallback" in lib.mzmcts_last_error(None)
    engine_mod = importlib.import_module("muzero-hypermodel_amd.engine")
    cartpole = importlib.import_module("muzero-hypermodel_amd.games.cartpole")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine_mod.BatchedMCTS(cartpole.MuZeroConfig(), 4)


def test_create_rejects_bad_configs(native):
    import ctypes
    lib = native.load()
    handle = ctypes.c_void_p()
    base = dict(num_envs=4, num_actions=2, num_simulations=5, num_players=1, support_size=10,
                hidden_floats=8, device=0, discount=0.997, pb_c_base=19652, pb_c_init=1.25,
                root_dirichlet_alpha=0.25, root_exploration_fraction=0.25, hidden_pool=None)
    cfg = native.MzConfig(**{**base, "num_players": 3})
    assert lib.mzmcts_create(ctypes.byref(cfg), ctypes.byref(handle)) == native.ERR_PLAYERS
    assert b"More than two player mode not implemented." in lib.mzmcts_last_error(None)
    cfg = native.MzConfig(**{**base, "num_envs": 0})
    assert lib.mzmcts_create(ctypes.byref(cfg), ctypes.byref(handle)) == native.ERR_INVALID


def test_downsample_launch_rejects_what_it_does_not_cover(native):
    """mzmcts_downsample_cnn's argument checks run before anything touches the GPU: shapes beyond config #5's family, a
    hidden width whose LDS plan does not fit a CU, misaligned frames and null pointers come back as MZMCTS_ERR_INVALID
    (the caller then keeps its convolution library); an empty batch of a covered shape is MZMCTS_OK."""
    lib = native.load()
    ptr = 0x10000                                  # never dereferenced: every call below returns before a launch

    def call(batch=0, c=4, h=84, w=84, mid=10, k1=12, cout=16, oh=6, ow=6, x=ptr, w1=ptr):
        return lib.mzmcts_downsample_cnn(x, batch, c, h, w, w1, ptr, mid, k1, ptr, ptr, cout, oh, ow, ptr, None)

    assert call() == 0
    assert call(c=3) == native.ERR_INVALID and call(h=96, w=96) == native.ERR_INVALID
    assert call(k1=8) == native.ERR_INVALID and call(cout=17) == native.ERR_INVALID and call(oh=9) == native.ERR_INVALID
    assert call(mid=3) == native.ERR_INVALID and call(mid=11) == native.ERR_INVALID      # 11: 164 KB of LDS
    assert call(x=ptr + 4) == native.ERR_INVALID and call(x=None) == native.ERR_INVALID and call(batch=-1) == native.ERR_INVALID


@pytest.mark.parametrize("seed", [0, 1, 12345, 2**32 - 1])
def test_host_rng_matches_numpy_fixture(native, golden, seed):
    fx = golden("g7_numpy_rng")
    r = native.HostRng(seed)
    assert [r.next_u32() for _ in range(4)] == fx[f"seed{seed}_words"].tolist()
    r.seed(seed)
    assert numpy.array_equal([r.random_sample() for _ in range(700)], fx[f"seed{seed}_doubles"])
    r.seed(seed)
    assert [r.choice(k) for k in (2, 3, 5, 7, 9, 4, 6, 8, 121, 1, 2)] == fx[f"seed{seed}_choice"].tolist()
    for alpha, k in ((0.25, 2), (0.1, 9), (0.3, 7), (0.25, 4), (1.0, 3), (2.5, 5), (0.03, 121)):
        r.seed(seed)
        got = numpy.array([r.dirichlet(alpha, k) for _ in range(6)])
        assert numpy.array_equal(got, fx[f"seed{seed}_dirichlet_{alpha}_{k}"])
        assert numpy.array_equal([r.random_sample(), r.random_sample()], fx[f"seed{seed}_dirichlet_{alpha}_{k}_next"])


def test_host_rng_live_numpy_and_state_exchange(native):
    numpy.random.seed(77)
    r = native.HostRng(77)
    for _ in range(200):
        k = int(numpy.random.randint(1, 10))
        assert r.choice(9) + 1 == k
        assert numpy.array_equal(numpy.random.dirichlet([0.25] * k), r.dirichlet(0.25, k))
        p = numpy.random.dirichlet([1.0] * 4)
        r.set_state(numpy.random.get_state())
        assert numpy.random.choice(4, p=p) == r.choice_p(p)
    numpy.random.standard_normal(1)  # cached gaussian travels with the state
    r.set_state(numpy.random.get_state())
    assert numpy.array_equal(numpy.random.dirichlet([3.0] * 3), r.dirichlet(3.0, 3))
    numpy.random.set_state(r.get_state())
    assert numpy.random.random_sample() == r.random_sample()


def test_host_select_action_g8(native, golden):
    fx = golden("g8_select_action")
    for i in range(int(fx["n_sets"])):
        visits, actions = fx[f"set{i}_visits"], fx[f"set{i}_actions"]
        for T in (0, 0.25, 0.5, 1.0, 0.7, float("inf")):
            r = native.HostRng(100 + i)
            picks = [int(actions[r.select_action(visits, T)]) for _ in range(12)]
            assert picks == fx[f"set{i}_T{T}"].tolist(), (i, T)


def test_host_rng_replay_sampling_helpers_match_numpy(native):
    """choice_p_many == RandomState.choice(n, size, p=p); choice_priorities == the reference's sample_position
    arithmetic (float32 priorities / Python sum of float32 scalars, replay_buffer.py:178-181)."""
    import numpy as np
    rs = np.random.RandomState(5)
    p32 = rs.rand(37).astype(np.float32)
    p32 /= np.sum(p32)
    got = native.HostRng(123).choice_p_many(p32, 50)
    want = np.random.RandomState(123).choice(37, 50, p=p32)
    assert np.array_equal(got, want)
    priorities = (rs.rand(23) * 3).astype(np.float32)
    probs = priorities / sum(priorities)
    for seed in (1, 2, 3):
        idx, prob = native.HostRng(seed).choice_priorities(priorities)
        want_idx = np.random.RandomState(seed).choice(len(probs), p=probs)
        assert idx == want_idx and prob == probs[want_idx]
