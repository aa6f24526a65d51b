"""include/mzmcts.h mzmcts_board_conv3x3 -- the residual networks' 3x3 convolution + BatchNorm2d (eval) + skip + ReLU
as one launch on the matrix cores (csrc/board_conv.hip; reference models.py:213-229, 318-330, 399-420) -- against:

  * integer-valued inputs and weights, where every product and partial sum is exact in fp32: the kernel must equal the
    integer convolution EXACTLY whatever the summation order (catches any tile / lane / tap / channel mix-up);
  * an fp64 convolution on random data, within the error of one fp32 fmaf chain over K = 9 * cin terms;
  * the module path it replaces (torch convolution + mzmcts_affine_act) on the residual block and on a whole Connect4
    network, within fp32 rounding of the different summation orders.
"""
import importlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # batch, cin, cout, h, w -- every kernel instantiation, ragged batches, padded channel counts
    (37, 64, 64, 6, 7), (1024, 65, 64, 6, 7), (3, 3, 64, 6, 7), (9, 16, 16, 6, 7), (50, 17, 16, 6, 6), (21, 16, 16, 6, 6),
    (7, 64, 64, 6, 6), (130, 17, 16, 3, 3), (64, 16, 16, 3, 3), (33, 64, 64, 3, 3), (1, 16, 16, 3, 3),
]


def _call(lib, x, weight, scale, shift, residual, relu):
    b, cin, h, w = x.shape
    cout = weight.shape[0]
    packed = torch.empty(lib.mzmcts_board_conv_packed_floats(cin, cout), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.mzmcts_board_conv_pack(weight.data_ptr(), packed.data_ptr(), cin, cout, stream) == 0
    out = torch.full((b, cout, h, w), float("nan"), device="cuda")
    rc = lib.mzmcts_board_conv3x3(x.data_ptr(), packed.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                  residual.data_ptr() if residual is not None else None, out.data_ptr(), b, cin, cout, h, w,
                                  1 if relu else 0, stream)
    assert rc == 0
    torch.cuda.synchronize()
    return out


@pytest.fixture(scope="module")
def lib(pkg):
    importlib.import_module("muzero-hypermodel_amd.build").build_native()
    return importlib.import_module("muzero-hypermodel_amd._native").load()


@pytest.mark.parametrize("shape", SHAPES)
def test_integer_data_is_exact(lib, shape):
    b, cin, cout, h, w = shape
    assert lib.mzmcts_board_conv_supported(cin, cout, h, w)
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randint(-3, 4, (b, cin, h, w), generator=g).float().cuda()
    weight = torch.randint(-2, 3, (cout, cin, 3, 3), generator=g).float().cuda()      # asymmetric in every index
    scale = torch.randint(1, 4, (cout,), generator=g).float().cuda()
    shift = torch.randint(-5, 6, (cout,), generator=g).float().cuda()
    residual = torch.randint(-4, 5, (b, cout, h, w), generator=g).float().cuda()
    conv = torch.nn.functional.conv2d(x.double().cpu(), weight.double().cpu(), padding=1)
    for res, relu in ((None, True), (residual, True), (residual, False), (None, False)):
        want = conv * scale.double().cpu().view(1, -1, 1, 1) + shift.double().cpu().view(1, -1, 1, 1)
        if res is not None:
            want = want + res.double().cpu()
        if relu:
            want = want.clamp_min(0)
        got = _call(lib, x, weight, scale, shift, res, relu)
        assert np.array_equal(got.cpu().numpy(), want.float().numpy()), (shape, res is not None, relu)


@pytest.mark.parametrize("shape", [(1024, 64, 64, 6, 7), (300, 65, 64, 6, 7), (4096, 16, 16, 3, 3), (512, 17, 16, 6, 6)])
def test_random_data_within_one_fp32_chain_of_fp64(lib, shape):
    b, cin, cout, h, w = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn((b, cin, h, w), generator=g).cuda()
    weight = (torch.randn((cout, cin, 3, 3), generator=g) / (9 * cin) ** 0.5).cuda()
    one, zero = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    got = _call(lib, x, weight, one, zero, None, False).double().cpu()
    want = torch.nn.functional.conv2d(x.double().cpu(), weight.double().cpu(), padding=1)
    magnitude = torch.nn.functional.conv2d(x.abs().double().cpu(), weight.abs().double().cpu(), padding=1)   # sum |a b|
    # a k-ordered fp32 fmaf chain: |error| <= ~K eps sum|a b| worst case, ~sqrt(K) eps typically (guide: 0.75-1.5e-7 sum|a b|)
    assert float(((got - want).abs() / magnitude).max()) < 4e-7
    # NaN / inf propagate like any fp32 arithmetic
    x[0, 0, 1, 1] = float("nan")
    got = _call(lib, x, weight, one, zero, None, True)
    assert torch.isnan(got[0, :, 0:3, 0:3]).all() and not torch.isnan(got[1:]).any()


def test_rejects_what_it_does_not_cover(lib):
    assert not lib.mzmcts_board_conv_supported(64, 32, 6, 7) and not lib.mzmcts_board_conv_supported(64, 64, 8, 8)
    x = torch.zeros(2, 64, 8, 8, device="cuda")
    out = torch.zeros(2, 64, 8, 8, device="cuda")
    s = torch.zeros(64, device="cuda")
    rc = lib.mzmcts_board_conv3x3(x.data_ptr(), x.data_ptr(), s.data_ptr(), s.data_ptr(), None, out.data_ptr(), 2, 64, 64, 8, 8,
                                  1, torch.cuda.current_stream().cuda_stream)
    assert rc != 0


def test_residual_block_and_connect4_network_follow_the_torch_path(pkg, monkeypatch):
    """The modules route Connect4-sized boards through the kernel; the torch convolution + affine_act path they took
    before gives the same numbers to fp32 rounding, and a weight refresh is seen (packed weights refilled in place)."""
    models = importlib.import_module("muzero-hypermodel_amd.models")
    from parity_helpers import synthetic_model
    config = importlib.import_module("muzero-hypermodel_amd.games.connect4").MuZeroConfig()
    model, _ = synthetic_model(models, config, "cuda")
    g = torch.Generator().manual_seed(3)
    obs = torch.randint(0, 2, (96, 3, 6, 7), generator=g).float().cuda()
    action = torch.randint(0, 7, (96, 1), generator=g).cuda()
    block = model.prediction_network.module.resblocks[0]
    with torch.no_grad():
        assert block.conv1.takes_mfma_path(torch.zeros(4, 64, 6, 7, device="cuda"))
        fused = model.recurrent_inference(model.initial_inference(obs)[3], action)
        packed_ptr = block.conv1.packed().data_ptr()
        monkeypatch.setenv("MZ_BOARD_CONV", "off")
        assert not block.conv1.takes_mfma_path(torch.zeros(4, 64, 6, 7, device="cuda"))
        plain = model.recurrent_inference(model.initial_inference(obs)[3], action)
        monkeypatch.setenv("MZ_BOARD_CONV", "auto")
        for a, b in zip(fused, plain):                     # (the hidden state is min-max rescaled: small spans amplify)
            torch.testing.assert_close(a, b, rtol=5e-5, atol=5e-5)
        # weight refresh behind the module's back (what a broadcast into the flat buffer does)
        block.conv1.weight.data.mul_(0.5)
        model.refresh_inference_constants()
        assert block.conv1.packed().data_ptr() == packed_ptr
        after = model.recurrent_inference(model.initial_inference(obs)[3], action)
        monkeypatch.setenv("MZ_BOARD_CONV", "off")
        plain_after = model.recurrent_inference(model.initial_inference(obs)[3], action)
    assert not torch.allclose(after[0], fused[0])
    for a, b in zip(after, plain_after):
        torch.testing.assert_close(a, b, rtol=5e-5, atol=5e-5)


@pytest.mark.parametrize("name,batch,precision", [("connect4", 37, "fp32"), ("connect4", 1024, "fp32"), ("connect4", 37, "split"),
                                                  ("connect4", 1024, "split"), ("tictactoe", 130, "fp32"), ("atari84", 50, "fp32")])
def test_recurrent_tower_equals_the_per_layer_path(pkg, monkeypatch, name, batch, precision):
    """mzmcts_board_tower (dynamics + rescale + prediction towers in one launch, activations resident in LDS) against
    the per-layer path.  Connect4's per-layer path is the same MFMA kernel per convolution and the same rescale
    operations, so the tower must reproduce it bit for bit; the small boards' per-layer path is the dense GEMM (another
    summation order): equal to fp32 rounding."""
    models = importlib.import_module("muzero-hypermodel_amd.models")
    from parity_helpers import synthetic_model
    if name == "atari84":
        config = importlib.import_module("muzero-hypermodel_amd.games.breakout").atari84_config()
    else:
        config = importlib.import_module(f"muzero-hypermodel_amd.games.{name}").MuZeroConfig()
    model, _ = synthetic_model(models, config, "cuda")
    g = torch.Generator().manual_seed(5)
    engine_mod = importlib.import_module("muzero-hypermodel_amd.engine")
    shape = engine_mod.hidden_state_shape(config)
    state = torch.rand((batch,) + tuple(shape), generator=g).cuda()
    action = torch.randint(0, len(config.action_space), (batch, 1), generator=g).cuda()
    out_state = torch.full_like(state, float("nan"))
    monkeypatch.setenv("MZ_BOARD_CONV_PRECISION", precision)
    with torch.no_grad():
        monkeypatch.setenv("MZ_BOARD_TOWER", "on")
        planes = models.state_action_planes(state, action, len(config.action_space))
        assert model._recurrent_tower(planes, None) is not None          # the tower path is really taken
        fused = model.recurrent_inference(state, action, out_state=out_state)
        assert fused[3].data_ptr() == out_state.data_ptr() and not torch.isnan(out_state).any()
        monkeypatch.setenv("MZ_BOARD_TOWER", "off")
        plain = model.recurrent_inference(state, action)
    for a, b, what in zip(fused, plain, ("value", "reward", "policy", "state")):
        if name == "connect4" and precision == "fp32":
            assert torch.equal(a, b), what
        elif precision == "split":
            # two-half fp16 operands, three products: fp32-level accuracy (the logits of 13 stacked convolutions agree
            # with the exact-fp32 chain to a few 1e-6; the rescaled state amplifies small spans)
            print(what, float((a - b).abs().max()))
            tol = 1e-4 if what == "state" else 5e-6       # (a plane's min-max rescale divides by its span)
            torch.testing.assert_close(a, b, rtol=tol, atol=tol, msg=what)
        else:
            torch.testing.assert_close(a, b, rtol=5e-5, atol=5e-5, msg=what)


def test_split_tower_single_layer_against_fp64(lib):
    """One 64 -> 64 layer through mzmcts_board_tower_split on random data, and a 65 -> 64 layer whose last input plane
    is constant per sample (the dynamics input): error against an fp64 convolution below an fp32 fmaf chain's."""
    native = importlib.import_module("muzero-hypermodel_amd._native")
    g = torch.Generator().manual_seed(11)
    for cin, const_plane in ((64, 0), (65, 1)):
        b, cout, h, w = 203, 64, 6, 7
        x = torch.randn((b, cin, h, w), generator=g).cuda()
        if const_plane:
            x[:, -1] = torch.rand((b, 1, 1), generator=g).cuda()
        weight = (torch.randn((cout, cin, 3, 3), generator=g) / (9 * cin) ** 0.5).cuda()
        halves = torch.empty(lib.mzmcts_board_conv_split_halfs(cin - const_plane, cout), dtype=torch.float16, device="cuda")
        table = torch.empty(cout * h * w, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        assert lib.mzmcts_board_conv_pack_split(weight.data_ptr(), halves.data_ptr(), table.data_ptr(), cin, cout, const_plane,
                                                h, w, stream) == 0
        one, zero = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        out = torch.full((b, cout, h, w), float("nan"), device="cuda")
        layer = (native.MzTowerLayer * 1)(native.MzTowerLayer(halves.data_ptr(), one.data_ptr(), zero.data_ptr(),
                                                              table.data_ptr() if const_plane else None, out.data_ptr(), None,
                                                              cin, 0, 0, 0))
        import ctypes
        assert lib.mzmcts_board_tower_split(x.data_ptr(), b, cin, const_plane, cout, h, w, ctypes.addressof(layer), 1, stream) == 0
        torch.cuda.synchronize()
        want = torch.nn.functional.conv2d(x.double().cpu(), weight.double().cpu(), padding=1)
        magnitude = torch.nn.functional.conv2d(x.abs().double().cpu(), weight.abs().double().cpu(), padding=1)
        err = float(((out.double().cpu() - want).abs() / magnitude).max())
        print(f"split tower layer {cin}->{cout}: max error / sum|a b| = {err:.2e}")
        assert err < 4e-7


def test_split_tower_overflow_falls_back_to_the_fp32_tower(pkg, monkeypatch):
    """The split-precision tower carries activations as fp16 halves scaled by 8: |x| >= 8188 does not fit.  With the
    hand-over gate (include/mzmcts.h mzmcts_tower_layer.gate) the launch flags the blocks of samples where that happened
    and the exact-fp32 tower behind it re-runs exactly those: the flagged samples come out as the fp32 tower computes them
    (bit for bit -- it IS that kernel), the others as the split tower does, nothing is inf / NaN, and the count of blocks
    that fell back is available to the host.  Without the gate the same input produces non-finite values."""
    models = importlib.import_module("muzero-hypermodel_amd.models")
    from parity_helpers import synthetic_model
    config = importlib.import_module("muzero-hypermodel_amd.games.connect4").MuZeroConfig()
    model, _ = synthetic_model(models, config, "cuda")
    batch = 203                                                    # 51 fp32-tower blocks of 4 samples, the last one ragged
    g = torch.Generator().manual_seed(3)
    state = torch.rand((batch, 64, 6, 7), generator=g).cuda()
    big = [5, 6, 77, 202]                                          # samples whose input is far outside the fp16 range / 8
    state[big] *= 40000.0
    action = torch.randint(0, 7, (batch, 1), generator=g).cuda()
    outs = {}
    with torch.no_grad():
        planes = models.state_action_planes(state, action, 7)
        for mode, env in (("pair", {}), ("fp32", {"MZ_BOARD_CONV_PRECISION": "fp32"}), ("split_only", {"MZ_SPLIT_FALLBACK": "off"})):
            for k in ("MZ_BOARD_CONV_PRECISION", "MZ_SPLIT_FALLBACK"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            if mode == "pair":
                model.split_tower_fallbacks()                      # (clears the counters)
            raw, unit, features = model._recurrent_tower(planes, None)
            outs[mode] = [t.clone() for t in (raw, unit, features)]
            if mode == "pair":
                fell_back = model.split_tower_fallbacks()
    # the split launch flags ITS workgroups (the count the host reads); the fp32 tower re-runs every one of its own
    # workgroups -- 4 boards -- that holds a flagged sample
    native = importlib.import_module("muzero-hypermodel_amd._native")
    per_split_block = 4 // int(native.load().mzmcts_board_tower_blocks(4, 64, 6, 7))
    assert fell_back == len({s // per_split_block for s in big})
    blocks = sorted({s // 4 for s in big})
    flagged = torch.zeros(batch, dtype=torch.bool)
    for blk in blocks:
        flagged[blk * 4: blk * 4 + 4] = True
    for got, exact, alone, what in zip(outs["pair"], outs["fp32"], outs["split_only"], ("raw", "unit", "features")):
        assert torch.isfinite(got).all(), what
        assert torch.equal(got[flagged], exact[flagged]), what      # re-run by the exact-fp32 kernel
        assert torch.equal(got[~flagged], alone[~flagged]), what    # untouched split results
        assert not torch.isfinite(alone[big]).all(), what           # (what the split tower alone makes of them)


def test_search_with_overflowing_activations_runs_on_the_fallback(pkg, monkeypatch):
    """A Connect4 network whose dynamics activations leave the split tower's range (batch-norm gains scaled up: what a
    trained network may do mid-run) searches without a NaN: every simulation's tower pair falls back, hipGraph-captured
    loop included, and the search equals the one run on the exact-fp32 towers from the start."""
    import numpy as np
    models = importlib.import_module("muzero-hypermodel_amd.models")
    eng = importlib.import_module("muzero-hypermodel_amd.engine")
    from parity_helpers import synthetic_model
    config = importlib.import_module("muzero-hypermodel_amd.games.connect4").MuZeroConfig()
    config.num_simulations = 30
    _, weights = synthetic_model(models, config, "cpu")
    weights = {k: v.clone() for k, v in weights.items()}
    weights["dynamics_network.module.bn.weight"] *= 3.0e4           # first dynamics layer: activations ~1e4 .. 1e5
    E = 16
    rs = np.random.RandomState(1)
    obs = rs.randint(-1, 2, (E, 3, 6, 7)).astype(np.float32)
    results = {}
    for mode in ("split", "fp32"):
        monkeypatch.setenv("MZ_BOARD_CONV_PRECISION", mode)
        model = models.MuZeroNetwork(config)
        model.set_weights(weights)
        model.cuda().eval()
        engine = eng.BatchedMCTS(config, E, seeds=list(range(E)), use_graph=True)
        out = []
        for _ in range(3):                                          # eager, capturing, replay
            st = engine.search(model, obs, [list(range(7))] * E, [0] * E, True)
            out.append((st["visits"].copy(), st["root_value_sum"].copy()))
        results[mode] = out
        if mode == "split":
            assert engine._graph is not None
            assert model.split_tower_fallbacks() >= 3 * config.num_simulations        # every launch handed blocks over
        engine.close()
    for (v_a, r_a), (v_b, r_b) in zip(results["split"], results["fp32"]):
        assert np.isfinite(r_a).all()
        assert np.array_equal(v_a, v_b) and np.array_equal(r_a, r_b)


@pytest.mark.parametrize("game,batch", [("tictactoe", 5), ("tictactoe", 130), ("tictactoe", 1000), ("tictactoe", 16384),
                                        ("atari84", 3), ("atari84", 130), ("atari84", 1001), ("atari84", 16384)])
def test_board_column_tower_is_bit_identical_to_the_row_tile_tower(pkg, monkeypatch, game, batch):
    """board_tower_cols_kernel / board_tower_patch_kernel (round 3: a wavefront owns whole boards -- 16 3 x 3 boards, or 4
    6 x 6 boards as 16 patches --, positions are the MFMA tiles, no barrier between layers, the layer's output replaces its
    input in LDS in place) against board_tower_kernel: the same k-ordered chains (minus exact zero terms on 3 x 3 boards),
    the same epilogue -- every export equal bit for bit, for the tensor input and for the input gathered from a
    hidden-state pool, ragged last wavefronts included."""
    models = importlib.import_module("muzero-hypermodel_amd.models")
    native = importlib.import_module("muzero-hypermodel_amd._native")
    from parity_helpers import synthetic_model
    if game == "atari84":
        config = importlib.import_module("muzero-hypermodel_amd.games.breakout").atari84_config()
        side, A = 6, 4
    else:
        config = importlib.import_module("muzero-hypermodel_amd.games.tictactoe").MuZeroConfig()
        side, A = 3, 9
    hidden = 16 * side * side
    model, _ = synthetic_model(models, config, "cuda")
    g = torch.Generator().manual_seed(batch)
    state = torch.rand((batch, 16, side, side), generator=g).cuda()
    state[::7] = 0.0                                              # flat planes: the rescale's span + 1e-5 branch
    action = torch.randint(0, A, (batch, 1), generator=g).cuda()
    outs = {}
    with torch.no_grad():
        planes = models.state_action_planes(state, action, A)
        # gathered form: a pool of 3 slabs whose rows are picked by a parent index per env
        pool = torch.rand((3, batch, hidden), generator=g).cuda()
        parent = torch.randint(0, 3, (batch,), generator=g, dtype=torch.int32).cuda()
        gather = native.MzTowerGather(pool.data_ptr(), parent.data_ptr(), action.data_ptr(), batch, hidden, float(A))
        for mode in ("off", "on"):
            monkeypatch.setenv("MZ_TOWER_COLS", mode)
            a = model._recurrent_tower(planes, None)
            b = model._recurrent_tower(None, None, gather=gather, shape=(batch, 17, side, side), device=state.device)
            assert a is not None and b is not None
            outs[mode] = [t.clone() for t in a + b]
    for x, y in zip(outs["off"], outs["on"]):
        assert torch.isfinite(y).all()
        assert torch.equal(x, y)
    # and it is the gathered rows it read: the tensor form on the gathered tensor gives the same
    with torch.no_grad():
        rows = pool[parent.long(), torch.arange(batch, device="cuda")].view(batch, 16, side, side)
        c = model._recurrent_tower(models.state_action_planes(rows, action, A), None)
    for x, y in zip(c, outs["on"][3:]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("batch", [7, 130, 4099])
def test_heads_inside_the_tower_launch_are_bit_identical(pkg, monkeypatch, batch):
    """mzmcts_board_tower_heads (opt-in, MZ_TOWER_HEADS=on): reward / value / policy heads computed in the tower launch from the activations in LDS
    (TicTacToe: the reward head on the raw dynamics output in mid-tower, value and policy on the last layer) return the
    logits of the two-launch form (tower exports + conv_head_mfma_kernel) bit for bit, and the same next state -- for the
    tensor input and the pool-gathered input."""
    models = importlib.import_module("muzero-hypermodel_amd.models")
    native = importlib.import_module("muzero-hypermodel_amd._native")
    from parity_helpers import synthetic_model
    config = importlib.import_module("muzero-hypermodel_amd.games.tictactoe").MuZeroConfig()
    model, _ = synthetic_model(models, config, "cuda")
    g = torch.Generator().manual_seed(100 + batch)
    state = torch.rand((batch, 16, 3, 3), generator=g).cuda()
    action = torch.randint(0, 9, (batch, 1), generator=g).cuda()
    pool = torch.rand((2, batch, 144), generator=g).cuda()
    parent = torch.randint(0, 2, (batch,), generator=g, dtype=torch.int32).cuda()
    gather = native.MzTowerGather(pool.data_ptr(), parent.data_ptr(), action.data_ptr(), batch, 144, 9.0)
    results = {}
    with torch.no_grad():
        planes = models.state_action_planes(state, action, 9)
        for mode in ("off", "on"):
            monkeypatch.setenv("MZ_TOWER_HEADS", mode)
            if mode == "on":
                assert model._recurrent_fused(planes, None) is not None      # the fused launch is really taken
            out_a, out_b = torch.empty_like(state), torch.empty_like(state)
            a = model.recurrent_inference_from_planes(planes, out_state=out_a)
            b = model.recurrent_inference_from_pool(gather, batch, out_state=out_b)
            results[mode] = [t.clone() for t in a + b]
    names = ("value", "reward", "policy", "state") * 2
    for x, y, what in zip(results["off"], results["on"], names):
        assert torch.isfinite(y).all(), what
        assert torch.equal(x, y), (what, float((x - y).abs().max()))
