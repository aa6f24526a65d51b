"""The package's networks (muzero-hypermodel_amd/models.py) against outputs recorded from the
reference's models.py (fixtures G1-G3), on CPU torch: same ops, so agreement is to the last bits."""
import importlib

import numpy
import pytest
import torch

from parity_helpers import cartpole_model_and_weights, synthetic_model


@pytest.fixture(scope="module")
def models_mod(pkg):
    return importlib.import_module("muzero-hypermodel_amd.models")


def game_config(name):
    if name == "atari84":
        return importlib.import_module("muzero-hypermodel_amd.games.breakout").atari84_config()
    return importlib.import_module(f"muzero-hypermodel_amd.games.{name}").MuZeroConfig()


def check_inference(model, fx, tol):
    with torch.no_grad():
        v0, r0, p0, h0 = model.initial_inference(torch.from_numpy(fx["obs"]))
        out = torch.empty_like(h0)
        v1, r1, p1, h1 = model.recurrent_inference(torch.from_numpy(fx["init_hidden"]),
                                                   torch.from_numpy(fx["actions"]), out_state=out)
    assert h1.data_ptr() == out.data_ptr()          # next state written in place
    for got, key in ((v0, "init_value"), (p0, "init_policy"), (h0, "init_hidden"), (v1, "rec_value"),
                     (r1, "rec_reward"), (p1, "rec_policy"), (h1, "rec_hidden")):
        numpy.testing.assert_allclose(got.numpy(), fx[key], rtol=tol, atol=tol, err_msg=key)
    assert numpy.array_equal(r0.numpy(), fx["init_reward"])


def test_fc_network_matches_reference(models_mod, golden):
    model, _ = cartpole_model_and_weights(models_mod, game_config("cartpole"))
    check_inference(model, golden("g2_fc_inference"), 1e-6)


@pytest.mark.parametrize("name", ["tictactoe", "connect4", "atari84"])
def test_resnet_matches_reference(models_mod, golden, name):
    fx = golden(f"g3_{name}_inference")
    model, _ = synthetic_model(models_mod, game_config(name))
    assert list(model.state_dict().keys()) == fx["state_dict_keys"].tolist()   # DataParallel-style keys
    # 1e-5: eval-mode batch norm is evaluated as the folded affine x*scale+shift (models.BatchNorm2d),
    # which rounds differently from torch's (x-mean)*invstd*w+b in the last fp32 bits
    check_inference(model, fx, 1e-5)


def test_state_dict_keys_and_param_counts(models_mod, golden):
    w = golden("cartpole_weights")
    model = models_mod.MuZeroNetwork(game_config("cartpole"))
    assert list(model.state_dict().keys()) == list(w.files)
    assert sum(p.numel() for p in model.parameters()) == 1532
    counts = {"tictactoe": 21715, "connect4": 730681}
    for name, n in counts.items():
        m = models_mod.MuZeroNetwork(game_config(name))
        assert sum(p.numel() for p in m.parameters()) == n
    cfg = game_config("cartpole")
    cfg.network = "transformer"
    with pytest.raises(NotImplementedError, match='should be "fullyconnected" or "resnet"'):
        models_mod.MuZeroNetwork(cfg)


def test_support_to_scalar_g1(models_mod, golden):
    fx = golden("g1_support_to_scalar")
    for logits, out, s in (("logits21", "out21", 10), ("logits601", "out601", 300), ("logits_init", "out_init", 10)):
        got = models_mod.support_to_scalar(torch.from_numpy(fx[logits]), s).numpy()
        assert numpy.array_equal(got, fx[out])


def test_scalar_to_support_roundtrip(models_mod):
    x = torch.tensor([[0.0, 1.5, -3.25, 99.0, -120.0]])
    enc = models_mod.scalar_to_support(x, 10)
    assert enc.shape == (1, 5, 21)
    numpy.testing.assert_allclose(enc.sum(-1).numpy(), 1.0, rtol=0, atol=1e-6)
    dec = models_mod.support_to_scalar(torch.log(enc[0] + 1e-30), 10)[:, 0]
    numpy.testing.assert_allclose(dec.numpy()[:4], x[0].numpy()[:4], rtol=2e-3, atol=2e-3)
