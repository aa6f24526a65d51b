"""Trainer against the reference's Trainer.update_weights on a recorded batch (fixture G14):
same learning rates, losses, new priorities and weights after one and two Adam steps (torch CPU on both sides)."""
import importlib

import numpy as np
import torch

from parity_helpers import load_golden


def batch_of(fx, as_tensors):
    keys = ("observation_batch", "action_batch", "value_batch", "reward_batch", "policy_batch", "weight_batch",
            "gradient_scale_batch")
    if as_tensors:
        return tuple(fx[k] if k == "weight_batch" else torch.from_numpy(fx[k]) for k in keys)
    return tuple(fx[k] if k == "weight_batch" else fx[k].tolist() for k in keys)


def run_steps(pkg, fx, device, as_tensors):
    tr_mod = importlib.import_module("muzero-hypermodel_amd.trainer")
    config = importlib.import_module("muzero-hypermodel_amd.games.cartpole").MuZeroConfig()
    config.batch_size = 32
    w = load_golden("cartpole_weights")
    weights = {k: torch.from_numpy(w[k]) for k in w.files}
    tr = tr_mod.Trainer({"weights": weights, "training_step": 0, "optimizer_state": None}, config, device=device)
    out = []
    for step in range(2):
        tr.update_lr()
        lr = tr.optimizer.param_groups[0]["lr"]
        batch = batch_of(fx, as_tensors)
        if as_tensors and device != "cpu":
            batch = tuple(b.to(device) if torch.is_tensor(b) else b for b in batch)
        priorities, total, v, r, p = tr.update_weights(batch)
        out.append((lr, priorities, np.array([total, v, r, p]), {k: t.detach().cpu().numpy().copy() for k, t in tr.model.get_weights().items()}))
    return tr, out


def check(fx, out, tol):
    for step, (lr, priorities, losses, weights) in enumerate(out):
        assert lr == float(fx[f"lr{step}"])
        np.testing.assert_allclose(losses, fx[f"losses{step}"], rtol=tol, atol=tol)
        np.testing.assert_allclose(priorities, fx[f"priorities{step}"], rtol=10 * tol, atol=10 * tol)
        for k, got in weights.items():
            np.testing.assert_allclose(got, fx[f"w{step}_{k}"], rtol=10 * tol, atol=tol, err_msg=f"step {step} {k}")


def test_trainer_matches_reference_on_cpu(pkg):
    fx = load_golden("g14_trainer_cartpole")
    for as_tensors in (False, True):          # the reference's list batches and the device store's tensors
        tr, out = run_steps(pkg, fx, "cpu", as_tensors)
        check(fx, out, 2e-6)
    assert tr.training_step == 2


def test_trainer_publishes_into_flat_weights(pkg):
    fx = load_golden("g14_trainer_cartpole")
    weights_mod = importlib.import_module("muzero-hypermodel_amd.weights")
    models = importlib.import_module("muzero-hypermodel_amd.models")
    tr, _ = run_steps(pkg, fx, "cpu", True)
    actor_model = models.MuZeroNetwork(tr.config)
    flat = weights_mod.FlatWeights(actor_model)
    tr.publish(flat)
    for k, t in tr.model.get_weights().items():
        assert torch.equal(actor_model.state_dict()[k], t), k
