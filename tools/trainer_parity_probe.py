"""Which weights leave the reference's after an Adam step on the GPU, and why (fixtures G14 / G16).

Adam's first step is lr * g / (|g| + eps): the sign of the gradient for every |g| >> eps = 1e-8.  Prints, per fixture,
how many entries deviate from the reference's recorded weights after step 0, and the size of their CPU gradient relative
to the largest gradient of their tensor -- the deviating entries are the ones whose gradient is zero up to the rounding of
a backward pass that sums in another order."""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from parity_helpers import load_golden  # noqa: E402
from test_trainer_cpu import batch_of  # noqa: E402


def setup(name):
    tr_mod = importlib.import_module("muzero-hypermodel_amd.trainer")
    models = importlib.import_module("muzero-hypermodel_amd.models")
    if name == "g14":
        fx = load_golden("g14_trainer_cartpole")
        config = importlib.import_module("muzero-hypermodel_amd.games.cartpole").MuZeroConfig()
        config.batch_size = 32
        w = load_golden("cartpole_weights")
        weights = {k: torch.from_numpy(w[k]) for k in w.files}
    else:
        from synth import synthetic_state_dict
        fx = load_golden("g16_trainer_tictactoe")
        config = importlib.import_module("muzero-hypermodel_amd.games.tictactoe").MuZeroConfig()
        config.batch_size = 24
        template = models.MuZeroNetwork(config).state_dict()
        weights = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(template, 0).items()}
    return tr_mod, config, fx, weights


def two_steps(tr_mod, config, fx, weights, device):
    """Two Trainer steps; returns per step (gradients, weights afterwards) and the learning rates."""
    tr = tr_mod.Trainer({"weights": {k: v.clone() for k, v in weights.items()}, "training_step": 0,
                         "optimizer_state": None}, config, device=device)
    batch = batch_of(fx, True)
    if device != "cpu":
        batch = tuple(b.to(device) if torch.is_tensor(b) else b for b in batch)
    out = []
    for _ in range(2):
        tr.update_lr()
        tr.update_weights(batch)
        grads = {k: p.grad.detach().cpu().numpy().copy() for k, p in tr.model.named_parameters()}
        w = {k: t.detach().cpu().numpy().copy() for k, t in tr.model.get_weights().items()}
        out.append((grads, w, float(tr._lr_host)))
    return out


def main():
    for name in ("g14", "g16"):
        tr_mod, config, fx, weights = setup(name)
        cpu = two_steps(tr_mod, config, fx, weights, "cpu")
        gpu = two_steps(tr_mod, config, fx, weights, "cuda")
        for step in (0, 1):
            if f"w{step}_{next(iter(cpu[0][0]))}" not in fx.files:
                continue
            lr = cpu[step][2]
            report = dict(fixture=name, step=step, lr=lr, entries=0, off=0, off_cpu=0, worst_dev_over_lr=0.0,
                          worst_rel_grad_of_off=0.0, noise_floor=0.0)
            detail = []
            for k in cpu[0][0]:
                ref = fx[f"w{step}_{k}"]
                dev = np.abs(gpu[step][1][k] - ref)
                dev_cpu = np.abs(cpu[step][1][k] - ref)
                # the smallest gradient the entry saw so far, relative to the largest of its tensor in that step
                rel = np.full(ref.shape, np.inf)
                noise = 0.0
                for s in range(step + 1):
                    g = np.abs(fx[f"g{s}_{k}"]) if f"g{s}_{k}" in fx.files else np.abs(cpu[s][0][k])   # the reference's own, if recorded
                    rel = np.minimum(rel, g / max(float(g.max()), 1e-30))
                    noise = max(noise, float(np.abs(gpu[s][0][k] - g * np.sign(cpu[s][0][k])).max() / max(float(g.max()), 1e-30)))
                off = dev > 1e-6
                report["entries"] += int(dev.size)
                report["off"] += int(off.sum())
                report["off_cpu"] += int((dev_cpu > 1e-6).sum())
                report["worst_dev_over_lr"] = max(report["worst_dev_over_lr"], float(dev.max() / lr))
                report["noise_floor"] = max(report["noise_floor"], noise)
                if off.any():
                    report["worst_rel_grad_of_off"] = max(report["worst_rel_grad_of_off"], float(rel[off].max()))
                    detail.append((k, int(off.sum()), int(dev.size), float(rel[off].max()), float(dev.max() / lr)))
            print(json.dumps(report))
            for row in detail:
                print("    ", row)


if __name__ == "__main__":
    main()
