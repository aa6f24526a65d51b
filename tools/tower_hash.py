#!/usr/bin/env python3
"""Regression aid for tower kernel work: recurrent inference of a game's residual network on fixed synthetic boards;
prints SHA-256 of every output (a change that should be bit-neutral must leave them alone) and the tower's time per launch.

    [MZ_LIB=other/libmzmcts.so] python tools/tower_hash.py [game=connect4] [boards for the hash=1001] [boards for the timing=8192]"""
import hashlib, importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from parity_helpers import synthetic_model
game = sys.argv[1] if len(sys.argv) > 1 else "connect4"
n_hash = int(sys.argv[2]) if len(sys.argv) > 2 else 1001
n_time = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
if os.environ.get("MZ_LIB"):                          # another build of the library (an A/B run against an earlier commit's)
    for name in ("build", "_native"):
        importlib.import_module("muzero-hypermodel_amd." + name).LIB_PATH = os.path.abspath(os.environ["MZ_LIB"])
models = importlib.import_module("muzero-hypermodel_amd.models")
config = importlib.import_module(f"muzero-hypermodel_amd.games.{game}").MuZeroConfig()
model, _ = synthetic_model(models, config, "cuda")
c, h, w = config.channels, config.observation_shape[1], config.observation_shape[2]
g = torch.Generator(device="cuda").manual_seed(0)
out = {"game": game}
with torch.no_grad():
    planes = torch.rand(n_hash, c + 1, h, w, generator=g, device="cuda")
    res = model.recurrent_inference_from_planes(planes)
    out["sha256"] = {k: hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
                     for k, t in zip(("value", "reward", "policy", "state"), res)}
    planes = torch.rand(n_time, c + 1, h, w, generator=g, device="cuda")
    state = torch.empty(n_time, c, h, w, device="cuda")
    for _ in range(3):
        model.recurrent_inference_from_planes(planes, out_state=state)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        model.recurrent_inference_from_planes(planes, out_state=state)
    b.record()
    torch.cuda.synchronize()
    out["recurrent_inference_us"] = 1e3 * a.elapsed_time(b) / 20
    out["boards"] = n_time
print(json.dumps(out))
