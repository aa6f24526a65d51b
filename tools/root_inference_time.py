"""Steady-state time of initial_inference (the root of every move) for a game's network at a batch, with the kernels of
one call (torch profiler): python tools/root_inference_time.py [game=atari84] [batch=32768]"""
import importlib, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from parity_helpers import synthetic_model
game = sys.argv[1] if len(sys.argv) > 1 else "atari84"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
models = importlib.import_module("muzero-hypermodel_amd.models")
if game == "atari84":
    config = importlib.import_module("muzero-hypermodel_amd.games.breakout").atari84_config()
else:
    config = importlib.import_module(f"muzero-hypermodel_amd.games.{game}").MuZeroConfig()
model, _ = synthetic_model(models, config, "cuda")
obs = torch.rand((b,) + tuple(config.observation_shape), device="cuda")
with torch.no_grad():
    for _ in range(3):
        model.initial_inference(obs)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        model.initial_inference(obs)
    e.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(e) / 5
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        model.initial_inference(obs)
        torch.cuda.synchronize()
rows = sorted(((ev.key[:70], ev.count, ev.device_time_total) for ev in prof.key_averages() if ev.device_time_total > 0), key=lambda r: -r[2])
print(json.dumps({"game": game, "batch": b, "initial_inference_ms": ms, "kernels_us": [[k, n, round(t, 1)] for k, n, t in rows[:14]]}, indent=1))
