"""cProfile of DeviceSelfPlay (host-side phases).  python tools/selfplay_profile.py [game] [envs] [step|batch]"""
import cProfile
import importlib
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

sp = importlib.import_module("muzero-hypermodel_amd.self_play")
models = importlib.import_module("muzero-hypermodel_amd.models")
game = sys.argv[1] if len(sys.argv) > 1 else "cartpole"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
mode = sys.argv[3] if len(sys.argv) > 3 else "step"
mod = importlib.import_module(f"muzero-hypermodel_amd.games.{game}")
config = mod.MuZeroConfig()
torch.manual_seed(0)
weights = models.MuZeroNetwork(config).get_weights()
actor = sp.DeviceSelfPlay({"weights": weights}, game, config, 0, E)
if mode == "batch":
    actor.engine.set_fused_options("auto", publish_tree=False)
    run = lambda: actor.play_moves(20, 1.0, on_games=lambda b: None)   # noqa: E731
    reps = 8
else:
    run = lambda: actor.step(1.0, None, on_games=lambda b: None)       # noqa: E731
    reps = 30
for _ in range(3):
    run()
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    run()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(24)
