"""cProfile of DeviceSelfPlay.step (host-side phases).  python tools/selfplay_profile.py [game] [envs]"""
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
mod = importlib.import_module(f"muzero-hypermodel_amd.games.{game}")
config = mod.MuZeroConfig()
torch.manual_seed(0)
weights = models.MuZeroNetwork(config).get_weights()
actor = sp.DeviceSelfPlay({"weights": weights}, game, config, 0, E)
sink = []
for _ in range(5):
    actor.step(1.0, None, on_games=lambda b: None)
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    actor.step(1.0, None, on_games=lambda b: None)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
