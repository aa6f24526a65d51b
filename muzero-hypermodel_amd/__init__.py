"""MI355X-native MuZero self-play / MCTS engine.

The directory name carries the upstream project's hyphen, so import the package by string:

    import importlib
    mz = importlib.import_module("muzero-hypermodel_amd")
    self_play = importlib.import_module("muzero-hypermodel_amd.self_play")   # drop-in for self_play.py

Layout:
    csrc/          HIP kernels (gfx950) + the C ABI of include/mzmcts.h  -> libmzmcts.so
    _native.py     ctypes binding (fails loudly when the library or the GPU is missing)
    engine.py      BatchedMCTS: Python host of the batched search
    self_play.py   SelfPlay / MCTS / Node / GameHistory / MinMaxStats with the reference's surface
    models.py      the reference's networks (same state-dict keys), PyTorch-ROCm inference
    weights.py     flat fp32 weight buffer + RCCL broadcast to the per-GPU actors
    actor.py       one self-play actor per GPU (torch.distributed, no Ray)
    games/         Game plugins + MuZeroConfig (reference plugin API)
"""
from .build import build_native  # noqa: F401

__all__ = ["build_native"]
