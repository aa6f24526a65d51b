#!/usr/bin/env python3
"""Diagnostic: where does board_tower_kernel spend its cycles?  Builds a SEPARATE library (tools/_stamps/, with
-DMZ_TOWER_STAMPS) whose tower kernel sums wave 0's cycles per phase, runs the recurrent inference of a residual network
on synthetic boards and prints the shares (stamps perturb: shares, not run times).

    python tools/stamp_tower.py [game=tictactoe] [boards=65536]"""
import ctypes, importlib, json, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
game = sys.argv[1] if len(sys.argv) > 1 else "tictactoe"
boards = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
extra = sys.argv[3:]                                   # further -D switches of the diagnostic build (MZ_TOWER_NO_WLOAD, MZ_TOWER_NO_ALOAD)
build = importlib.import_module("muzero-hypermodel_amd.build")
out_dir = os.path.join(ROOT, "tools", "_stamps")
os.makedirs(out_dir, exist_ok=True)
lib_path = os.path.join(out_dir, "libmzmcts_tower.so")
subprocess.check_call([build._hipcc()] + build.HIPCC_FLAGS + ["-shared", "-DMZ_TOWER_STAMPS"] + ["-D" + e for e in extra] + ["-o", lib_path] + build.SOURCES, cwd=build.CSRC)
native = importlib.import_module("muzero-hypermodel_amd._native")
native.LIB_PATH = lib_path
build.LIB_PATH = lib_path
models = importlib.import_module("muzero-hypermodel_amd.models")
from parity_helpers import synthetic_model
config = importlib.import_module(f"muzero-hypermodel_amd.games.{game}").MuZeroConfig()
model, _ = synthetic_model(models, config, "cuda")
lib = native.load()
c, h, w = config.channels, config.observation_shape[1], config.observation_shape[2]
g = torch.Generator(device="cuda").manual_seed(0)
planes = torch.rand(boards, c + 1, h, w, generator=g, device="cuda")
out_state = torch.empty(boards, c, h, w, device="cuda")
stamps = (ctypes.c_ulonglong * 16)()
with torch.no_grad():
    for _ in range(3):
        model.recurrent_inference_from_planes(planes, out_state=out_state)
    torch.cuda.synchronize()
    lib.mzmcts_tower_stamps(stamps, 1)
    for _ in range(10):
        model.recurrent_inference_from_planes(planes, out_state=out_state)
    torch.cuda.synchronize()
    lib.mzmcts_tower_stamps(stamps, 1)
per_wg = {"tictactoe": 14, "connect4": int(os.environ.get("MZ_SPLIT_BOARDS", "2"))}.get(game, 4)    # boards per workgroup (board_conv.hip's dispatch at large batch)
names = ["zero LDS", "input fill", "main loops (MFMA)", "epilogues", "barrier after a layer", "exports + rescale"]
vals = [int(stamps[i]) for i in range(6)]
total = sum(vals) or 1
print(json.dumps({"game": game, "boards": boards, "switches": extra, "env": {k: v for k, v in os.environ.items() if k.startswith("MZ_")}, "shares": {n: round(v / total, 4) for n, v in zip(names, vals)},
                  "cycles_per_workgroup_wave0": {n: round(v / 10 / max(1, -(-boards // per_wg))) for n, v in zip(names, vals)}}))
