#!/usr/bin/env python3
"""Diagnostic: where does the fused whole-move kernel spend its cycles?

Builds a SEPARATE library (tools/_stamps/libmzmcts.so, -DMZ_STAMPS) whose fused kernel accumulates
s_memtime differences per phase into a debug buffer, runs a few moves of the bench workload and prints
the shares.  Never used by the product path; the numbers are shares, not run times (stamps perturb).
"""
import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    group = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    variant = sys.argv[3] if len(sys.argv) > 3 else "auto"
    build = importlib.import_module("muzero-hypermodel_amd.build")
    out_dir = os.path.join(ROOT, "tools", "_stamps")
    os.makedirs(out_dir, exist_ok=True)
    lib_path = os.path.join(out_dir, "libmzmcts.so")
    cmd = [build._hipcc()] + build.HIPCC_FLAGS + ["-shared", "-DMZ_STAMPS", "-fno-slp-vectorize", "-o", lib_path] + build.SOURCES
    if not (os.environ.get("MZ_STAMPS_PREBUILT") and os.path.exists(lib_path)):   # (build here, measure on the GPU box)
        subprocess.check_call(cmd, cwd=build.CSRC)
    if os.environ.get("MZ_STAMPS_BUILD_ONLY"):
        return
    build.LIB_PATH = lib_path
    native = importlib.import_module("muzero-hypermodel_amd._native")
    native.LIB_PATH = lib_path
    engine_mod = importlib.import_module("muzero-hypermodel_amd.engine")
    models = importlib.import_module("muzero-hypermodel_amd.models")
    cartpole = importlib.import_module("muzero-hypermodel_amd.games.cartpole")
    from parity_helpers import cartpole_model_and_weights
    config = cartpole.MuZeroConfig()
    model, _ = cartpole_model_and_weights(models, config, "cuda")
    engine = engine_mod.BatchedMCTS(config, E, group_width=group)
    engine.configure_fused_fc(model)
    engine.set_fused_options(variant, publish_tree=False)
    variant = engine.fused_variant()
    if variant == "narrow":
        names_narrow = {8: "  descend: fetch child records", 9: "  descend: score (table + division)",
                        10: "  descend: arg-max + ties", 11: "  descend: publish + loop"}
    lib = native.load()
    lib.mzmcts_debug_read_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int32]
    obs = torch.from_numpy(np.random.RandomState(0).uniform(-0.05, 0.05, (E, 1, 1, 4)).astype(np.float32)).cuda()
    legal = np.tile(np.arange(2, dtype=np.int32), (E, 1))
    nl = np.full(E, 2, np.int32)
    sums = (ctypes.c_ulonglong * 16)()
    engine.search_fused(obs, legal, np.zeros(E, np.int32), True, num_legal=nl)
    lib.mzmcts_debug_read_stamps(sums, 1)
    moves = 5
    for _ in range(moves):
        engine.search_fused(obs, legal, np.zeros(E, np.int32), True, num_legal=nl)
    torch.cuda.synchronize()
    lib.mzmcts_debug_read_stamps(sums, 1)
    names = ["stage tables+weights", "root inference+expand", "descend", "fc_recurrent", "decode+softmax",
             "write children+hidden", "backup", "publish", "  fc: stage x", "  fc: dynamics L1", "  fc: dynamics L2",
             "  fc: (pre 3)", "  fc: rescale", "  fc: heads L1", "  fc: heads L2", "  fc: (post 3)"]
    if variant == "narrow":
        for k, v in names_narrow.items():
            names[k] = v
    trees_per_wg = 16 if variant == "narrow" else 64 // group      # narrow: 4 wavefronts per workgroup at this size
    wgs = (E + trees_per_wg - 1) // trees_per_wg
    total = sum(sums)
    print(f"variant={variant} group={group} E={E} workgroups={wgs}: mean cycles per workgroup per move, share")
    for n, v in zip(names, sums):
        if not v:
            continue
        per = v / wgs / moves
        print(f"  {n:26s} {per:12.0f} cycles  {100 * v / total:5.1f}%   per-sim {per / config.num_simulations:9.0f}")
    print(f"  total {total / wgs / moves:.0f} cycles per workgroup per move")
    engine.close()


if __name__ == "__main__":
    main()
