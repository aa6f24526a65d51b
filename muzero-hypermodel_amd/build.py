"""Builds libmzmcts.so (HIP kernels + C ABI) for gfx950 in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting
.so is git-ignored but travels to the GPU box with the working tree.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmzmcts.so")
SOURCES = ["mcts_kernels.hip", "fused_narrow.hip", "mzmcts_capi.hip", "mzmcts_moves.hip", "mzmcts_rng.hip", "mzhist.hip", "env_kernels.hip", "mzenv_capi.hip", "mzreplay.hip", "net_kernels.hip", "board_conv.hip", "trainer_kernels.hip"]
HEADERS = ["engine_host.h", "np_legacy_rng.h", "glibc_libm.h", "glibc_libm_tables.inc", "tree_layout.h", "tree_device.h", "fc_net_device.h", "narrow_device.h", "kernel_common.h", "env_layout.h", os.path.join("..", "..", "include", "mzmcts.h"),
           os.path.join("..", "..", "include", "mzenv.h"), os.path.join("..", "..", "include", "mzreplay.h"), os.path.join("..", "..", "include", "mzhist.h"), os.path.join("..", "..", "include", "mztrain.h")]

# -fno-slp-vectorize: packing adjacent scalar f32 FMAs into v_pk_fma_f32 costs more in register shuffles
# than it saves in issue slots here (measured: fused kernel 595 -> 555 us).
# -ffp-contract=off is part of the numerical contract: the fp64 UCB / backup arithmetic must not be
# fused into FMAs or it stops being bit-identical to the reference's Python floats.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-variable"] + os.environ.get("MZMCTS_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X MCTS engine cannot be built without ROCm")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > built for d in deps)


def build_native(force=False, verbose=False):
    """Compile csrc/*.hip into muzero-hypermodel_amd/libmzmcts.so; returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH] + SOURCES
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)
    return LIB_PATH


def build_tools(force=False):
    """Stand-alone measurement programs under tools/ (no library, no torch) -> tools/_bin/."""
    tools = os.path.join(os.path.dirname(PKG_DIR), "tools")
    out_dir = os.path.join(tools, "_bin")
    os.makedirs(out_dir, exist_ok=True)
    built = []
    for name in ("random_access_ceiling", "record_size_ceiling"):
        src, out = os.path.join(tools, name + ".hip"), os.path.join(out_dir, name)
        if force or not os.path.exists(out) or os.path.getmtime(src) > os.path.getmtime(out):
            proc = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-o", out, src],
                                  capture_output=True, text=True)
            if proc.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)
        built.append(out)
    return built


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
    print(build_tools(force=True))
