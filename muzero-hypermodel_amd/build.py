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
SOURCES = ["mcts_kernels.hip", "fused_narrow.hip", "mzmcts_capi.hip", "mzmcts_moves.hip", "mzmcts_rng.hip", "mzhist.hip", "env_kernels.hip", "mzenv_capi.hip", "mzreplay.hip", "net_kernels.hip", "board_conv.hip", "downsample_cnn.hip", "trainer_kernels.hip"]
HEADERS = ["engine_host.h", "np_legacy_rng.h", "glibc_libm.h", "glibc_libm_tables.inc", "tree_layout.h", "tree_device.h", "fc_net_device.h", "narrow_device.h", "kernel_common.h", "env_layout.h", os.path.join("..", "..", "include", "mzmcts.h"),
           os.path.join("..", "..", "include", "mzenv.h"), os.path.join("..", "..", "include", "mzreplay.h"), os.path.join("..", "..", "include", "mzhist.h"), os.path.join("..", "..", "include", "mztrain.h")]

# -fno-slp-vectorize: packing adjacent scalar f32 FMAs into v_pk_fma_f32 costs more in register shuffles
# than it saves in issue slots here (measured: fused kernel 595 -> 555 us).
# -ffp-contract=off is part of the numerical contract: the fp64 UCB / backup arithmetic must not be
# fused into FMAs or it stops being bit-identical to the reference's Python floats.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-variable"] + os.environ.get("MZMCTS_EXTRA_HIPCC_FLAGS", "").split()
OBJ_DIR = os.path.join(CSRC, "_obj")


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X MCTS engine cannot be built without ROCm")


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")


def _flags_stamp():
    return " ".join(HIPCC_FLAGS)


def _object_deps(src):
    """Prerequisites of one object: the compiler's own -MMD list when there is one, else every header."""
    dep = os.path.splitext(_obj(src))[0] + ".d"
    if os.path.exists(dep):
        text = open(dep).read().replace("\\\n", " ")
        names = text.split(":", 1)[1].split() if ":" in text else []
        if names:
            return [n if os.path.isabs(n) else os.path.join(CSRC, n) for n in names]
    return [os.path.join(CSRC, s) for s in [src] + HEADERS]


def _stale_objects():
    stamp = os.path.join(OBJ_DIR, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != _flags_stamp():
        return list(SOURCES)
    stale = []
    for src in SOURCES:
        obj = _obj(src)
        if not os.path.exists(obj):
            stale.append(src)
            continue
        built = os.path.getmtime(obj)
        if any((not os.path.exists(d)) or os.path.getmtime(d) > built for d in _object_deps(src)):
            stale.append(src)
    return stale


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    if not os.path.isdir(OBJ_DIR):
        # a tree that arrived with a prebuilt library only (the GPU box): compare against the sources
        built = os.path.getmtime(LIB_PATH)
        return any(os.path.getmtime(os.path.join(CSRC, d)) > built for d in SOURCES + HEADERS)
    if _stale_objects():
        return True
    built = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(_obj(s)) > built for s in SOURCES)


def build_native(force=False, verbose=False, jobs=None):
    """Compile csrc/*.hip into muzero-hypermodel_amd/libmzmcts.so; returns its path.

    One object per source under csrc/_obj/ (with the compiler's dependency lists), compiled in parallel and only when
    the source, a header it includes or the flags changed; one link.  Touching a kernel file costs that file's compile."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = list(SOURCES) if force else _stale_objects()
    hipcc = _hipcc()
    jobs = jobs or min(len(todo) or 1, max(1, (os.cpu_count() or 2)))
    procs, failures = [], []

    def reap(block):
        for item in list(procs):
            src, proc = item
            if block or proc.poll() is not None:
                out, err = proc.communicate()
                if proc.returncode != 0:
                    failures.append(f"{src}:\n{out}{err}")
                procs.remove(item)
                if block:
                    return

    for src in todo:
        while len(procs) >= jobs:
            reap(True)
        obj = _obj(src)
        cmd = [hipcc] + HIPCC_FLAGS + ["-c", "-MMD", "-MF", os.path.splitext(obj)[0] + ".d", "-o", obj, src]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    while procs:
        reap(True)
    if failures:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failures))
    with open(os.path.join(OBJ_DIR, "flags.txt"), "w") as f:
        f.write(_flags_stamp())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + proc.stdout + proc.stderr)
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
