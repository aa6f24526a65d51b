"""csrc/glibc_libm.h -- glibc's `log` / `pow` restated for the device-side Dirichlet noise -- against the libm of the
machine the tests run on (the library numpy's legacy gamma sampler calls; reference self_play.py:468-477).

* the committed tables (csrc/glibc_libm_tables.inc) are the ones inside this machine's libm.so.6;
* the host build of the restated functions returns libm's bits on 20 M arguments over the sampler's domain, incl.
  arguments next to 1, subnormal arguments, zero bases and results down to the subnormal range.

The device build of the same header is checked on the GPU by tests/test_gpu_dirichlet.py."""
import importlib.util
import json
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "muzero-hypermodel_amd", "csrc")


def _extractor():
    spec = importlib.util.spec_from_file_location("extract_libm_tables", os.path.join(ROOT, "tools", "extract_libm_tables.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _committed_tables():
    text = open(os.path.join(CSRC, "glibc_libm_tables.inc")).read()
    out = {}
    for m in re.finditer(r"MZ_LIBM_TABLE\((?:double|unsigned long long), (\w+), (\d+)\) = \{(.*?)\};", text, re.S):
        name, n, body = m.group(1), int(m.group(2)), m.group(3)
        vals = [v.strip() for v in body.replace("\n", " ").split(",") if v.strip()]
        assert len(vals) == n, name
        out[name] = [int(v[:-3], 16) for v in vals] if name == "exp_tab" else [float.fromhex(v) for v in vals]
    return out


def test_committed_tables_are_this_machines_libm():
    ex = _extractor()
    path = ex.find_libm()
    if path is None:
        pytest.skip("no libm.so.6 at the usual places")
    live = ex.read_tables(path)
    if live is None:
        pytest.skip("this libm does not hold the ARM optimized-routines tables (not glibc >= 2.28?)")
    committed = _committed_tables()
    assert set(committed) == set(live)
    for name, vals in live.items():
        assert list(vals) == committed[name], f"{name}: this machine's libm differs from glibc_libm_tables.inc"


def test_host_build_returns_libm_bits(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "glibc_libm_check")
    flags = ["-O2", "-std=c++17", "-ffp-contract=off", "-I", CSRC]
    if "fma" in open("/proc/cpuinfo").read():
        flags.append("-mfma")          # (without it __builtin_fma calls libm's fma: same values, slower)
    subprocess.run([gxx] + flags + ["-o", exe, os.path.join(ROOT, "tests", "glibc_libm_check.cpp"), "-lm"], check=True)
    proc = subprocess.run([exe, "20000000"], capture_output=True, text=True)
    report = json.loads(proc.stdout.strip().splitlines()[-1])
    assert proc.returncode == 0 and report["log_mismatches"] == 0 and report["pow_mismatches"] == 0, proc.stdout
    assert report["pow_results_below_1e-200"] > 100000     # the scaled / subnormal tail of pow was exercised


def test_device_dirichlet_sampler_equals_host_sampler_on_cpu(tmp_path):
    """np_legacy_rng.h DeviceStream (what root_noise_kernel runs) built for the host: the same Dirichlet rows, word
    counts and stream states as HostStream on this machine's libm, 50 400 rows over 400 seeds, shapes 0.03 ... 1 and
    2 ... 121 entries (HostStream itself is pinned to numpy's vectors by fixture G7, tests/test_native_abi.py)."""
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "device_stream_check")
    flags = ["-O2", "-std=c++17", "-ffp-contract=off", "-I", CSRC]
    if "fma" in open("/proc/cpuinfo").read():
        flags.append("-mfma")
    subprocess.run([gxx] + flags + ["-o", exe, os.path.join(ROOT, "tests", "device_stream_check.cpp"), "-lm"], check=True)
    proc = subprocess.run([exe], capture_output=True, text=True)
    report = json.loads(proc.stdout.strip().splitlines()[-1])
    assert proc.returncode == 0 and report["mismatches"] == 0 and report["rows"] == 400 * 7 * 6 * 3, proc.stdout
