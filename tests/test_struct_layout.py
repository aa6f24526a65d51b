"""The ctypes mirrors in _native.py against the C structs of include/*.h: same field names in the same order, same
offsets, same size -- checked by compiling a small C program with the headers (host compiler only: runs without a GPU)."""
import importlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = {  # C struct -> (header, ctypes mirror)
    "mzmcts_config": ("mzmcts.h", "MzConfig"), "mzmcts_root_stats": ("mzmcts.h", "MzRootStats"),
    "mzmcts_profile": ("mzmcts.h", "MzProfile"), "mzmcts_fc_desc": ("mzmcts.h", "MzFcDesc"),
    "mzmcts_head_desc": ("mzmcts.h", "MzHeadDesc"), "mzmcts_tower_layer": ("mzmcts.h", "MzTowerLayer"),
    "mzmcts_tower_gather": ("mzmcts.h", "MzTowerGather"), "mzhist_moves": ("mzhist.h", "MzHistMoves"),
    "mztrain_loss_args": ("mztrain.h", "MzTrainLossArgs"), "mzreplay_config": ("mzreplay.h", "replay_buffer.MzReplayConfig"),
}


def c_fields(header, struct):
    text = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
    body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (struct, struct), text, flags=re.S).group(1)
    names = []
    for statement in body.split(";"):
        statement = statement.strip()
        if not statement:
            continue
        for declarator in statement.split(","):
            names.append(re.findall(r"[A-Za-z_]\w*", re.sub(r"\[.*?\]", "", declarator))[-1])
    return names


def test_ctypes_structs_match_the_headers(pkg, tmp_path):
    native = importlib.import_module("muzero-hypermodel_amd._native")
    lines = ["#include <stddef.h>", "#include <stdio.h>"] + [f'#include "{h}"' for h in sorted({h for h, _ in PAIRS.values()})]
    lines.append("int main(void) {")
    fields = {}
    for struct, (header, _) in PAIRS.items():
        fields[struct] = c_fields(header, struct)
        lines.append(f'    printf("{struct} %zu", sizeof({struct}));')
        for f in fields[struct]:
            lines.append(f'    printf(" %zu", offsetof({struct}, {f}));')
        lines.append('    printf("\\n");')
    lines += ["    return 0;", "}"]
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text("\n".join(lines))
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    seen = 0
    for row in subprocess.check_output([str(exe)], text=True).splitlines():
        struct, size, *offsets = row.split()
        where, _, name = PAIRS[struct][1].rpartition(".")
        mirror = getattr(importlib.import_module("muzero-hypermodel_amd." + where) if where else native, name)
        assert [name for name, *_ in mirror._fields_] == fields[struct], struct
        assert native.ctypes.sizeof(mirror) == int(size), struct
        assert [getattr(mirror, f).offset for f in fields[struct]] == [int(o) for o in offsets], struct
        seen += 1
    assert seen == len(PAIRS)
