#!/usr/bin/env python3
"""Duration of the matrix-core board convolution (include/mzmcts.h mzmcts_board_conv3x3, epilogue fused) against the
path it replaces (torch / MIOpen convolution + mzmcts_affine_act) on the residual networks' shapes.
    python tools/conv_bench.py [batch:cin:cout:h:w ...]      one JSON line per shape
"""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
models = importlib.import_module("muzero-hypermodel_amd.models")

def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps

specs = [a for a in sys.argv[1:] if ":" in a] or ["1024:64:64:6:7", "4096:64:64:6:7", "1024:65:64:6:7", "4096:16:16:3:3", "1024:16:16:6:6", "4096:16:16:6:6"]
for spec in specs:
    b, cin, cout, h, w = (int(v) for v in spec.split(":"))
    conv = models.conv3x3(cin, cout).cuda().eval()
    bn = models.BatchNorm2d(cout).cuda().eval()
    x = torch.randn(b, cin, h, w, device="cuda")
    res = torch.randn(b, cout, h, w, device="cuda")
    with torch.no_grad():
        os.environ["MZ_BOARD_CONV"] = "all"
        assert conv.takes_mfma_path(x)
        t_fused = timed(lambda: conv.fused(x, bn, residual=res))
        os.environ["MZ_BOARD_CONV"] = "off"
        t_torch_conv = timed(lambda: torch.nn.Conv2d.forward(conv, x))
        t_module = timed(lambda: models.conv_epilogue(conv(x), bn, residual=res))
    flops = 2.0 * b * h * w * cout * cin * 9
    print(json.dumps({"shape": spec, "fused_mfma_us": t_fused, "fused_TFLOPs": flops / t_fused / 1e6,
                      "frac_of_fp32_matrix_peak": flops / t_fused / 1e6 / 157.3,
                      "miopen_conv_only_us": t_torch_conv, "previous_module_path_us": t_module}), flush=True)
