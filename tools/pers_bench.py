#!/usr/bin/env python3
"""conv_gemm_f32 (one tile per workgroup) against conv_gemm_pers_f32 (tile queue, stream of stages) on the pointwise layers of
the backbone: time per launch for every tile shape, and bit-equality of the outputs.   B=2 python tools/pers_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from i2vsgg_amd import ops, _lib

B = int(os.environ.get("B", "2"))
SHAPES = [  # name, Cin, H, W, Cout, residual
    ("l1 c1 64->64", 64, 150, 250, 64, False),
    ("l1 c3 64->256 +res", 64, 150, 250, 256, True),
    ("l1 c1 256->64", 256, 150, 250, 64, False),
    ("l2 c3 128->512 +res", 128, 75, 125, 512, True),
    ("l2 c1 512->128", 512, 75, 125, 128, False),
    ("l3 c3 256->1024 +res", 256, 38, 63, 1024, True),
    ("l3 c1 1024->256", 1024, 38, 63, 256, False),
]
TILES = ["128x128", "128x64", "96x64", "80x64", "64x64", "32x64"]
dev = "cuda:0"
lib = _lib.lib
PERSIST = _lib.TUNE["I2V_GEMM_PERSIST"]


def run(x, wt, sc, sh, res):
    return ops.conv2d(x, wt, sc, sh, res, 1, 0, relu=True)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


print("%-22s %6s | " % ("shape", "GFLOP") + " ".join("%13s" % t for t in TILES) + " | auto      (us one-tile / us persistent)")
for name, cin, h, w, cout, with_res in SHAPES:
    x = torch.randn(B, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 1, 1, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.rand(cout, device=dev)
    res = torch.randn(B, cout, h, w, device=dev).contiguous(memory_format=torch.channels_last) if with_res else None
    fl = 2.0 * B * h * w * cout * cin
    cells = []
    for cfg in list(range(len(TILES))) + [-1]:
        lib.i2v_conv_set_tile(cfg if cfg >= 0 else 0xFF)
        lib.i2v_set_tuning(PERSIST, 0)
        y0 = run(x, wt, sc, sh, res).clone()
        t0 = timed(lambda: run(x, wt, sc, sh, res))
        lib.i2v_set_tuning(PERSIST, int(os.environ.get("PER", "2")))
        y1 = run(x, wt, sc, sh, res).clone()
        t1 = timed(lambda: run(x, wt, sc, sh, res))
        eq = torch.equal(y0, y1)
        cells.append("%5.1f/%5.1f%s" % (t0 * 1e6, t1 * 1e6, " " if eq else "!"))
    lib.i2v_conv_set_tile(-1)
    lib.i2v_set_tuning(PERSIST, 0)
    print("%-22s %6.2f | " % (name, fl / 1e9) + " ".join("%13s" % c for c in cells[:-1]) + " | " + cells[-1], flush=True)
print("'!' = outputs differ")
