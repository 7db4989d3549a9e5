#!/usr/bin/env python3
"""Per-shape timing of conv_igemm_f32 for every tile shape (interleaved rounds in one process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from i2vsgg_amd import ops, _lib

B = int(os.environ.get("B", "2"))
SHAPES = [  # name, Cin, H, W, Cout, K, stride, pad
    ("stem 7x7s2", 4, 600, 1000, 64, 7, 2, 3),
    ("l1 c1 64->64", 64, 150, 250, 64, 1, 1, 0),
    ("l1 c2 3x3 64", 64, 150, 250, 64, 3, 1, 1),
    ("l1 c3 64->256", 64, 150, 250, 256, 1, 1, 0),
    ("l1 c1 256->64", 256, 150, 250, 64, 1, 1, 0),
    ("l2 c1 256->128 s2", 256, 150, 250, 128, 1, 2, 0),
    ("l2 c2 3x3 128", 128, 75, 125, 128, 3, 1, 1),
    ("l2 c3 128->512", 128, 75, 125, 512, 1, 1, 0),
    ("l2 c1 512->128", 512, 75, 125, 128, 1, 1, 0),
    ("l3 c1 512->256 s2", 512, 75, 125, 256, 1, 2, 0),
    ("l3 ds 512->1024 s2", 512, 75, 125, 1024, 1, 2, 0),
    ("l3 c2 3x3 256", 256, 38, 63, 256, 3, 1, 1),
    ("l3 c3 256->1024", 256, 38, 63, 1024, 1, 1, 0),
    ("l3 c1 1024->256", 1024, 38, 63, 256, 1, 1, 0),
]
if os.environ.get("GEMM"):
    B = 1
    SHAPES = [("gemm 4096^3", 4096, 64, 64, 4096, 1, 1, 0), ("gemm 8192x2048x2048", 2048, 64, 128, 2048, 1, 1, 0),
              ("gemm 16384x1024x1024", 1024, 128, 128, 1024, 1, 1, 0)]
TILES = ["128x128", "128x64", "96x64", "80x64", "64x64", "32x64"]
dev = "cuda:0"
print("%-20s %9s | " % ("shape", "GFLOP") + " ".join("%9s" % t for t in TILES) + " |  auto")
for name, cin, h, w, cout, k, s, p in SHAPES:
    x = torch.randn(B, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    sc = torch.rand(cout, device=dev) + 0.5
    sh = torch.rand(cout, device=dev)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    fl = 2.0 * B * ho * wo * cout * k * k * cin
    res = []
    SP = int(os.environ.get("SPEC", "0"))
    for cfg in list(range(len(TILES))) + [-1]:
        _lib.lib.i2v_conv_set_tile((cfg if cfg >= 0 else 0xFF) | (SP << 8))
        for _ in range(2):
            ops.conv2d(x, wt, sc, sh, None, s, p, relu=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            ops.conv2d(x, wt, sc, sh, None, s, p, relu=True)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e-3)
    _lib.lib.i2v_conv_set_tile(-1)
    print("%-20s %9.2f | " % (name, fl / 1e9) + " ".join("%5.0fus/%3.0f" % (t * 1e6, fl / t / 1e12) for t in res[:len(TILES)])
          + " | %5.0fus/%3.0fTF" % (res[-1] * 1e6, fl / res[-1] / 1e12))
