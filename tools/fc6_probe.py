#!/usr/bin/env python3
"""vrd.fc6 forward (M rows x 50176 -> 4096) per tile shape (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from i2vsgg_amd import ops, _lib
TILES = ["128x128", "128x64", "96x64", "80x64", "64x64", "32x64", "auto"]
for M in (128, 1024):
    N = 4096 if M == 128 else 512           # 1024 x 512: the column-parallel shard at 8 GPUs
    x = torch.randn(M, 50176, device="cuda")
    w = torch.randn(N, 50176, device="cuda") * 0.01
    b = torch.zeros(N, device="cuda")
    out = []
    for cfg in list(range(6)) + [-1]:
        _lib.lib.i2v_conv_set_tile(cfg if cfg >= 0 else -1)
        for _ in range(2): y = ops.linear(x, w, b, relu=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): y = ops.linear(x, w, b, relu=True)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5 * 1e-3
        out.append("%s %4.0fus/%3.0fTF" % (TILES[cfg], t * 1e6, 2.0 * M * 50176 * N / t / 1e12))
    _lib.lib.i2v_conv_set_tile(-1)
    print("M %4d N %4d: " % (M, N) + " | ".join(out))
