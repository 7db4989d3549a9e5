#!/usr/bin/env python3
"""vrd.fc6's fused filter gradient + SGD(momentum) update (i2v_conv_wgrad_sgd: 128 rows, 50176 -> 4096, 822 MB of filter and
as much momentum read AND written = 3.29 GB per call) per filters-per-workgroup setting, alone on the chip; and fc7's.

    python tools/fc6_update_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from i2vsgg_amd import _lib  # noqa: E402
from i2vsgg_amd._lib import lib, ptr  # noqa: E402

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for name, M, K, N in (("fc6", 128, 50176, 4096), ("fc7", 128, 4096, 4096)):
    x = torch.randn(M, K, device=dev)
    g = torch.randn(M, N, device=dev) * 1e-3
    w = torch.randn(N, K, device=dev) * 0.01
    m = torch.zeros_like(w)
    for tile in (128, 64):
        lib.i2v_set_tuning(_lib.TUNE["I2V_WGRAD_FUSED_TILE"], tile)

        def call():
            rc = lib.i2v_conv_wgrad_sgd(ptr(x), ptr(g), ptr(w), ptr(m), M, 1, 1, K, N, 1, 1, 1, 0, 1e-4, 0.9, 5e-4, st)
            assert rc == 0, _lib.last_error()

        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        print("%s update, %3d filters per workgroup: %7.1f us  %.2f TB/s of filter + momentum traffic  %.0f TF" % (
            name, tile, t * 1e6, 16.0 * N * K / t / 1e12, 2.0 * M * N * K / t / 1e12))
    lib.i2v_set_tuning(_lib.TUNE["I2V_WGRAD_FUSED_TILE"], 128)
