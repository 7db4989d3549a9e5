#!/usr/bin/env python3
"""The filter gradient of the pointwise layers (conv_wgrad2_f32) on the shapes round 5's review names -- layer3's 1x1 layers at 8
frames of 600x1000 (M = 8 x 38 x 63 pixels) and their neighbours -- with the stage tiles through registers (I2V_TUNE_WGRAD_DMA = 0,
rounds 2-5) and by LDS-DMA (1, round 6): time alone (HIP events around `reps` back-to-back launches behind a blocker GEMM, best of
5) and whether the two forms give the same bits (ordered sums: the split's partial filters are added in split order)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from i2vsgg_amd import ops  # noqa: E402
from i2vsgg_amd._lib import TUNE, lib  # noqa: E402

DEV = "cuda:0"
blocker = torch.randn(8192, 8192, device=DEV)


def alone(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.mm(blocker, blocker)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


# (name, frames, H, W, Cin, Cout)
cases = [("layer3 conv1 (1024 -> 256), 8 frames", 8, 38, 63, 1024, 256), ("layer3 conv3 (256 -> 1024), 8 frames", 8, 38, 63, 256, 1024),
         ("layer3 conv1, 4 frames", 4, 38, 63, 1024, 256), ("layer2 conv1 (512 -> 128), 8 frames", 8, 75, 125, 512, 128),
         ("layer2 conv3 (128 -> 512), 8 frames", 8, 75, 125, 128, 512), ("layer1 conv3 (64 -> 256), 8 frames", 8, 150, 250, 64, 256),
         ("layer4 conv1 (2048 -> 512), 256 rois 7x7", 256, 7, 7, 2048, 512)]
only = sys.argv[1:]
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
for name, B, H, W, C, N in cases:
    if only and not any(o in name for o in only):
        continue
    x, g = cl(torch.rand(B, C, H, W, device=DEV) * 2 - 1), cl(torch.rand(B, N, H, W, device=DEV) * 2 - 1)
    fl = 2.0 * B * H * W * C * N
    out = {}
    print("== %s  M %d (%.2f GFLOP)" % (name, B * H * W, fl * 1e-9), flush=True)
    for tiles in ((1, 2, 3) if os.environ.get("WGRAD_TILES") else ()):      # the larger tiles (I2V_TUNE_WGRAD_V2 = 2: 128x128, 3: 128x64)
        lib.i2v_set_tuning(TUNE["I2V_WGRAD_V2"], tiles)
        for dma in (0, 1):
            lib.i2v_set_tuning(TUNE["I2V_WGRAD_DMA"], dma)
            ctx = ops.LaunchContext(DEV, ordered=True)

            def fn2():
                with ctx, torch.no_grad():
                    return ops._conv_wgrad_raw(x, g, (N, C, 1, 1), 1, 0)
            t = alone(fn2, 20)
            print("   tiles %s ordered staging %-9s %8.2f us %6.1f TF" % ({1: "64x64", 2: "128x128", 3: "128x64"}[tiles],
                                                                         "LDS-DMA" if dma else "registers", t, fl / t / 1e6), flush=True)
    lib.i2v_set_tuning(TUNE["I2V_WGRAD_V2"], 1)
    for ordered in (False, True):
        ctx = ops.LaunchContext(DEV, ordered=ordered)
        for dma in (0, 1):
            assert lib.i2v_set_tuning(TUNE["I2V_WGRAD_DMA"], dma) == 0

            def fn():
                with ctx, torch.no_grad():
                    return ops._conv_wgrad_raw(x, g, (N, C, 1, 1), 1, 0)
            out[(ordered, dma)] = fn().clone()
            t = alone(fn, 20)
            print("   %-8s staging %-9s %8.2f us %6.1f TF" % ("ordered" if ordered else "atomics", "LDS-DMA" if dma else "registers",
                                                             t, fl / t / 1e6), flush=True)
        lib.i2v_set_tuning(TUNE["I2V_WGRAD_DMA"], 1)
    want = torch.nn.grad.conv2d_weight(x.double(), (N, C, 1, 1), g.double(), 1, 0)
    sc = float(want.abs().max())
    print("   ordered: DMA == registers bit for bit: %s; max |err| vs float64 torch / max |want|: registers %.2e, DMA %.2e" % (
        torch.equal(out[(True, 0)], out[(True, 1)]), float((out[(True, 0)].double() - want).abs().max()) / sc,
        float((out[(True, 1)].double() - want).abs().max()) / sc), flush=True)
