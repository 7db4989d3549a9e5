#!/usr/bin/env python3
"""Intra-workgroup K split (conv_gemm_f32<.., KG = 4>, I2V_KGROUPS) against the memory-side split-K it replaces, on the
pointwise GEMMs of the backbone that split over K: layer3 conv1 (K 1024 -> N 256) for one frame and a frame pair, its
data-gradient shape, layer4's conv1 shapes.  One HIP-event pair around 50 back-to-back launches behind a blocker GEMM."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

from i2vsgg_amd import ops  # noqa: E402
from i2vsgg_amd._lib import TUNE, lib  # noqa: E402

DEV = "cuda:0"
blocker = torch.randn(8192, 8192, device=DEV)


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.mm(blocker, blocker)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


cases = [("layer3 conv1, 2 frames", 4788, 1024, 256, False), ("layer3 conv1, 1 frame", 2394, 1024, 256, False),
         ("layer3 conv1, 4 frames", 9576, 1024, 256, False), ("layer3 conv3 (no split), 2 frames", 4788, 256, 1024, True),
         ("layer4 conv1 (32 ROI x 7x7)", 1568, 1024, 512, False), ("layer2 conv1, 2 frames", 18750, 512, 128, False)]
for name, M, K, N, res in cases:
    x = torch.randn(M, K, 1, 1, device=DEV)
    w = torch.randn(N, K, 1, 1, device=DEV) / K ** 0.5
    sc, sh = torch.rand(N, device=DEV) + 0.5, torch.rand(N, device=DEV)
    r = torch.randn(M, N, 1, 1, device=DEV) if res else None
    row, outs = [], []
    for kg in (4, 2, 0):
        lib.i2v_set_tuning(TUNE["I2V_KGROUPS"], kg)
        us = timeit(lambda: ops.conv2d(x, w, sc, sh, r, 1, 0, relu=True))
        outs.append(ops.conv2d(x, w, sc, sh, r, 1, 0, relu=True).clone())
        ws = lib.i2v_conv_split_workspace_bytes(1, 1, M, K, N, 1, 1, 1, 0)
        row.append((us, ws))
    lib.i2v_set_tuning(TUNE["I2V_KGROUPS"], 0)
    fl = 2.0 * M * N * K
    err = max(float((o - outs[2]).abs().max()) for o in outs[:2]) / float(outs[2].abs().max())
    print("%-36s M %5d K %4d N %4d: 4 K groups %6.1f us (%5.1f TF) | 2 K groups %6.1f us (%5.1f TF) | memory split %6.1f us (%5.1f TF, ws %5.1f MB)  max dev %.1e" % (
        name, M, K, N, row[0][0], fl / row[0][0] / 1e6, row[1][0], fl / row[1][0] / 1e6, row[2][0], fl / row[2][0] / 1e6, row[2][1] / 1e6, err))
