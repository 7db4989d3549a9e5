#!/usr/bin/env python3
"""Filter gradient of the 3x3 bottleneck layers, 8 frames of 600x1000: direct kernel vs Winograd F(4x4,3x3) form."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2vsgg_amd  # noqa: F401,E402
import torch  # noqa: E402

from i2vsgg_amd import ops  # noqa: E402

DEV = "cuda:0"
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
for name, B, C, H, W in [("l1 conv2 64", 8, 64, 150, 250), ("l2 conv2 128", 8, 128, 75, 125), ("l3 conv2 256", 8, 256, 38, 63),
                         ("l4 conv2 512 (256 ROI, 7x7)", 256, 512, 7, 7), ("l4 conv2 512 (256 ROI, 4x4)", 256, 512, 4, 4),
                         ("RPN 1024->512 (4 frames)", 4, 1024, 38, 63)]:
    x, g = cl(torch.randn(B, C, H, W, device=DEV)), cl(torch.randn(B, C if 'RPN' not in name else 512, H, W, device=DEV))
    fl = 2.0 * B * H * W * (C if 'RPN' not in name else 512) * 9 * C
    out = []
    for wino in (False, True):
        for _ in range(3):
            ops._conv_wgrad_raw(x, g, (C if 'RPN' not in name else 512, C, 3, 3), 1, 1, winograd=wino)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops._conv_wgrad_raw(x, g, (C if 'RPN' not in name else 512, C, 3, 3), 1, 1, winograd=wino)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e3
        out.append("%s %7.1f us (%5.1f TF alg)" % ("winograd" if wino else "direct  ", t, fl / t / 1e6))
    print("%-28s %6.1f GF | %s | %s" % (name, fl / 1e9, out[0], out[1]), flush=True)
