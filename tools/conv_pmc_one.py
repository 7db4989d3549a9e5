#!/usr/bin/env python3
"""One conv shape, N launches -- the target of `rocprofv3 --pmc ...` runs (tools/conv_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from i2vsgg_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
B = 2
cin, cout, k, s, p = {"c2": (256, 256, 3, 1, 1), "c1": (1024, 256, 1, 1, 0), "c3": (256, 1024, 1, 1, 0)}[which]
x = torch.randn(B, cin, 38, 63, device="cuda").contiguous(memory_format=torch.channels_last)
w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
for _ in range(10):
    ops.conv2d(x, w, None, None, None, s, p)
torch.cuda.synchronize()
