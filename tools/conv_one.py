#!/usr/bin/env python3
"""Run one conv shape a few times (for rocprofv3 --pmc passes).  Usage: conv_one.py Cin H W Cout K stride pad [tilecfg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from i2vsgg_amd import ops, _lib
cin, h, w, cout, k, s, p = (int(v) for v in sys.argv[1:8])
cfg = int(sys.argv[8]) if len(sys.argv) > 8 else -1
B = int(os.environ.get("B", "2"))
x = torch.randn(B, cin, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
wt = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
sc = torch.rand(cout, device="cuda") + 0.5
sh = torch.rand(cout, device="cuda")
_lib.lib.i2v_conv_set_tile(cfg)
for _ in range(5):
    ops.conv2d(x, wt, sc, sh, None, s, p, relu=True)
torch.cuda.synchronize()
