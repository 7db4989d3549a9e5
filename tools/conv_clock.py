#!/usr/bin/env python3
"""In-kernel clock of conv_igemm_f32 under sustained load (diagnostic): s_memtime / s_memrealtime per workgroup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from i2vsgg_amd import ops, _lib
B = 2
for name, cin, h, w, cout, k, s, p in (("l3 c2 3x3 256", 256, 38, 63, 256, 3, 1, 1), ("gemm 4096^3 (B=1)", 4096, 64, 32, 4096, 1, 1, 0)):
    x = torch.randn(B, cin, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    buf = torch.zeros(8 * 65536, dtype=torch.int64, device="cuda")
    for _ in range(300):                      # warm: let DVFS settle under load
        ops.conv2d(x, wt, None, None, None, s, p)
    _lib.lib.i2v_conv_debug_clock(buf.data_ptr())
    ops.conv2d(x, wt, None, None, None, s, p)
    _lib.lib.i2v_conv_debug_clock(None)
    torch.cuda.synchronize()
    v = buf.view(-1, 8).cpu()
    v = v[v[:, 1] > 0]
    clk = (v[:, 3].double() / v[:, 1].double() * 100e6)
    print("%-20s workgroups %5d  loop cycles median %8.0f  in-kernel clock median %.2f GHz (min %.2f max %.2f)" % (
        name, v.shape[0], v[:, 0].double().median().item(), clk.median().item() / 1e9, clk.min().item() / 1e9, clk.max().item() / 1e9))
