#!/usr/bin/env python3
"""Where a conv_gemm_f32 launch spends its time (diagnostic instantiation with in-kernel stamps): per workgroup start /
end on the 100 MHz clock, prologue / K loop / epilogue in shader cycles, the CU it ran on.

usage: gemm_phase.py [M K N res]...   (default: the layer3 shapes of two 600x1000 frames)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import torch
from i2vsgg_amd import ops, _lib

shapes = [(4788, 256, 1024, 1), (4788, 1024, 256, 0), (18750, 128, 512, 1), (75000, 64, 256, 1)]
if len(sys.argv) > 4:
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)]
dev = "cuda:0"
for M, K, N, res in shapes:
    x = torch.randn(M, K, 1, 1, device=dev)
    w = (torch.randn(N, K, 1, 1, device=dev) * 0.05)
    sc, sh = torch.rand(N, device=dev) + 0.5, torch.rand(N, device=dev)
    r = torch.randn(M, N, 1, 1, device=dev) if res else None
    f = lambda: ops.conv2d(x, w, sc, sh, r, 1, 0, relu=True)
    for _ in range(200):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record()
    torch.cuda.synchronize()
    t_plain = e0.elapsed_time(e1) / 50 * 1e3
    buf = torch.zeros(8 * 65536, dtype=torch.int64, device=dev)
    _lib.lib.i2v_conv_debug_clock(buf.data_ptr())
    f()
    _lib.lib.i2v_conv_debug_clock(None)
    torch.cuda.synchronize()
    v = buf.view(-1, 8).cpu()
    v = v[v[:, 7] == 1]
    st, en = (v[:, 0] - v[:, 0].min()).double() / 100.0, (v[:, 1] - v[:, 0].min()).double() / 100.0      # us
    med = lambda t: float(t.double().median())
    print("M%d K%d N%d res%d: %.1f us per launch (events, back to back); stamped launch: %d workgroups, span %.1f us" % (
        M, K, N, res, t_plain, v.shape[0], float(en.max())))
    print("   per workgroup (median): start %.1f us, life %.1f us; prologue %.0f cycles, K loop %.0f, epilogue %.0f" % (
        med(st), med(en - st), med(v[:, 2]), med(v[:, 3]), med(v[:, 4])))
    # per-CU residency: (xcc, se/sh/cu bits of HW_ID)
    cu = collections.defaultdict(list)
    for i in range(v.shape[0]):
        cu[(int(v[i, 6]) & 0xF, (int(v[i, 5]) >> 8) & 0xFF)].append((float(st[i]), float(en[i])))
    n_per = sorted(len(x) for x in cu.values())
    print("   CUs used %d; workgroups per CU min %d median %d max %d" % (len(cu), n_per[0], n_per[len(n_per) // 2], n_per[-1]))
    q = [0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0]
    print("   start times (us) quantiles:", " ".join("%.1f" % float(st.quantile(t)) for t in q))
    print("   end   times (us) quantiles:", " ".join("%.1f" % float(en.quantile(t)) for t in q))
