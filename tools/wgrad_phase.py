#!/usr/bin/env python3
"""Where a conv_wgrad2_f32 launch spends its time (diagnostic instantiation with in-kernel stamps).
usage: wgrad_phase.py   (instance_styleD shapes at 8 frames: layer3 conv1, layer2 conv3, layer1 conv2 3x3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import torch
from i2vsgg_amd import ops, _lib

dev = "cuda:0"
B = 8
shapes = [("l3 conv1 1024->256", 1024, 38, 63, 256, 1, 1, 0), ("l3 conv3 256->1024", 256, 38, 63, 1024, 1, 1, 0),
          ("l2 conv3 128->512", 128, 75, 125, 512, 1, 1, 0), ("l1 conv2 3x3 64", 64, 150, 250, 64, 3, 1, 1),
          ("l3 conv2 3x3 256", 256, 38, 63, 256, 3, 1, 1)]
for name, cin, h, w, cout, k, s, p in shapes:
    x = torch.randn(B, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, cout, (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1, device=dev).contiguous(memory_format=torch.channels_last)
    f = lambda: ops._conv_wgrad_raw(x, g, (cout, cin, k, k), s, p)
    for _ in range(30):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * g.shape[0] * g.shape[2] * g.shape[3] * cout * cin * k * k
    buf = torch.zeros(8 * 65536, dtype=torch.int64, device=dev)
    _lib.lib.i2v_conv_debug_clock(buf.data_ptr())
    abl = int(os.environ.get("ABL", "0"))     # stamped instantiation only: 1 = no staging after the first stage, 2 = no MFMAs, 4 = no barrier
    if abl:
        _lib.lib.i2v_conv_set_tile(0xFF | (abl << 10))
    f()
    _lib.lib.i2v_conv_set_tile(-1)
    _lib.lib.i2v_conv_debug_clock(None)
    torch.cuda.synchronize()
    v = buf.view(-1, 8).cpu()
    v = v[v[:, 7] == 1]
    if v.shape[0] == 0:
        print("%s: %.1f us (%.1f TF); no stamps (kernel form without the diagnostic instantiation)" % (name, t, fl / t / 1e6))
        continue
    st, en = (v[:, 0] - v[:, 0].min()).double() / 100.0, (v[:, 1] - v[:, 0].min()).double() / 100.0
    med = lambda a: float(a.double().median())
    cu = collections.Counter(((int(v[i, 6]) & 0xF, (int(v[i, 5]) >> 8) & 0xFF)) for i in range(v.shape[0]))
    n_per = sorted(cu.values())
    q = [0.0, 0.25, 0.5, 0.75, 1.0]
    print("%s: %.1f us per launch incl. its clear (%.1f TF); %d workgroups, span %.1f us; per CU %d..%d" % (
        name, t, fl / t / 1e6, v.shape[0], float(en.max()), n_per[0], n_per[-1]))
    print("   median per workgroup: life %.1f us, prologue %.0f cycles, pixel loop %.0f, epilogue %.0f; starts %s ends %s" % (
        med(en - st), med(v[:, 2]), med(v[:, 3]), med(v[:, 4]), " ".join("%.1f" % float(st.quantile(t_)) for t_ in q),
        " ".join("%.1f" % float(en.quantile(t_)) for t_ in q)))
    # per XCD: do the workgroups of one XCD run longer (cycles) or slower (cycles per us)?
    xcc = v[:, 6] & 0xF
    rows = []
    for xid in sorted(set(xcc.tolist())):
        sel = xcc == xid
        life_us = (en - st)[sel]
        cyc = (v[sel, 2] + v[sel, 3] + v[sel, 4]).double()
        rows.append("x%d: n %d loop %.0fk end %.1f us %.2f GHz" % (xid, int(sel.sum()), float(v[sel, 3].double().median()) / 1e3,
                                                                  float(en[sel].median()), float((cyc / life_us).median()) / 1e3))
    print("   " + " | ".join(rows))

