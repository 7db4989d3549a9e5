#!/usr/bin/env python3
"""Where does a K-loop stage go?  In-kernel cycle stamps (s_memtime) of the main loop of conv_igemm_f32 at ~1
workgroup per CU (layer3 3x3, 80x64 tile, no split-K), with staging or MFMAs ablated.  Diagnostic only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["I2V_SPLIT_BELOW"] = os.environ.get("I2V_SPLIT_BELOW", "0")
import torch
from i2vsgg_amd import ops, _lib
B = 2
TILES = [int(t) for t in os.environ.get("TILES", "3").split(",")]
BOUND = {0: 4096, 1: 2048, 2: 1536, 3: 1280, 4: 1024, 5: 512}   # MFMA issue cycles per 32-deep stage
cases = [("l3 c2 3x3 256 (72 stages)", 256, 38, 63, 256, 3, 1, 1, 72), ("l3 c1 1024->256 (32 stages)", 1024, 38, 63, 256, 1, 1, 0, 32)]
if os.environ.get("CASES") == "c3":         # layer3 conv3: K = 256 (8 stages), N = 1024: 960 workgroups, two rounds
    cases = [("l3 c3 256->1024 (8 stages)", 256, 38, 63, 1024, 1, 1, 0, 8)]
if os.environ.get("CASES") == "smallk":     # the HBM-bound layers: K = 64 / 128, two to four stages per workgroup
    cases = [("l1 c3 64->256 (2 stages)", 64, 150, 250, 256, 1, 1, 0, 2), ("l2 c3 128->512 (4 stages)", 128, 75, 125, 512, 1, 1, 0, 4)]
for name, cin, h, w, cout, k, s, p, stages in cases:
    x = torch.randn(B, cin, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    for tile in TILES:
        for label, cfg in (("plain", tile | (1 << 8)), ("plain, no staging", tile | (1 << 8) | (1 << 10)),
                           ("plain, no MFMA", tile | (1 << 8) | (2 << 10)), ("specialised", tile | (2 << 8)),
                           ("plain, no LDS stores", tile | (1 << 8) | (64 << 10)),
                           ("plain, no global loads", tile | (1 << 8) | (128 << 10)),
                           ("no staging, no barrier", tile | (1 << 8) | (5 << 10)),
                           ("no staging/barrier/ds_read", tile | (1 << 8) | (13 << 10))):
            if tile != 3 and "no " in label:
                continue                       # the ablated variants are only instantiated for the 80x64 tile
            buf = torch.zeros(8 * 65536, dtype=torch.int64, device="cuda")
            _lib.lib.i2v_conv_set_tile(cfg)
            for _ in range(50):
                ops.conv2d(x, wt, None, None, None, s, p)
            _lib.lib.i2v_conv_debug_clock(buf.data_ptr())
            ops.conv2d(x, wt, None, None, None, s, p)
            _lib.lib.i2v_conv_debug_clock(None)
            torch.cuda.synchronize()
            v = buf.view(-1, 8).cpu()
            v = v[v[:, 1] > 0]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv2d(x, wt, None, None, None, s, p)
            e1.record(); torch.cuda.synchronize()
            print("%-28s tile %d %-28s WGs %4d  loop cycles/stage %6.0f (MFMA issue bound %d)  setup %6.0f cyc  kernel %6.1f us" % (
                name, tile, label, v.shape[0], v[:, 0].double().median().item() / stages, BOUND[tile],
                v[:, 2].double().median().item(), e0.elapsed_time(e1) / 20 * 1e3))
            if label.startswith("plain"):
                ghz = (v[:, 3].double() / v[:, 1].double()).median().item() * 0.1
                print("    shader clock during the K loop: %.2f GHz (s_memtime cycles per 100 MHz s_memrealtime tick)" % ghz)
                fin = v[v[:, 5] > 0]
                post = (v[:, 4] - v[:, 3]).double()
                postf = (fin[:, 4] - fin[:, 3]).double()
                print("    after the K loop: all workgroups median %6.0f cyc (max %6.0f); epilogue-running ones median %6.0f (max %6.0f); last workgroup ends at %6.0f" % (
                    post.median().item(), post.max().item(), postf.median().item(), postf.max().item(), v[:, 4].double().max().item()))
            if label == "specialised":
                m = lambda i: v[:, i].double().median().item() / stages
                print("    specialised per stage: loader LDS-store %5.0f  load-issue %5.0f  barrier-wait %5.0f | MFMA wave compute %5.0f" % (m(4), m(5), m(6), m(7)))
    _lib.lib.i2v_conv_set_tile(-1)
