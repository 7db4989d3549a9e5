#!/usr/bin/env python3
"""Does a layer3 kernel of ONE frame fill the chip?  A chain of 40 launches of the same conv captured into a graph, replayed
alone and as 2 / 3 / 4 parallel branches of one graph (own outputs, own split-K workspace): time per launch of the
ensemble.  Flat time per launch = the kernel alone already owns the machine; halving = it leaves half of it idle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i2vsgg_amd  # noqa: F401,E402
import torch  # noqa: E402

from i2vsgg_amd import _lib, ops  # noqa: E402

DEV = torch.device("cuda:0")
B = int(os.environ.get("B", "1"))
SHAPES = [("l3 c3 256->1024", 256, 38, 63, 1024, 1), ("l3 c1 1024->256", 1024, 38, 63, 256, 1)] if os.environ.get("TILE") else [("l3 c3 256->1024", 256, 38, 63, 1024, 1), ("l3 c1 1024->256", 1024, 38, 63, 256, 1), ("l3 c2 3x3 256 (direct)", 256, 38, 63, 256, 3),
          ("l2 c3 128->512", 128, 75, 125, 512, 1), ("l1 c3 64->256", 64, 150, 250, 256, 1)]
N = 40
if os.environ.get("TILE"):
    _lib.lib.i2v_conv_set_tile(int(os.environ["TILE"]))
main = torch.cuda.Stream()
torch.cuda.set_stream(main)
for name, cin, h, w, cout, k in SHAPES:
    fl = 2.0 * B * h * w * cout * k * k * cin
    out = []
    for nb in (1, 2, 3, 4):
        xs = [torch.randn(B, cin, h, w, device=DEV).contiguous(memory_format=torch.channels_last) for _ in range(nb)]
        wt = (torch.randn(cout, cin, k, k, device=DEV) * 0.05).contiguous(memory_format=torch.channels_last)
        sc, sh = torch.rand(cout, device=DEV) + 0.5, torch.rand(cout, device=DEV)
        ctxs = [ops.LaunchContext(DEV) for _ in range(nb)]
        sts = [torch.cuda.Stream() for _ in range(nb)]
        for x, ctx in zip(xs, ctxs):
            for _ in range(2):
                with ctx:
                    ops.conv2d(x, wt, sc, sh, None, 1, k // 2, relu=True)
            ctx.fit()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cap = torch.cuda.current_stream()
            for x, ctx, st in zip(xs, ctxs, sts):
                st.wait_stream(cap)
                with torch.cuda.stream(st):
                    with ctx:
                        for _ in range(N):
                            ops.conv2d(x, wt, sc, sh, None, 1, k // 2, relu=True)
            for st in sts:
                cap.wait_stream(st)
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5 / N * 1e3           # us per "round" of nb launches
        out.append("%d: %5.1f us/round %5.1f us/launch %5.1f TF" % (nb, t, t / nb, fl * nb / t / 1e6))
        del g
    print("%-24s %6.2f GF | " % (name, fl / 1e9) + " | ".join(out), flush=True)
