#!/usr/bin/env python3
"""Timing probe (no data-flow bookkeeping): the relation step with its backbone cut by STAGE instead of by frame --
[head of batch k] beside [stem .. layer3[:cut] of batch k+2, both frames in one chain] beside [layer3[cut:] of batch k+1, both
frames] -- against the product's schedule [head | frame 0 | frame 1].  Three chains either way; the stage cut runs 2-frame
kernels (fuller launches, half as many, Winograd-domain filters read once per pair of frames).
  tools/stage_split_probe.py [cut ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

import bench  # noqa: E402
from i2vsgg_amd import ops, train  # noqa: E402

DEV = torch.device("cuda:0")
cuts = [int(a) for a in sys.argv[1:]] or [6, 7, 8]
net = train.build_sgg_net(101, device=DEV)
step = train.SGGEmbStep(net, 2, seed=1, device=DEV)
assert step.capture(warmup=2), step.graph_error
base_ms = 1e3 * bench.timed_steps(step, 5, 40, DEV) / 40
print("product schedule [head | frame 0 | frame 1]: %.3f ms per step" % base_ms)
fs = step.shapes[step._staged]
base = net.RCNN_base
sA, sB = step._frame_streams
for cut in cuts:
    with torch.no_grad():
        mid_shape = base.forward_front(fs.im, cut).shape
    mk = lambda: torch.zeros(mid_shape, device=DEV).contiguous(memory_format=torch.channels_last)
    mid_next, mid_cur = mk(), mk()
    ctxA, ctxB = ops.LaunchContext(DEV), ops.LaunchContext(DEV)
    for _ in range(2):
        with ctxA, torch.no_grad():
            base.forward_front(fs.im, cut, out=mid_next)
        ctxA.fit()
        with ctxB, torch.no_grad():
            base.forward_back(mid_cur, cut, out=step._fmap_dst(fs))
        ctxB.fit()
    torch.cuda.synchronize()

    def body():
        main = torch.cuda.current_stream(DEV)
        step._rotate()
        step.fmap_head_flat.copy_(step.fmap_flat)
        mid_cur.copy_(mid_next)
        with ops.branch(sA, main), ctxA, torch.no_grad():
            base.forward_front(fs.im, cut, out=mid_next)
        with ops.branch(sB, main), ctxB, torch.no_grad():
            base.forward_back(mid_cur, cut, out=step._fmap_dst(fs))
        step._head()
        ops.join(main, sA, sB)

    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        body()
    ms = 1e3 * bench.timed_steps(lambda: train.replay_graph(g, DEV), 5, 40, DEV) / 40
    print("stage split, cut %2d: %.3f ms per step (%.1f frames/s)   loss %.6f" % (cut, ms, 2e3 / ms, float(step.loss)))
    del g
# four chains: head | stem .. layer3[:c1] | layer3[c1:c2] | layer3[c2:]
sC = ops.role_stream(DEV, "side")
for c1, c2 in ((3, 13), (4, 13), (5, 14)):
    with torch.no_grad():
        m1 = base.forward_front(fs.im, c1)
    mk = lambda t: torch.zeros(t.shape, device=DEV).contiguous(memory_format=torch.channels_last)
    a_next, a_cur, b_next, b_cur = mk(m1), mk(m1), mk(m1), mk(m1)
    ctxs = [ops.LaunchContext(DEV) for _ in range(3)]
    parts = [lambda: base.forward_front(fs.im, c1, out=a_next), lambda: base.forward_back(a_cur, c1, out=b_next, end=c2),
             lambda: base.forward_back(b_cur, c2, out=step._fmap_dst(fs))]
    for _ in range(2):
        for ctx, part in zip(ctxs, parts):
            with ctx, torch.no_grad():
                part()
            ctx.fit()
    torch.cuda.synchronize()

    def body4():
        main = torch.cuda.current_stream(DEV)
        step._rotate()
        step.fmap_head_flat.copy_(step.fmap_flat)
        a_cur.copy_(a_next); b_cur.copy_(b_next)
        for st, ctx, part in zip((sA, sB, sC), ctxs, parts):
            with ops.branch(st, main), ctx, torch.no_grad():
                part()
        step._head()
        ops.join(main, sA, sB, sC)

    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        body4()
    ms = 1e3 * bench.timed_steps(lambda: train.replay_graph(g, DEV), 5, 40, DEV) / 40
    print("four chains, cuts %d / %d: %.3f ms per step (%.1f frames/s)" % (c1, c2, ms, 2e3 / ms))
    del g
step.opt.unfuse()
