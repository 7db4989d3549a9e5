#!/usr/bin/env python3
"""Which kernels of the relation step are not bit-reproducible?  Two runs from equal weights on the same batch:
(1) eager, unfused update: the logits and every parameter gradient of step 1 compared bit for bit;
(2) captured / overlapped step, 5 steps: losses and every vrd parameter compared bit for bit.
    python tools/determinism_probe.py [full|small]"""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from i2vsgg_amd import train  # noqa: E402
from i2vsgg_amd.model.utils import config as c  # noqa: E402

DEV = "cuda:0"
full = (sys.argv[1] if len(sys.argv) > 1 else "full") == "full"
c.cfg_from_file(c.default_cfg_file("res101"))
kw = dict(h=600, w=1000, n_boxes=32, n_pairs=32) if full else dict(h=200, w=320, n_boxes=6, n_pairs=5)
layers = 101 if full else 50


def make(graph, fuse):
    net = train.build_sgg_net(layers=layers, seed=5, device=DEV)
    net.vrd.dropout = False
    step = train.SGGEmbStep(net, 2, seed=3, device=DEV, fuse_sgd=fuse, use_graph=graph, **kw)
    return net, step


def eager_once():
    net, step = make(False, False)
    got = {}
    fwd = net.vrd.forward_device

    def spy(*a, **k):
        out = fwd(*a, **k)
        got["score"] = out[0].detach().clone()
        got["feat"] = out[1].detach().clone()
        return out
    net.vrd.forward_device = spy
    loss = step().clone()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if n.startswith("vrd.") and p.grad is not None}
    step.opt.unfuse()
    return loss, got, grads


def trajectory(graph, overlap, fuse, n=5):
    net = train.build_sgg_net(layers=layers, seed=5, device=DEV)
    net.vrd.dropout = False
    step = train.SGGEmbStep(net, 2, seed=3, device=DEV, fuse_sgd=fuse, use_graph=graph, overlap=overlap, **kw)
    if graph:
        assert step.capture(warmup=1, restore=True), step.graph_error
    losses = [step().clone() for _ in range(n)]
    step.opt.flush_pending()
    torch.cuda.synchronize()
    w = {k: v.detach().clone() for k, v in net.named_parameters() if k.startswith("vrd.")}
    step.opt.unfuse()
    return torch.stack(losses), w


MODES = {"eu": ("eager, unfused update", (False, False, False)), "ef": ("eager, fused update", (False, False, True)),
         "sf": ("sequential graph, fused", (True, False, True)), "of": ("overlapped graph, fused", (True, True, True)),
         "ou": ("overlapped graph, unfused", (True, True, False))}
order = sys.argv[2].split(",") if len(sys.argv) > 2 else ["eu", "ef", "sf", "of", "ou"]
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for key in order:
    name, (graph, overlap, fuse) = MODES[key]
    print("== %s, %d steps: runs A, B, C" % (name, nsteps))
    runs = [trajectory(graph, overlap, fuse, nsteps) for _ in range(3)]
    for a, b in ((0, 1), (1, 2), (0, 2)):
        (la, wa), (lb, wb) = runs[a], runs[b]
        bad = [(n, (wa[n] - wb[n]).abs().max().item() / max(wa[n].abs().max().item(), 1e-30)) for n in wa if not torch.equal(wa[n], wb[n])]
        print(" %s vs %s: losses equal %s; %d of %d tensors differ %s" % ("ABC"[a], "ABC"[b], torch.equal(la, lb), len(bad), len(wa),
                                                                       " ".join("%s:%.1e" % (n.replace("vrd.", ""), d) for n, d in bad[:8])))
