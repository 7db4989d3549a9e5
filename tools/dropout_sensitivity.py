"""how far does ONE head forward of the full-size relation step move its loss when only the dropout mask
changes (same weights, same minibatch, same features)?  The round-2 outlier was a final loss off by 5.9e-5 with weights equal
in their abs-sum; every data input of that test is constant from step to step, so the step-varying inputs of the last head
forward are the weights, the RNG offset of the dropout kernels and the zero state of the arenas."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import numpy as np
import torch
from i2vsgg_amd import ops, train

dev = "cuda:0"
net = train.build_sgg_net(101, device=dev)
step = train.SGGEmbStep(net, 2, seed=1, device=dev, use_graph=False)
for _ in range(20):                       # the trajectory length of the test
    step()
step.opt.flush_pending()
fs = step.shapes[step._staged]
losses = []
with torch.no_grad():
    for seed in range(32):
        torch.manual_seed(1000 + seed)
        score, _ = net.vrd.forward_device(step.fmap, step.boxes, step.relb, step.masks, step.ixs, step.ixo)
        losses.append(float(ops.bce_rows(score, step.labels, step.wrow)))
l = np.array(losses)
print("loss over 32 dropout masks, fixed weights: mean %.7f  std %.2e  max |dev| %.2e" % (l.mean(), l.std(), np.abs(l - l.mean()).max()))
print("pairwise |difference| median %.2e" % np.median(np.abs(l[:, None] - l[None, :])[np.triu_indices(32, 1)]))
torch.manual_seed(5)
a = float(ops.bce_rows(net.vrd.forward_device(step.fmap, step.boxes, step.relb, step.masks, step.ixs, step.ixo)[0].detach(), step.labels, step.wrow))
torch.manual_seed(5)
b = float(ops.bce_rows(net.vrd.forward_device(step.fmap, step.boxes, step.relb, step.masks, step.ixs, step.ixo)[0].detach(), step.labels, step.wrow))
print("same mask twice: %.9f %.9f (|diff| %.1e: the atomics of the split-K GEMMs)" % (a, b, abs(a - b)))
step.opt.unfuse()
