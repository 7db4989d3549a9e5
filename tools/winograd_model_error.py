#!/usr/bin/env python3
"""End-to-end effect of the Winograd 3x3 layers on the ResNet-101 C4 feature map and the relation logits (full size)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from i2vsgg_amd import train
from i2vsgg_amd.model.faster_rcnn import layers
from i2vsgg_amd.model.utils import config as c
c.cfg_from_file(c.default_cfg_file("res101"))
net = train.build_sgg_net(101, device="cuda:0")
net.vrd.dropout = False
step = train.SGGEmbStep(net, 2, seed=1, device="cuda:0", use_graph=False)
out = {}
for mode, mincin in ((0, 1 << 30), (2, 128), (4, 64)):
    layers.WINOGRAD, layers.WINOGRAD_MIN_CIN = mode, mincin
    for m in net.modules():
        if hasattr(m, "_wino_key"):
            m._wino_key = None
    with torch.no_grad():
        fmap = net.RCNN_base(step.im)
        score, _ = net.vrd.forward_device(fmap, step.boxes, step.relb, step.masks, step.ixs, step.ixo)
    out[mode] = (fmap.double(), score.double())
f0, s0 = out[0]
for mode in (2, 4):
    f, s = out[mode]
    print("I2V_WINOGRAD=%d vs direct: feature map max|d|/max|ref| = %.2e, relation logits max|d| = %.2e (logits in [-1,1])" % (
        mode, ((f - f0).abs().max() / f0.abs().max()).item(), (s - s0).abs().max().item()))
