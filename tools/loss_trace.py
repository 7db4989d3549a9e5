#!/usr/bin/env python3
"""Per-step loss of the configs[1] step under a given schedule, without syncing inside the loop (races stay exposed).
usage: [I2V_OVERLAP=0|1] [I2V_WINOGRAD=..] loss_trace.py [steps] [--no-graph]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from i2vsgg_amd import train
from i2vsgg_amd.model.utils import config as c

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
c.cfg_from_file(c.default_cfg_file("res101"))
c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30",
                 "TRAIN.BATCH_SIZE", "32", "TRAIN.RPN_POST_NMS_TOP_N_TARGET", "32"])
dev = torch.device("cuda:0")
net = train.build_sgg_net(101, device=dev)
step = train.SGGEmbStep(net, 2, seed=1, device=dev, use_graph="--no-graph" not in sys.argv)
step.capture(warmup=2)
buf = torch.zeros(steps, device=dev)
for i in range(steps):
    step()
    buf[i].copy_(step.loss.detach().reshape(()))
torch.cuda.synchronize()
w = net.vrd.fc7.fc.weight
print(" ".join("%.6f" % v for v in buf.tolist()))
print("fc7 |w| %.6f  fc6 |w| %.6f  graph_error %s" % (float(w.abs().sum()), float(net.vrd.fc6.fc.weight.abs().sum()), getattr(step, "graph_error", None)))
