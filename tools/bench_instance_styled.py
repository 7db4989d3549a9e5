#!/usr/bin/env python3
"""Secondary measurement: BASELINE.json configs[2] -- cfgs/res101.yml, instance_styleD D+G adversarial step,
batch = 4 source + 4 target frames of 600x1000, 32 ROI/frame, 1 MI355X (eager; see train.InstanceStyleDStep)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from i2vsgg_amd import train
from i2vsgg_amd.model.utils import config as c

B = int(os.environ.get("B", "4"))
c.cfg_from_file(c.default_cfg_file("res101"))
c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30",
                 "TRAIN.BATCH_SIZE", "32", "TRAIN.RPN_POST_NMS_TOP_N_TARGET", "32"])
np.random.seed(c.cfg.RNG_SEED)
net = train.build_instance_styled_net(101)
step = train.InstanceStyleDStep(net, B)
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(json.dumps({"workload": "configs[2] instance_styleD D+G step, %d source + %d target frames 600x1000, 32 ROI/frame" % (B, B),
                  "ms_per_step": dt * 1e3, "frames_per_s": 2 * B / dt,
                  "losses": {k: float(v) for k, v in step.losses.items()},
                  "max_mem_GB": torch.cuda.max_memory_allocated() / 2**30}))
