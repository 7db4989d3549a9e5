#!/usr/bin/env python3
"""Which aten ops of one eager SGG_emb step launch the small glue kernels (copies, fills, element-wise)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from i2vsgg_amd import train
from i2vsgg_amd.model.utils import config as c
c.cfg_from_file(c.default_cfg_file("res101"))
net = train.build_sgg_net(101, device="cuda:0")
step = train.SGGEmbStep(net, 2, seed=1, device="cuda:0", use_graph=False)
for _ in range(2):
    step._body()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step._head()
    step.opt.step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=40, max_shapes_column_width=60))
print("---- aten ops that launch device work (count, self device time) ----")
rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.key.startswith("aten::") and e.self_device_time_total > 0]
for k, c, t in sorted(rows, key=lambda r: -r[2]):
    print("%-44s %4d calls %9.1f us" % (k, c, t))
print("total aten device time %.1f us in %d launching calls" % (sum(r[2] for r in rows), sum(r[1] for r in rows)))
