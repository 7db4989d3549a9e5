"""Which launches of a step are not ours: one eager SGG_emb step (and optionally one instance_styleD step) under
torch.profiler, aten / runtime kernels grouped by the operator that launched them, with the python frame that called it.
Usage: python tools/glue_trace.py [sgg|isd]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
from torch.profiler import profile, ProfilerActivity
from i2vsgg_amd import train
from i2vsgg_amd.model.utils import config as c

which = sys.argv[1] if len(sys.argv) > 1 else "sgg"
c.cfg_from_file(c.default_cfg_file("res101"))
c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30", "TRAIN.BATCH_SIZE", "32",
                 "TRAIN.RPN_POST_NMS_TOP_N_TARGET", "32"])
dev = "cuda:0"
if which == "sgg":
    net = train.build_sgg_net(101, device=dev)
    step = train.SGGEmbStep(net, 2, device=dev)
    step.capture(warmup=2)
    step._pipelined = False
    body = step._body
else:
    net = train.build_instance_styled_net(101, device=dev)
    step = train.InstanceStyleDStep(net, 4, device=dev)
    step.capture(warmup=2)
    body = step._body_branches
body()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    body()
    torch.cuda.synchronize()
rows = []
for a in prof.key_averages(group_by_stack_n=8):
    if a.device_time_total <= 0 or not a.key.startswith("aten::"):
        continue
    frame = ""
    for fr in a.stack:
        if "i2vsgg_amd" in fr:
            frame = fr.split("repo/")[-1][:100]
            if "/ops.py" not in frame:
                break
    rows.append((a.count, a.device_time_total, a.key, frame))
print("aten ops with device time in one eager %s step: %d calls" % (which, sum(r[0] for r in rows)))
for n, t, k, f in sorted(rows, key=lambda r: -r[0]):
    print("%4d %8.1f us  %-34s %s" % (n, t, k, f))

print("\nlargest by device time, with input shapes:")
big = []
for a in prof.key_averages(group_by_input_shape=True):
    if a.device_time_total > 0 and a.key.startswith("aten::"):
        big.append((a.device_time_total, a.count, a.key, str(a.input_shapes)[:110]))
for t, n, k, sh in sorted(big, reverse=True)[:45]:
    print("%8.1f us %4d  %-26s %s" % (t, n, k, sh))
