#!/usr/bin/env python3
"""Launches of ONE step from a `rocprofv3 --kernel-trace` csv: the window between two consecutive launches of an anchor kernel
that occurs once per step, split into the library's own kernels and everything else (aten / rocclr).  The `--stats` summary
divided by the number of steps also counts the one-time set-up of the process (parameter uploads through the blit kernel,
folded-BN arithmetic, filter transforms: ~1400 launches in front of the first step of bench.py).

  step_launches.py <kernel_trace.csv> [anchor substring = roi_pool_fwd | maxpool3x3s2] [occurrences of the anchor per step = 1]"""
import collections, csv, sys

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "roi_pool_fwd"
per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
OURS = ("conv_", "wino", "roi_", "sgd_", "bce_", "l2norm", "epilogue", "maxpool", "nms", "sort_", "dstyle", "dpixel", "rpn_", "bbox_",
        "weight_dgrad", "fc_fold", "gather_dets", "write_rois", "image_prep", "det_", "pair_gather", "dp_transpose", "half_mse", "smooth_l1", "signed_sqrt")
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if anchor in n]
if len(idx) < 2 * per + 1:
    sys.exit("anchor %r occurs %d times" % (anchor, len(idx)))
a, b = idx[-1 - 2 * per], idx[-1 - per]            # the last complete step but one
ours, glue = collections.Counter(), collections.Counter()
t_ours = t_glue = 0.0
for r in rows[a:b]:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "Cijk_" in n:                                  # bench.py's blocker GEMM in front of a profiled eager step
        continue
    if any(t in n for t in OURS):
        ours[n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]] += 1
        t_ours += d
    else:
        glue[n.replace("void ", "")[:100]] += 1
        t_glue += d
print("one step (trace rows %d..%d of %d; %d steps in the trace)" % (a, b, len(rows), len(idx) // per))
print("library kernels: %d launches, %.1f us" % (sum(ours.values()), t_ours))
print("aten / rocclr:   %d launches, %.1f us" % (sum(glue.values()), t_glue))
for k, v in glue.most_common():
    print("  %3d  %s" % (v, k))
