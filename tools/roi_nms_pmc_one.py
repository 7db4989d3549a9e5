#!/usr/bin/env python3
"""ONE case of bench.py's ROI_NMS_CASES (the launches behind ``roofline_hbm`` / ``also.roi_nms``) in ONE cache state, for the
rocprofv3 passes of tools/roi_nms_pmc.sh (kernel trace, --pmc FETCH_SIZE, --pmc WRITE_SIZE: three separate runs per case
and state).  N + 5 ops are launched; the summary drops each kernel's first 5 ops (cold start, first-touch of the buffers).

usage: roi_nms_pmc_one.py CASE warm|cold      (no argument: list the case names)
Prints one JSON line: the case, its ops, the kernels with their launches per op and the algorithmic bytes per op."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
import bench

N, DROP = 100, 5
if len(sys.argv) < 3:
    print(" ".join(bench.ROI_NMS_CASES))
    sys.exit(0)
name, state = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
case = bench.roi_nms_case(dev, name, cold=state == "cold")
sets = case["sets"]
n = -(-N // sets) * sets                      # whole rotations
ring = [None] * sets
torch.cuda.synchronize()
for i in range(DROP + n):
    ring[i % sets] = case["fn"](i)
torch.cuda.synchronize()
print(json.dumps({"case": name, "state": state, "ops": n, "dropped": DROP, "sets": sets, "algorithmic_bytes": case["nbytes"],
                  "kernels": case["kernels"]}))
