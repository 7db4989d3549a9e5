#!/usr/bin/env python3
"""ONE case of bench.py's ``roi_nms_cases`` (the launches behind ``also.roi_nms``), N launches, for the rocprofv3 passes of
tools/roi_nms_pmc.sh (kernel trace, --pmc FETCH_SIZE, --pmc WRITE_SIZE: three separate runs per case).

usage: roi_nms_pmc_one.py CASE      (no argument: list the case names)
Prints one JSON line: the case, its launches and its algorithmic bytes per launch."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
import bench

N = 10
dev = torch.device("cuda:0")
cases = bench.roi_nms_cases(dev)
if len(sys.argv) < 2:
    print(" ".join(c[0] for c in cases))
    sys.exit(0)
name, fn, nbytes, kernels = next(c for c in cases if c[0] == sys.argv[1])
torch.cuda.synchronize()
for _ in range(N):
    fn()
torch.cuda.synchronize()
print(json.dumps({"case": name, "launches": N, "algorithmic_bytes": nbytes, "kernels": kernels}))
