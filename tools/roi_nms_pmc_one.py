#!/usr/bin/env python3
"""One configuration of every HBM-side kernel of the path, launched N times each, for the rocprofv3 passes of
tools/roi_nms_pmc.sh (kernel trace, --pmc FETCH_SIZE, --pmc WRITE_SIZE: three separate runs of this script).

usage: roi_nms_pmc_one.py CASE      CASE = b1 (one 600x1000 frame) | b4 (the multi-frame shapes of configs[1] / configs[2])
Prints one JSON line: the algorithmic bytes per launch of every op, keyed by the kernel that dominates it."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from i2vsgg_amd import ops, synthetic as syn
from i2vsgg_amd.model.rpn.generate_anchors import generate_anchors

case = sys.argv[1] if len(sys.argv) > 1 else "b1"
dev, N = "cuda:0", 10
B = 1 if case == "b1" else 4
R = 32
feat = torch.randn(B, 1024, 38, 63, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
rois = np.zeros((B * R, 5), np.float32)
for b in range(B):
    rois[b * R:(b + 1) * R, 0] = b
    rois[b * R:(b + 1) * R, 1:] = syn.boxes(b, R)
rt = torch.from_numpy(rois).to(dev)
fbytes, obytes = B * 1024 * 38 * 63 * 4, B * R * 1024 * 49 * 4
alg = {}
for _ in range(N):
    ops.roi_align(feat.detach(), rt, 7, 7, 1 / 16.0, avg=True)
alg["roi_align_fwd_nhwc_cols"] = fbytes + obytes
out = ops.roi_align(feat, rt, 7, 7, 1 / 16.0, avg=True)
g = torch.randn_like(out)
for _ in range(N):
    feat.grad = None
    out.backward(g, retain_graph=True)
alg["roi_align_bwd_kernel"] = fbytes + obytes
# ROIPool: the SGG_emb step pools boxes AND union boxes: 64 rois per frame (2 frames per GPU in configs[1])
Bp = 1 if case == "b1" else 2
featp = torch.randn(Bp, 1024, 38, 63, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
rp = np.zeros((Bp * 64, 5), np.float32)
for b in range(Bp):
    rp[b * 64:(b + 1) * 64, 0] = b
    rp[b * 64:(b + 1) * 64, 1:] = syn.boxes(10 + b, 64)
rpt = torch.from_numpy(rp).to(dev)
for _ in range(N):
    ops.roi_pool(featp.detach(), rpt, 7, 7, 1 / 16.0, out_nchw=True)
alg["roi_pool_fwd_c128_kernel"] = Bp * 1024 * 38 * 63 * 4 + 2 * Bp * 64 * 1024 * 49 * 4          # map read + values and argmax written
outp = ops.roi_pool(featp, rpt, 7, 7, 1 / 16.0, out_nchw=True)
gp = torch.randn_like(outp)
for _ in range(N):
    featp.grad = None
    outp.backward(gp, retain_graph=True)
alg["roi_pool_bwd_kernel"] = Bp * 1024 * 38 * 63 * 4 + 2 * Bp * 64 * 1024 * 49 * 4
n = 12000 if case == "b1" else 6000
dets = torch.from_numpy(syn.tie_free_dets(n, n, clustered=True)).to(dev)
for _ in range(N):
    ops.nms_sorted(dets, 0.7, 2000 if n == 12000 else 300)
ref_bytes = 20 * n + 2 * 8 * n * ((n + 63) // 64)
alg["nms_mask_kernel"] = ref_bytes
alg["nms_scan_pipelined_kernel"] = ref_bytes
base = torch.from_numpy(generate_anchors(scales=np.array([8, 16, 32]), ratios=np.array([0.5, 1, 2]))).float().to(dev)
cls = torch.randn(B, 18, 38, 63, device=dev).contiguous(memory_format=torch.channels_last)
box = (torch.randn(B, 36, 38, 63, device=dev) * 0.2).contiguous(memory_format=torch.channels_last)
info = torch.tensor([[600, 1000, 1.0]] * B, device=dev)
for _ in range(N):
    ops.rpn_proposal(cls, box, info, base, 16, 12000, 2000 if case == "b1" else 32, 0.7)
alg["rpn_decode_kernel"] = B * 21546 * (4 + 4 + 1 + 4) * 4
torch.cuda.synchronize()
print(json.dumps({"case": case, "launches_per_op": N, "algorithmic_bytes": alg}))
