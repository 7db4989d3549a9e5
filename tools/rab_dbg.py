import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from i2vsgg_amd import ops as O
from i2vsgg_amd._lib import lib, TUNE
import i2vsgg_amd.synthetic as syn
DEV = "cuda:0"
for (C, H, W, B, R, avg) in [(128, 9, 11, 2, 6, True), (128, 19, 32, 2, 9, True), (128, 19, 32, 2, 9, False), (1024, 38, 63, 4, 32, True)]:
    rois = np.concatenate([np.concatenate([np.full((R, 1), b, np.float32), syn.boxes(R * 7 + b, R, H * 16, W * 16, 8, min(H, W) * 12)], 1) for b in range(B)]).astype(np.float32)
    rng = np.random.default_rng(1)
    gout = rng.standard_normal((rois.shape[0], C, 7, 7), dtype=np.float32)
    rt, gt = torch.from_numpy(rois).to(DEV), torch.from_numpy(gout).to(DEV).contiguous(memory_format=torch.channels_last)
    res = {}
    for form in (0, 1):
        lib.i2v_set_tuning(TUNE["I2V_ROIALIGN_BWD"], form)
        feat = torch.zeros((B, C, H, W), device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
        O.roi_align(feat, rt, 7, 7, 1.0 / 16.0, avg=avg).backward(gt)
        res[form] = feat.grad.clone()
    d = (res[0] - res[1]).abs()
    print((C, H, W, B, R, avg), "max diff", float(d.max()), "scale", float(res[0].abs().max()), "n bad", int((d > 1e-5 * float(res[0].abs().max())).sum()), "of", d.numel())
    if float(d.max()) > 1e-5:
        idx = torch.nonzero(d > 1e-5 * float(res[0].abs().max()))
        print("  first bad (b,c,y,x):", idx[:8].tolist())
        print("  bad rows y:", sorted(set(idx[:, 2].tolist()))[:20], "cols x:", sorted(set(idx[:, 3].tolist()))[:30], "channels:", sorted(set((idx[:, 1] // 4).tolist()))[:40])
        b, c, y, x = idx[0].tolist()
        print("  values form0 %g form1 %g" % (float(res[0][b, c, y, x]), float(res[1][b, c, y, x])))
