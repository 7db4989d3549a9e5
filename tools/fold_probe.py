"""Time i2v_fc_fold_fwd alone on the fc6 shape, with the diagnostic ablations of I2V_TUNE_FC_FOLD (where the
fused forward + pending-update kernel spends its time).  Usage: python tools/fold_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from i2vsgg_amd import ops
from i2vsgg_amd._lib import lib, ptr, stream

dev = "cuda:0"
M, N, K = 128, 4096, 50176
x = torch.randn(M, K, device=dev); xp = torch.randn(M, K, device=dev); gp = torch.randn(M, N, device=dev) * 1e-3
w = torch.randn(N, K, device=dev) * 0.01; m = torch.zeros(N, K, device=dev); b = torch.zeros(N, device=dev)
y = torch.zeros(M, N, device=dev)
valid = torch.ones(1, dtype=torch.int32, device=dev)


def run(n=10):
    for _ in range(2):
        lib.i2v_fc_fold_fwd(ptr(x), ptr(xp), ptr(gp), ptr(valid), ptr(w), ptr(m), ptr(b), ptr(y), M, M, N, K, 1e-6, 0.9, 5e-4, stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        lib.i2v_fc_fold_fwd(ptr(x), ptr(xp), ptr(gp), ptr(valid), ptr(w), ptr(m), ptr(b), ptr(y), M, M, N, K, 1e-6, 0.9, 5e-4, stream())
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, bits in (("as shipped", 0), ("no gradient MFMAs", 1), ("no forward MFMAs", 2), ("no MFMAs", 3), ("no x loads", 4),
                   ("no w/m stores", 8), ("no xp staging", 16), ("no xp staging, no gradient MFMAs", 17),
                   ("memory only (no MFMAs, no staging)", 19), ("MFMAs only (no x loads, no stores, no staging)", 28)):
    lib.i2v_set_tuning(15, bits)
    print("%-52s %8.1f us" % (name, run()))
lib.i2v_set_tuning(15, 0)
valid.zero_()
print("%-52s %8.1f us" % ("no pending update (plain forward)", run()))
