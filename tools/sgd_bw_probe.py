import os, sys
sys.path.insert(0, "/root/repo")
import i2vsgg_amd, torch
from i2vsgg_amd import ops
n = 4096 * 50176
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda"); m = torch.zeros(n, device="cuda")
for _ in range(3): ops.sgd_momentum_(p, g, m, 1e-4, 0.9, 5e-4)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.sgd_momentum_(p, g, m, 1e-4, 0.9, 5e-4)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
print("sgd_momentum on 822 MB: %.1f us, %.2f TB/s (5 streams), %.2f TB/s counting 4 streams" % (t * 1e6, 5 * n * 4 / t / 1e12, 4 * n * 4 / t / 1e12))
q = torch.empty_like(p)
for _ in range(3): q.copy_(p)
e0.record()
for _ in range(10): q.copy_(p)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
print("copy 822 MB: %.1f us, %.2f TB/s" % (t * 1e6, 2 * n * 4 / t / 1e12))
