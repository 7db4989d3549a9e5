import sys, torch
sys.path.insert(0, "/root/repo")
from i2vsgg_amd import ops
for M in (64, 128):
    x = torch.randn(M, 50176, device="cuda")
    w = torch.randn(4096, 50176, device="cuda") * 0.01
    b = torch.zeros(4096, device="cuda")
    for _ in range(3): y = ops.linear(x, w, b, relu=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y = ops.linear(x, w, b, relu=True)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    ref = torch.relu(x.double() @ w.double().t())
    print("M", M, "fc6 fwd %.1f us  %.1f TF  rel err %.2e" % (t * 1e6, 2.0 * M * 50176 * 4096 / t / 1e12, ((y.double() - ref).abs().max() / ref.abs().max()).item()))
