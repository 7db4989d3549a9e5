import sys, torch
sys.path.insert(0, "/root/repo")
from i2vsgg_amd import ops, _lib
from i2vsgg_amd._lib import lib
def run(M, N, bias=True, relu=True):
    gy = torch.randn(M, N, device="cuda"); y = torch.randn(M, N, device="cuda"); g = torch.empty_like(gy)
    gb = torch.zeros(N, device="cuda") if bias else None
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: lib.i2v_epilogue_bwd(gy.data_ptr(), y.data_ptr(), None, g.data_ptr(), None, gb.data_ptr() if bias else None, M, N, int(relu), st)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print("M %6d N %6d bias %d: %.1f us" % (M, N, bias, e0.elapsed_time(e1) / 20 * 1e3))
for M, N in ((128, 4096), (64, 4096), (128, 256), (64, 300), (16384, 96), (4096, 128), (64, 64)):
    run(M, N, True); run(M, N, False)
