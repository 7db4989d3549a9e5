import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
x = torch.randn(128, 50176, device="cuda")
out = torch.empty(128, 50176, device="cuda")
g2 = torch.randn(64, 64, device="cuda")
# warm up outside capture
dist.all_gather_into_tensor(out, x); dist.all_reduce(g2)
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        y = x * 2
        dist.all_gather_into_tensor(out, y)
        z = out.sum()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        y = x * 2
        dist.all_gather_into_tensor(out, y)
        z = out.sum()
    x.fill_(1.0)
    g.replay(); torch.cuda.synchronize()
    print("captured collective ok: z =", z.item(), "expected", 2.0 * 128 * 50176)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("replay %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
except Exception as e:
    print("capture FAILED:", repr(e)[:300])
dist.destroy_process_group()
