import os, time, torch
print({k:v for k,v in os.environ.items() if "SDMA" in k or "HSA" in k or "HIP" in k or "ROC" in k})
dev=torch.device("cuda:0")
src=torch.randn(2,3,600,1000).pin_memory()
dst=torch.empty_like(src,device=dev)
cs=torch.cuda.Stream()
a=torch.randn(8192,8192,device=dev)
def copy_time(busy):
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    if busy:
        for _ in range(6): torch.mm(a,a)
    with torch.cuda.stream(cs):
        e0.record()
        dst.copy_(src,non_blocking=True)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for _ in range(3): copy_time(False)
print("H2D 14.4 MB alone: %.3f ms"%min(copy_time(False) for _ in range(5)))
print("H2D 14.4 MB beside 6 big GEMMs: %.3f ms"%min(copy_time(True) for _ in range(5)))
u8=torch.randint(0,255,(2,600,1000,3),dtype=torch.uint8).pin_memory(); d8=torch.empty_like(u8,device=dev)
def c8(busy):
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    if busy:
        for _ in range(6): torch.mm(a,a)
    with torch.cuda.stream(cs):
        e0.record(); d8.copy_(u8,non_blocking=True); e1.record()
    torch.cuda.synchronize(); return e0.elapsed_time(e1)
print("H2D 3.6 MB alone %.3f ms, busy %.3f ms"%(min(c8(False) for _ in range(5)), min(c8(True) for _ in range(5))))
