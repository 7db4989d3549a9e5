#!/usr/bin/env python3
"""Host time per step of the loader-fed relation step (bench.py's sgg_loader leg): stage_batch() and the graph launch, timed on
the host with the GPU free-running, against the device time per step.   python tools/loader_host_probe.py [u8]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


class A:
    layers = 101
    no_graph = False
    steps = 40
    warmup = 8


u8 = len(sys.argv) > 1 and sys.argv[1] == "u8"
dev = torch.device("cuda:0")
orig = bench.timed_steps
rec = {}


def spy(fn, warmup, steps, dev):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t_host = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        a = time.perf_counter()
        fn()
        t_host += time.perf_counter() - a
    torch.cuda.synchronize()
    rec["host_ms"] = 1e3 * t_host / steps
    rec["wall_ms"] = 1e3 * (time.perf_counter() - t0) / steps
    return time.perf_counter() - t0


bench.timed_steps = spy
_orig_dl = torch.utils.data.DataLoader


class _DL(_orig_dl):                      # keep the collated minibatches for the profile below
    def __iter__(self):
        for d in super().__iter__():
            rec.setdefault("batches", []).append(d)
            yield d


torch.utils.data.DataLoader = _DL
line, step, net = bench.run_sgg_loader(A, 0, 1, dev, u8=u8)
print("host %.2f ms per step (stage + launch), wall %.2f ms per step; sizes %s" % (rec["host_ms"], rec["wall_ms"], line["config"]["frame_sizes"]))
# the same step objects with the batches already staged: device time alone
if os.environ.get("PROFILE"):
    import cProfile
    import pstats
    batches = rec["batches"]
    stage = step.stage_batch_u8 if u8 else step.stage_batch
    pos = [0]

    def fn():
        pos[0] += 1
        stage(batches[pos[0] % len(batches)])
        step()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(40):
        fn()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
