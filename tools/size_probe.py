#!/usr/bin/env python3
"""Why is the relation step SLOWER on a 600x801 minibatch than on 600x1000 (bench.py also.sgg_loader.resident_ms_by_size)?
The captured step resident at each size, and the launches of one eager backbone+head step with their HIP-event times.
  tools/size_probe.py 600x1000 600x801 [--dump DIR]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

import bench  # noqa: E402
from i2vsgg_amd import train  # noqa: E402

DEV = torch.device("cuda:0")
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if "x" in a]
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
for h, w in sizes:
    net = train.build_sgg_net(101, device=DEV)
    step = train.SGGEmbStep(net, 2, seed=1, device=DEV, h=h, w=w)
    assert step.capture(warmup=2), step.graph_error
    el = bench.timed_steps(step, 5, 30, DEV)
    pp, step._pipelined = step._pipelined, False
    rec = bench.profile_eager(step._body, 2, DEV)
    step._pipelined = pp
    per = len(rec) // 2
    last = rec[-per:]
    kinds = {}
    for r in last:
        k = r["tag"] + (" [gemm]" if "[gemm]" in r["desc"] else " winograd" if "winograd" in r["desc"] else "")
        d = kinds.setdefault(k, [0, 0.0, 0.0])
        d[0] += 1; d[1] += r["t"] * 1e6; d[2] += r["flops"]
    print("== %dx%d: captured step %.3f ms; eager launches %d, event time %.0f us" % (h, w, 1e3 * el / 30, per, sum(r["t"] for r in last) * 1e6))
    for k, (n, us, fl) in sorted(kinds.items(), key=lambda kv: -kv[1][1]):
        print("   %-16s %4d launches %9.1f us %7.1f TF" % (k, n, us, fl / us / 1e6 if us else 0))
    if dump:
        os.makedirs(dump, exist_ok=True)
        with open(os.path.join(dump, "launches_%dx%d.txt" % (h, w)), "w") as f:
            for r in last:
                f.write("%-6s %-60s %8.1f us %7.2f GF %6.1f TF\n" % (r["tag"], r["desc"], r["t"] * 1e6, r["flops"] / 1e9, r["flops"] / r["t"] / 1e12))
    step.opt.unfuse()
    del step, net
    torch.cuda.empty_cache()
