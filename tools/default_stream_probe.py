#!/usr/bin/env python3
"""A process whose HIP runtime is up BEFORE i2vsgg_amd is imported (so the package cannot switch the runtime's graph packet path
off) and that calls the captured relation step from the legacy default stream: 20 steps back to back, no host
synchronisation, against the same steps synchronised one by one on a created stream.  train.replay_graph moves the replays onto
a private stream between event edges; the trajectories must agree per step and per traced tensor.  Prints OK / exits 1.

    env -u DEBUG_CLR_GRAPH_PACKET_CAPTURE python tools/default_stream_probe.py [zeros|is_available]

``zeros`` (default): a tensor is made on the device before the import (torch.cuda.is_initialized() is True, the package leaves the
environment alone).  ``is_available``: only torch.cuda.is_available() runs before the import -- hipGetDeviceCount brings the
runtime up with its defaults, torch does not count that as initialised, so the package DOES write the variable, too late to
matter (round-3 advice).  In both orders the step must not trust the variable.  A top-level program: never start it from a
process that has touched the GPU."""
import os
import sys

os.environ.pop("DEBUG_CLR_GRAPH_PACKET_CAPTURE", None)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

MODE = sys.argv[1] if len(sys.argv) > 1 else "zeros"
if MODE == "is_available":
    assert torch.cuda.is_available()                 # hipGetDeviceCount: the runtime is up, torch.cuda.is_initialized() is not
    assert not torch.cuda.is_initialized()
else:
    torch.zeros(1, device="cuda:0")                  # the HIP runtime initialises here, with its defaults
import i2vsgg_amd  # noqa: E402,F401
from i2vsgg_amd import train  # noqa: E402

if MODE == "is_available":
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"      # written -- after the runtime had read its defaults
else:
    assert "DEBUG_CLR_GRAPH_PACKET_CAPTURE" not in os.environ, "the package must leave a running runtime's setting alone"
DEV, N = torch.device("cuda:0"), 20
cols = train.SGGEmbStep.TRACE_COLS
tol = dict(loss=1e-5, features=1e-9, scores=1e-5, embedding=1e-5, rng_canary=0.0, fc7_weight=1e-7, fc6_weight_head=1e-7,
           boxes=0.0, labels=0.0)


def run(own_stream, sync):
    net = train.build_sgg_net(101, device=DEV)
    step = train.SGGEmbStep(net, 2, seed=1, device=DEV, trace_rows=N + 8)
    prev = torch.cuda.current_stream()
    if own_stream:
        s = torch.cuda.Stream()
        s.wait_stream(prev)
        torch.cuda.set_stream(s)
    try:
        assert step.capture(warmup=2) and step.overlap, step.graph_error
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        for _ in range(N):
            step()
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return step.trace[:N + 5].cpu().numpy()
    finally:
        torch.cuda.set_stream(prev)
        step.opt.unfuse()


want = run(True, True)
bad = []
for rep in range(2):
    got = run(False, False)                          # the default stream, back to back
    for r in range(want.shape[0]):
        for c, name in enumerate(cols):
            if abs(want[r, c] - got[r, c]) > tol[name] * abs(want[r, c]):
                bad.append("run %d step %d %s: %.10g vs %.10g" % (rep, r, name, got[r, c], want[r, c]))
                break
        if bad:
            break
print("redirected replays: %d private stream(s)" % len(train._REPLAY_STREAMS))
if bad or not train._REPLAY_STREAMS:
    print("MISMATCH", bad)
    sys.exit(1)
print("OK (%s before the import): %d back-to-back steps from the default stream follow the synchronised trajectory (loss %.6f -> %.6f)" % (
    MODE, N, want[2, 0], want[-1, 0]))
