#!/usr/bin/env python3
"""Round-3 review item 2: WHICH change removed the host segfault in hipGraphLaunch that round 3 met with the copy-stream upload
(tests/test_gpu_configs.py followed by tests/test_gpu_data_layer.py in one process)?  Two things changed since the crashing
commit: (a) no graph is dropped while a replay of it may still be in flight (invalidate_graphs / the LRU eviction synchronise
first), (b) streams come from ops.role_stream (own HIP streams) instead of torch.cuda.Stream() (32 pooled handles, round robin).
This program runs the crashing order ONCE with (a) in place and (b) undone: ops.role_stream is replaced by the round-3 behaviour
(a fresh torch.cuda.Stream() per request) and ops.branch's alias refusal by a logger.  Every fork is checked for aliases -- with
the forking stream, with open siblings, with torch.cuda.graph's capture stream, with the copy stream -- and logged (flushed)
to gpurun_out/r04_alias_repro.txt before it happens, so that the record survives a crash.  A top-level program.

    I2V_UPLOAD_STREAM=1 python tools/alias_repro.py"""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ["I2V_UPLOAD_STREAM"] = "1"
os.environ["I2V_ALIAS_REPRO"] = "1"          # the test skips its role-table bookkeeping (there is no table here) and goes on to its loss / weight comparison
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytest  # noqa: E402
import torch  # noqa: E402

from i2vsgg_amd import ops  # noqa: E402

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "r05_alias_repro.txt"), "w", buffering=1)
DRAWS = []


def say(msg):
    LOG.write(msg + "\n")
    LOG.flush()
    os.fsync(LOG.fileno())


def pooled(device, role, priority=0):
    st = torch.cuda.Stream(torch.device(device), priority=priority)
    DRAWS.append((role, st.cuda_stream))
    same = [r for r, h in DRAWS[:-1] if h == st.cuda_stream]
    cap = getattr(torch.cuda.graph, "default_capture_stream", None)
    say("draw %d: role %s -> %#x%s%s" % (len(DRAWS), role, st.cuda_stream,
                                          "  == earlier draws %s" % same if same else "",
                                          "  == torch.cuda.graph capture stream" if cap is not None and cap.cuda_stream == st.cuda_stream else ""))
    return st


class LoggingBranch(ops.branch):
    def __enter__(self):
        h, ho = self.stream.cuda_stream, self.origin.cuda_stream
        cap = torch.cuda.is_current_stream_capturing()
        if h == ho:
            say("FORK ONTO THE FORKING STREAM ITSELF: %#x (capturing: %s)" % (h, cap))
        if h in ops._FORKED:
            say("FORK ONTO AN OPEN SIBLING: %#x (capturing: %s)" % (h, cap))
        copy = [hh for r, hh in DRAWS if r == "copy"]
        if h in copy or ho in copy:
            say("THE COPY STREAM %#x IS %s (capturing: %s)" % (copy[-1], "a branch" if h in copy else "the forking stream", cap))
        ops._FORKED.pop(h, None)
        self.stream.wait_stream(self.origin)
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        ops._FORKED[h] = ho
        ops._BRANCH_DEPTH += 1
        return self


ops.role_stream = pooled
ops.branch = LoggingBranch
say("crashing order, copy stream ON, pooled streams as in round 3, graphs dropped only after a synchronise (HEAD)")
rc = pytest.main(["-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_gpu_configs.py"),
                  os.path.join(ROOT, "tests", "test_gpu_data_layer.py"), "-k", "not alias"])
say("pytest exit code %d after %d pooled draws" % (rc, len(DRAWS)))
sys.exit(int(rc))
