#!/usr/bin/env python3
"""Ordering probe for the two-stream SGG_emb step (round-1 failure: wrong / zero-logit / NaN losses when the step was
replayed back to back from HIP's legacy default stream).

Drives the overlapped replay sequence by hand on a chosen stream, WITHOUT the detour / synchronize of
SGGEmbStep.__call__, and records:
  * the per-step loss trajectory (compared with a reference run that synchronises after every step);
  * an in-graph canary: the head graph increments `start` when it begins and `end` after its SGD update, and logs
    start - end at its beginning.  Anything but 1 means two replays of the head graph overlapped on the device.

usage: race_probe.py MODE[,MODE...] [reps]
modes: own0 own-1 def0 def-1 def_nobb def_nodrop def_gap own_nobb
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

if os.environ.get("RACE_LATE_ENV"):      # does the runtime still honour the flag when it is set after `import torch`?
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"

from i2vsgg_amd import train  # noqa: E402
from i2vsgg_amd.model.utils import config as c  # noqa: E402

DEV = torch.device("cuda:0")
N = 23


def run(mode, sync=False):
    own = mode.startswith("own")
    prio = -1 if "-1" in mode else 0
    nobb = "nobb" in mode
    nodrop = "nodrop" in mode
    gap = "gap" in mode
    noev = "noev" in mode           # the backbone graph replays on its stream with NO event tying it to the head's stream
    os.environ["I2V_BB_PRIORITY"] = str(prio)
    net = train.build_sgg_net(101, device=DEV)
    if nodrop:
        net.vrd.dropout = False
    step = train.SGGEmbStep(net, 2, seed=1, device=DEV)
    if os.environ.get("RACE_WS_UNIQUE"):
        # every workspace request of the given tag gets a buffer of its own, kept alive for the life of the process:
        # no two launches share scratch (tests whether the shared grow-only scratch is the hazard)
        from i2vsgg_amd import ops
        keep = globals().setdefault("_KEEP", [])
        tags = os.environ["RACE_WS_UNIQUE"].split(",")
        ws0 = globals().setdefault("_WS0", ops.workspace)

        def ws_unique(nbytes, device, tag="default"):
            if tag in tags:
                b = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
                keep.append(b)
                return b
            return ws0(nbytes, device, tag)
        ops.workspace = ws_unique
    dot = os.environ.get("RACE_DOT")
    if dot and not globals().get("_DOT_DONE"):
        G0 = torch.cuda.CUDAGraph
        made = []

        class G(G0):
            def __init__(self, *a, **k):
                super().__init__(*a, **k)
                self.enable_debug_mode()
                made.append(self)
        torch.cuda.CUDAGraph = G
    start = torch.zeros(1, dtype=torch.long, device=DEV)
    end = torch.zeros(1, dtype=torch.long, device=DEV)
    log = torch.full((64,), -7.0, device=DEV)
    head0, opt0 = step._head, step.opt.step

    def head():
        start.add_(1)
        log.index_copy_(0, start, (start - end).float())
        head0()

    def optstep():
        opt0()
        end.add_(1)

    step._head, step.opt.step = head, optstep
    prev = torch.cuda.current_stream()
    if own:
        s = torch.cuda.Stream()
        s.wait_stream(prev)
        torch.cuda.set_stream(s)
    try:
        assert step.capture(warmup=2) and step.overlap, getattr(step, "graph_error", None)
        gbb, gh = step.graph[0], step.graph[1]
        if dot and not globals().get("_DOT_DONE"):
            globals()["_DOT_DONE"] = True
            torch.cuda.CUDAGraph = G0
            for i, g in enumerate(made[:2]):
                try:
                    g.debug_dump("%s.%d.dot" % (dot, i))
                    txt = open("%s.%d.dot" % (dot, i)).read()
                    import re
                    kinds = {k: len(re.findall(k, txt)) for k in ("MEMSET", "MEMCPY", "KERNEL", "->")}
                    print("DOT graph %d: %s, %d bytes" % (i, kinds, len(txt)), flush=True)
                except Exception as e:
                    print("DOT dump failed:", repr(e), flush=True)
        cur = torch.cuda.current_stream(DEV)
        torch.cuda.synchronize()
        if os.environ.get("RACE_WARM_NULL") and not own:
            # one EAGER head pass on the stream the replays will use (the legacy default stream): every kernel of the head
            # has then been dispatched once through the runtime's ordinary path on this stream's hardware queue
            head0()
            torch.cuda.synchronize()
        start.zero_(); end.zero_(); log.fill_(-7.0)
        trace = torch.zeros(N, device=DEV)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            if not noev:
                cur.wait_event(step.ev_bb)
            step.fmap_head.copy_(step.fmap)
            if not noev:
                step.ev_copy.record(cur)
            if not nobb:
                with torch.cuda.stream(step.s_bb):
                    if not noev:
                        step.s_bb.wait_event(step.ev_copy)
                    gbb.replay()
                    if not noev:
                        step.ev_bb.record(step.s_bb)
            gh.replay()
            trace[i].copy_(step.loss.detach().reshape(()))
            if sync:
                torch.cuda.synchronize()
            if gap:
                time.sleep(0.008)
            if i == 2:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N * 1e3
        lg = log.tolist()
        bad = [(k, v) for k, v in enumerate(lg[1:N + 1], 1) if v != 1.0]
        w = net.vrd.fc7.fc.weight.detach().double().abs().sum().item()
        return trace.tolist(), bad, w, dt
    finally:
        torch.cuda.set_stream(prev)
        step.opt.unfuse()


def main():
    modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["own0", "def0"]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    c.cfg_from_file(c.default_cfg_file("res101"))
    c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30",
                     "TRAIN.BATCH_SIZE", "32", "TRAIN.RPN_POST_NMS_TOP_N_TARGET", "32"])
    refs = {}
    for mode in modes:
        key = "nodrop" if "nodrop" in mode else "drop"
        if key not in refs:
            refs[key] = run("own0_nodrop" if key == "nodrop" else "own0", sync=True)
            print("REF %-6s final %.7f fc7 %.6f canary_bad %s" % (key, refs[key][0][-1], refs[key][2], refs[key][1]), flush=True)
        ref = refs[key][0]
        for r in range(reps):
            tr, bad, w, dt = run(mode)
            dev = [abs(a - b) if a == a else float("inf") for a, b in zip(tr, ref)]
            first = next((i for i, d in enumerate(dev) if d > 2e-6), None)
            print("RUN %-11s rep %d  %.3f ms/step  final %.7f  fc7 %.6f  first_dev_step %s  max_dev %.3g  canary_bad %s" % (
                mode, r, dt, tr[-1], w, first, max(dev), bad[:6]), flush=True)
            if first is not None:
                print("    trace " + " ".join("%.6f" % v for v in tr), flush=True)


if __name__ == "__main__":
    main()
