"""Frames/s of the two per-frame EVALUATION loops (test_net_instance_styleD_bilinear.py:140-221, test_net_SGG_emb.py per frame)
at full size: eager launches, one frame at a time as the reference evaluates (batch_size 1), host copies of the results
included -- the loop body as a user of i2vsgg_amd.eval runs it.

    python tools/eval_probe.py [--frames 20] [--boxes 8]
"""
import argparse
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--boxes", type=int, default=8)
    ap.add_argument("--layers", type=int, default=101)
    a = ap.parse_args()
    from i2vsgg_amd import eval as ev, synthetic as syn, train
    from i2vsgg_amd.model.utils.config import cfg, cfg_from_file
    cfg_from_file(os.path.join(os.path.dirname(train.__file__), "cfgs", "res101.yml"))
    dev = "cuda:0"

    def timed(fn, n):
        fn(0); fn(1)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n):
            fn(2 + i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    frames = [torch.from_numpy(syn.frames(100 + i, 1)[0]).to(dev).contiguous(memory_format=torch.channels_last) for i in range(4)]
    info = torch.tensor([[600.0, 1000.0, 1.0]], device=dev)
    z, nb = torch.zeros(1, 1, 5, device=dev), torch.zeros(1, device=dev)

    det = train.build_instance_styled_net(a.layers, device=dev).eval()
    t_det = timed(lambda i: ev.detect_frame(det, frames[i % 4], info, z, nb, thresh=0.0, max_per_image=100), a.frames)
    print("detect_frame   (TEST %d -> %d proposals): %.2f ms/frame = %.1f frames/s" % (
        cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N, 1e3 * t_det, 1 / t_det))
    for nf in (1, 2, 3, 4):
        step = ev.DetectStep(det, frames=nf, device=dev)
        batch = [(torch.cat([frames[(i + j) % 4] for j in range(nf)]), info.expand(nf, 3).contiguous()) for i in range(4)]
        for b in batch[:2]:
            step(*b)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 0
        for _ in step.run(batch[i % 4] for i in range(a.frames)):
            n += nf
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        print("DetectStep frames=%d graph=%s: %.2f ms/frame = %.1f frames/s %s" % (nf, bool(step.shapes[step._staged].graph), 1e3 * dt,
                                                                               1 / dt, step.graph_error or ""))
        del step
    del det
    torch.cuda.empty_cache()

    sgg = train.build_sgg_net(a.layers, device=dev).eval()
    sgg.vrd.target_gt_rels = {"f%d" % i: syn.relation_annotation(31 + i, a.boxes, a.boxes, 62, 16) for i in range(4)}
    t_rel = timed(lambda i: ev.relation_frame(sgg, frames[i % 4], info, "f%d" % (i % 4)), a.frames)
    print("relation_frame (%d boxes, %d ordered pairs): %.2f ms/frame = %.1f frames/s" % (
        a.boxes, a.boxes * (a.boxes - 1), 1e3 * t_rel, 1 / t_rel))
    for nf in (1, 2, 3, 4):
        step = ev.RelationStep(sgg, frames=nf, device=dev, cap_boxes=a.boxes + 1)
        batch = [(torch.cat([frames[(i + j) % 4] for j in range(nf)]), info.expand(nf, 3).cpu().numpy(),
                  ["f%d" % ((i + j) % 4) for j in range(nf)]) for i in range(4)]
        for b in batch[:2]:
            step(*b)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 0
        for _ in step.run(batch[i % 4] for i in range(a.frames)):
            n += nf
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        print("RelationStep frames=%d graph=%s: %.2f ms/frame = %.1f frames/s %s" % (nf, bool(step.shapes[step._staged].graph), 1e3 * dt,
                                                                                 1 / dt, step.graph_error or ""))
        del step


if __name__ == "__main__":
    main()
