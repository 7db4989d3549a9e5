#!/usr/bin/env python3
"""Relation test loop (test_net_SGG_emb.py, per-frame part) on the HIP path: ``combined_roidb(imdbval_name, False) ->
roibatchLoader(training=False, normalize=False) -> DataLoader(batch_size=1)``; per frame the backbone, the eval branch of
``forward_relation`` on the frame's annotated boxes (all ordered pairs) and ``detection_output``'s top-100 triplets.  The
results go to ``<output_dir>/relations.pkl`` as {image path: (rlp_labels, tuple_confs, sub_bboxes, obj_bboxes, rel_idex)} --
the inputs of the reference's ``association`` / ``evaluate`` over videos, whose modules the reference does not ship
(SURVEY.md A3).  ``--frames`` frames at a time as one replayed HIP graph (``eval.RelationStep``; 1: ``eval.relation_frame``)."""
import argparse
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")     # before HIP initialises: i2vsgg_amd/__init__.py

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Evaluate the SGG_emb relation head on MI355X")
    p.add_argument("--dataset", default="synthetic")
    p.add_argument("--imdbval_name", default="synthetic_16_v")
    p.add_argument("--net", default="res101", choices=["res101", "res50"])
    p.add_argument("--load_name", default="", help="checkpoint in the reference's layout; empty: random-init weights")
    p.add_argument("--target_gt_rels_path", default="", help="pickle of {frame: {boxes, box_classes, rels}} (the imdb's otherwise)")
    p.add_argument("--num_relations", type=int, default=62)
    p.add_argument("--num_classes", type=int, default=16)
    p.add_argument("--nw", dest="num_workers", type=int, default=0)
    p.add_argument("--frames", type=int, default=4)
    p.add_argument("--scale", type=int, default=0)
    p.add_argument("--output_dir", default="output")
    p.add_argument("--device_prep", action="store_true", help="the loader hands over uint8 frames as decoded; BGR swap, mean "
                   "subtraction and resize run on the GPU (roibatchLoader(device_prep=True) + stage_u8); needs --frames >= 2")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--set", dest="set_cfgs", nargs=argparse.REMAINDER, default=None)
    return p.parse_args(argv)


def main(argv=None):
    a = parse_args(argv)
    from i2vsgg_amd import eval as ev, train
    from i2vsgg_amd.model.faster_rcnn.layers import load_reference_state
    from i2vsgg_amd.model.utils import config as c
    from i2vsgg_amd.roi_data_layer.roibatchLoader import roibatchLoader
    from i2vsgg_amd.roi_data_layer.roidb import combined_roidb
    dev = torch.device("cuda:0")
    c.cfg_from_file(c.default_cfg_file(a.net))
    if a.scale:
        c.cfg_from_list(["TEST.SCALES", "(%d,)" % a.scale, "TRAIN.SCALES", "(%d,)" % a.scale])
    if a.set_cfgs:
        c.cfg_from_list(a.set_cfgs)
    np.random.seed(c.cfg.RNG_SEED)
    c.cfg.TRAIN.USE_FLIPPED = False
    imdb, roidb, ratio_list, ratio_index = combined_roidb(a.imdbval_name, False)
    print("%d roidb entries" % len(roidb))
    net = train.build_sgg_net(101 if a.net == "res101" else 50, a.num_relations, a.num_classes, device=dev)
    if a.load_name:
        load_reference_state(net, torch.load(a.load_name, map_location="cpu")["model"], strict=False)
        print("load checkpoint %s" % a.load_name)
    if a.target_gt_rels_path:
        with open(a.target_gt_rels_path, "rb") as f:
            net.vrd.target_gt_rels = pickle.load(f, encoding="bytes")
    elif hasattr(imdb, "gt_rels"):
        net.vrd.target_gt_rels = imdb.gt_rels(a.num_relations)
    else:
        raise SystemExit("no relation annotations: pass --target_gt_rels_path")
    net.eval()
    u8 = a.device_prep and a.frames > 1
    dataset = roibatchLoader(roidb, ratio_list, ratio_index, 1, imdb.num_classes, training=False, normalize=False, device_prep=u8)
    loader = torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=False, num_workers=a.num_workers, pin_memory=True)
    key = lambda path: path.split("/")[-1]                  # the annotation dicts are keyed by the frame's file name
    results = {}
    t0 = time.time()
    if a.frames <= 1:
        for data in loader:
            path = data[4][0]
            results[path] = ev.relation_frame(net, data[0].to(dev), data[1].to(dev), key(path))[1]
    else:
        step = ev.RelationStep(net, frames=a.frames, device=dev, use_graph=not a.no_graph)
        groups, order = {}, []

        def pack(g):
            order.append([t[0] for t in g])
            if u8:                                       # (frames as decoded, their meta rows, annotation keys)
                return [t[1] for t in g], torch.cat([t[2] for t in g]), [key(t[0]) for t in g]
            return torch.cat([t[1] for t in g]), torch.cat([t[2] for t in g]).numpy(), [key(t[0]) for t in g]

        def batches():
            for data in loader:
                size = (int(data[1][0][1]), int(data[1][0][2])) if u8 else tuple(data[0].shape[2:])
                g = groups.setdefault(size, [])
                g.append((data[4][0], data[0], data[1]))
                if len(g) == a.frames:
                    yield pack(g)
                    g.clear()
            for g in groups.values():
                if g:
                    yield pack(g)

        for k, res in enumerate(step.run(batches(), u8=u8)):
            for path, r in zip(order[k], res):
                results[path] = r
    dt = time.time() - t0
    n_trip = sum(len(r[1]) for r in results.values() if r[1] is not None)
    print("relation scoring: %d frames, %d triplets, %.2f ms per frame (%.1f frames/s)" % (
        len(results), n_trip, 1e3 * dt / max(len(results), 1), len(results) / max(dt, 1e-9)))
    out_dir = os.path.join(a.output_dir, a.net, a.dataset)
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "relations.pkl")
    with open(out, "wb") as f:
        pickle.dump(results, f, pickle.HIGHEST_PROTOCOL)
    print("wrote %s" % out)
    return results


if __name__ == "__main__":
    main()
