#!/usr/bin/env python3
"""Detection test loop with the reference's structure (test_net_instance_styleD_bilinear.py:42-234) on the HIP path:
``combined_roidb(imdbval_name, False) -> roibatchLoader(training=False, normalize=False) -> DataLoader(batch_size=1)``, the
checkpoint in the reference's layout, ``all_boxes[class][image]`` pickled to ``<output_dir>/detections.pkl``.

The per-frame body (:140-221) runs frame by frame (``--frames 1``: ``eval.detect_frame``) or, by default, four frames at a time
as one replayed HIP graph with a branch per frame (``eval.DetectStep``: same results, ~1.9x the frames/s).  Test frames keep
their own size (the loader pads nothing at batch_size 1), so frames are grouped by size and a group runs when it is full;
what is left at the end runs as short batches.  ``imdb.evaluate_detections`` is called when the imdb has one (the datasets
package is outside this repo's scope; the synthetic imdb has none)."""
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
    p = argparse.ArgumentParser(description="Evaluate the instance_styleD detector on MI355X")
    p.add_argument("--dataset", default="synthetic")
    p.add_argument("--imdbval_name", default="synthetic_16_v", help="test roidb (combined_roidb name)")
    p.add_argument("--net", default="res101", choices=["res101", "res50"])
    p.add_argument("--load_name", default="", help="checkpoint in the reference's layout ({'model': state_dict, 'pooling_mode': ...}); "
                                                   "empty: random-init weights (no checkpoint is reachable offline)")
    p.add_argument("--ic", action="store_true")
    p.add_argument("--gc", action="store_true")
    p.add_argument("--cag", dest="class_agnostic", action="store_true")
    p.add_argument("--nw", dest="num_workers", type=int, default=0)
    p.add_argument("--frames", type=int, default=4, help="frames per replayed graph (1: frame by frame, eager launches)")
    p.add_argument("--scale", type=int, default=0, help="shorter image side (cfg.TEST.SCALES; 0: the yml's 600)")
    p.add_argument("--output_dir", default="output")
    p.add_argument("--max_per_image", type=int, default=100)
    p.add_argument("--thresh", type=float, default=0.0)
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
    c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30"])   # parser_func.py:198-199
    if a.scale:
        c.cfg_from_list(["TEST.SCALES", "(%d,)" % a.scale, "TRAIN.SCALES", "(%d,)" % a.scale])
    if a.set_cfgs:
        c.cfg_from_list(a.set_cfgs)
    np.random.seed(c.cfg.RNG_SEED)
    c.cfg.TRAIN.USE_FLIPPED = False                                             # :60
    imdb, roidb, ratio_list, ratio_index = combined_roidb(a.imdbval_name, False)
    if hasattr(imdb, "competition_mode"):
        imdb.competition_mode(on=True)
    print("%d roidb entries" % len(roidb))

    net = train.build_instance_styled_net(101 if a.net == "res101" else 50, n_cls=imdb.num_classes, device=dev, ic=a.ic, gc=a.gc,
                                          class_agnostic=a.class_agnostic)
    if a.load_name:
        ck = torch.load(a.load_name, map_location="cpu")                        # :75-81
        load_reference_state(net, ck["model"], strict=False)
        if "pooling_mode" in ck:
            c.cfg.POOLING_MODE = ck["pooling_mode"]
        print("load checkpoint %s" % a.load_name)
    net.eval()

    num_images = len(roidb)
    all_boxes = [[[] for _ in range(num_images)] for _ in range(imdb.num_classes)]          # :115-116
    empty = np.zeros((0, 5), np.float32)
    u8 = a.device_prep and a.frames > 1
    dataset = roibatchLoader(roidb, ratio_list, ratio_index, 1, imdb.num_classes, training=False, normalize=False, device_prep=u8)
    loader = torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=False, num_workers=a.num_workers, pin_memory=True)

    def keep(i, per_class):
        for j in range(1, imdb.num_classes):
            all_boxes[j][i] = per_class[j] if len(per_class[j]) else empty

    t0 = time.time()
    if a.frames <= 1:
        z, nb = torch.zeros(1, 1, 5, device=dev), torch.zeros(1, device=dev)
        for i, data in enumerate(loader):
            keep(i, ev.detect_frame(net, data[0].to(dev), data[1].to(dev), z, nb, thresh=a.thresh, max_per_image=a.max_per_image,
                                    class_agnostic=a.class_agnostic))
    else:
        step = ev.DetectStep(net, frames=a.frames, thresh=a.thresh, max_per_image=a.max_per_image, class_agnostic=a.class_agnostic,
                             device=dev, use_graph=not a.no_graph)
        groups, order = {}, []

        def pack(g):
            order.append([t[0] for t in g])
            if u8:                                       # (frames as decoded, their meta rows)
                return [t[1] for t in g], torch.cat([t[2] for t in g])
            return torch.cat([t[1] for t in g]), torch.cat([t[2] for t in g])

        def batches():
            # frames of one (resized) size travel together; a group leaves when it is full, the rest at the end
            for i, data in enumerate(loader):
                size = (int(data[1][0][1]), int(data[1][0][2])) if u8 else tuple(data[0].shape[2:])
                g = groups.setdefault(size, [])
                g.append((i, data[0], data[1]))
                if len(g) == a.frames:
                    yield pack(g)
                    g.clear()
            for g in groups.values():
                if g:
                    yield pack(g)

        for k, res in enumerate(step.run(batches(), u8=u8)):
            for i, per_class in zip(order[k], res):
                keep(i, per_class)
    dt = time.time() - t0
    n_det = sum(len(all_boxes[j][i]) for j in range(1, imdb.num_classes) for i in range(num_images))
    print("im_detect: %d images, %d detections, %.2f ms per image (%.1f frames/s)" % (num_images, n_det, 1e3 * dt / max(num_images, 1),
                                                                                     num_images / max(dt, 1e-9)))
    out_dir = os.path.join(a.output_dir, a.net, a.dataset)
    os.makedirs(out_dir, exist_ok=True)
    det_file = os.path.join(out_dir, "detections.pkl")
    with open(det_file, "wb") as f:
        pickle.dump(all_boxes, f, pickle.HIGHEST_PROTOCOL)                      # :230-231
    print("wrote %s" % det_file)
    if hasattr(imdb, "evaluate_detections"):
        imdb.evaluate_detections(all_boxes, out_dir)                            # :233-234
    return all_boxes


if __name__ == "__main__":
    main()
