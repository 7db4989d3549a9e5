#!/usr/bin/env python3
"""Relation-head training loop with the reference's CLI surface (trainval_net_SGG_emb.py:189-320), on the HIP path.

No dataset is reachable offline, so frames / annotations come from the seeded synthetic source
(i2vsgg_amd.synthetic); with a real imdb registered through roi_data_layer.roidb.register_imdb the same loop
runs on it.  Flags keep the reference names (lib/model/utils/parser_func.py): --net, --bs, --epochs, --lr,
--vrd_lr, --lr_decay_step, --lr_decay_gamma, --o, --num_classes, --num_relations, --cuda, --disp_interval.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description="Train the SGG_emb relation head (pre_det) on MI355X")
    p.add_argument("--net", default="res101", choices=["res101", "res50"])
    p.add_argument("--bs", dest="batch_size", type=int, default=2, help="frames per step (per GPU)")
    p.add_argument("--epochs", dest="max_epochs", type=int, default=1)
    p.add_argument("--iters_per_epoch", type=int, default=10)
    p.add_argument("--lr", type=float, default=1.0)
    p.add_argument("--vrd_lr", type=float, default=1e-4)
    p.add_argument("--lr_decay_step", type=int, default=1)
    p.add_argument("--lr_decay_gamma", type=float, default=0.9)
    p.add_argument("--o", dest="optimizer", default="sgd", choices=["sgd"])
    p.add_argument("--num_classes", type=int, default=16)
    p.add_argument("--num_relations", type=int, default=62)
    p.add_argument("--vrd_task", default="pre_det")
    p.add_argument("--cuda", action="store_true", default=True)
    p.add_argument("--disp_interval", type=int, default=5)
    p.add_argument("--height", type=int, default=600)
    p.add_argument("--width", type=int, default=1000)
    p.add_argument("--set", dest="set_cfgs", nargs=argparse.REMAINDER, default=None)
    return p.parse_args()


def main():
    a = parse_args()
    from i2vsgg_amd import parallel, train
    from i2vsgg_amd.model.utils import config as c
    rank, world, dev = parallel.init_from_env()
    c.cfg_from_file(c.default_cfg_file(a.net))
    c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30"])
    if a.set_cfgs:
        c.cfg_from_list(a.set_cfgs)
    np.random.seed(c.cfg.RNG_SEED)
    net = train.build_sgg_net(101 if a.net == "res101" else 50, a.num_relations, a.num_classes, device=dev)
    vrd_lr = a.vrd_lr
    # eager steps: the minibatch changes every iteration (the data layer's role is played by reseed())
    if dev.type == "cuda":
        torch.cuda.set_stream(torch.cuda.Stream(dev))      # not the legacy default stream (train.SGGEmbStep.__call__)
    step = train.SGGEmbStep(net, a.batch_size, vrd_lr=vrd_lr, seed=rank, device=dev, h=a.height, w=a.width,
                            use_graph=False, fuse_sgd=False)
    for epoch in range(1, a.max_epochs + 1):
        if epoch > 1 and (epoch - 1) % a.lr_decay_step == 0:
            vrd_lr *= a.lr_decay_gamma                       # adjust_learning_rate (net_utils.py:113-116)
            step.opt.scale_lr(a.lr_decay_gamma)
        t0, acc = time.time(), 0.0
        for it in range(a.iters_per_epoch):
            step.reseed(1000 * epoch + it * world + rank)
            acc += float(step())
            if (it + 1) % a.disp_interval == 0 and rank == 0:
                dt = time.time() - t0
                print("[epoch %2d][iter %4d/%4d] loss: %.4f, vrd_lr: %.2e, %.1f frames/s" % (
                    epoch, it + 1, a.iters_per_epoch, acc / a.disp_interval, vrd_lr,
                    world * a.batch_size * a.disp_interval / dt))
                t0, acc = time.time(), 0.0
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
