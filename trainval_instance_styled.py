#!/usr/bin/env python3
"""Detector + instance / style discriminator training loop with the reference's CLI surface
(trainval_net_instance_styleD_bilinear.py:35-120 flags, :225-436 loop), on the HIP path.

No dataset is reachable offline, so source / target frames and GT boxes come from the seeded synthetic source
(i2vsgg_amd.synthetic); ``InstanceStyleDStep.stage`` takes one roi_data_layer batch (data, im_info, gt_boxes, num_boxes) per
domain, which is what the reference's two ``roibatchLoader`` iterators yield.  Flags keep the reference names
(lib/model/utils/parser_func.py): --net, --bs, --epochs, --lr, --lr_decay_step, --lr_decay_gamma, --eta, --eta_style,
--style_lambda, --ic, --gc, --cr, --cag, --s, --r, --checksession, --checkepoch, --disp_interval, --save_dir.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")     # before HIP initialises: i2vsgg_amd/__init__.py

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Train the instance_styleD detector + discriminators on MI355X")
    p.add_argument("--net", default="res101", choices=["res101", "res50"])
    p.add_argument("--bs", dest="batch_size", type=int, default=4, help="source frames per step and GPU (as many target frames)")
    p.add_argument("--epochs", dest="max_epochs", type=int, default=1)
    p.add_argument("--iters_per_epoch", type=int, default=10)
    p.add_argument("--lr", type=float, default=5e-4)
    p.add_argument("--lr_decay_step", type=int, default=5)
    p.add_argument("--lr_decay_gamma", type=float, default=0.1)
    p.add_argument("--eta", type=float, default=0.1)
    p.add_argument("--eta_style", type=float, default=0.001)
    p.add_argument("--style_lambda", type=float, default=1.0)
    p.add_argument("--ic", action="store_true", help="instance-level context vector")
    p.add_argument("--gc", action="store_true", help="image-level (style) context vector")
    p.add_argument("--cr", action="store_true", help="consistency regularisation between the two discriminators")
    p.add_argument("--cag", dest="class_agnostic", action="store_true")
    p.add_argument("--o", dest="optimizer", default="sgd", choices=["sgd"])
    p.add_argument("--cuda", action="store_true", default=True)
    p.add_argument("--disp_interval", type=int, default=5)
    p.add_argument("--height", type=int, default=600)
    p.add_argument("--width", type=int, default=1000)
    p.add_argument("--save_dir", default="models", help="checkpoints go to <save_dir>/<net>/<dataset> (reference layout)")
    p.add_argument("--dataset", default="synthetic")
    p.add_argument("--s", dest="session", type=int, default=1)
    p.add_argument("--r", dest="resume", action="store_true", help="resume from --checksession / --checkepoch")
    p.add_argument("--checksession", type=int, default=1)
    p.add_argument("--checkepoch", type=int, default=1)
    p.add_argument("--no-save", action="store_true")
    p.add_argument("--no-graph", action="store_true", help="eager launches (host-side target sampling from np.random, the "
                                                           "reference's RNG contract) instead of the captured step")
    p.add_argument("--set", dest="set_cfgs", nargs=argparse.REMAINDER, default=None)
    return p.parse_args(argv)


def checkpoint_name(a, session, epoch):
    """trainval_net_instance_styleD_bilinear.py:421-423: <save_dir>/<net>/<dataset>/instance_styleD_session_{s}_epoch_{e}.pth"""
    return os.path.join(a.save_dir, a.net, a.dataset, "instance_styleD_session_%d_epoch_%d.pth" % (session, epoch))


def save_checkpoint(a, net, opt, epoch):
    """The reference's per-epoch dict (:424-434; net_utils.py:119-120), the model under the reference's state_dict keys."""
    path = checkpoint_name(a, a.session, epoch)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    from i2vsgg_amd.model.utils.config import cfg
    torch.save({"session": a.session, "epoch": epoch, "model": {k: v.detach().cpu() for k, v in net.state_dict().items()},
                "optimizer": opt.state_dict(), "pooling_mode": cfg.POOLING_MODE, "class_agnostic": a.class_agnostic}, path)
    return path


def load_checkpoint(a, net, opt):
    path = checkpoint_name(a, a.checksession, a.checkepoch)
    ck = torch.load(path, map_location="cpu")
    from i2vsgg_amd.model.faster_rcnn.layers import load_reference_state
    load_reference_state(net, ck["model"], strict=False)
    opt.load_state_dict(ck["optimizer"])
    opt.bump()
    return ck["epoch"], path


def main(argv=None):
    a = parse_args(argv)
    from i2vsgg_amd import parallel, train
    from i2vsgg_amd.model.utils import config as c
    rank, world, dev = parallel.init_from_env()
    c.cfg_from_file(c.default_cfg_file(a.net))
    c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30"])   # parser_func.py:198-199
    if a.set_cfgs:
        c.cfg_from_list(a.set_cfgs)
    np.random.seed(c.cfg.RNG_SEED + rank)
    torch.manual_seed(c.cfg.RNG_SEED + rank)
    net = train.build_instance_styled_net(101 if a.net == "res101" else 50, device=dev, ic=a.ic, gc=a.gc,
                                          class_agnostic=a.class_agnostic)
    lr = a.lr
    seed_of = lambda epoch, it: 1000 * epoch + it * world + rank           # the two data loaders' role is played by reseed()
    step = train.InstanceStyleDStep(net, a.batch_size, lr=lr, eta=a.eta, eta_style=a.eta_style, style_lambda=a.style_lambda,
                                    seed=seed_of(1, 0), device=dev, h=a.height, w=a.width, cr=a.cr)
    start_epoch = 1
    if a.resume:
        done, path = load_checkpoint(a, net, step.opt)
        start_epoch = done + 1
        for e in range(2, start_epoch + 1):
            if (e - 1) % a.lr_decay_step == 0:
                lr *= a.lr_decay_gamma
        if rank == 0:
            print("resumed %s (epoch %d)" % (path, done))
    if start_epoch > a.max_epochs:
        return
    step.reseed(seed_of(start_epoch, 0))
    graphed = False
    if not a.no_graph and dev.type == "cuda":
        # the loop runs the benchmarked step: both forwards, the backward, the gradient exchange and the update as ONE HIP
        # graph with device-side target sampling; the capture's warm-up steps leave no trace in parameters / momentum / RNG
        graphed = step.capture(warmup=2, restore=True)
    if rank == 0:
        print("step: %s" % ("HIP graph" if graphed else "eager (%s)" % (step.graph_error or "--no-graph")))
    for epoch in range(start_epoch, a.max_epochs + 1):
        if epoch > 1 and (epoch - 1) % a.lr_decay_step == 0 and epoch != start_epoch:
            lr *= a.lr_decay_gamma                           # adjust_learning_rate (net_utils.py:113-116)
            step.opt.scale_lr(a.lr_decay_gamma)
            if graphed:
                graphed = step.capture(warmup=0)
        t0 = time.time()
        acc = {k: torch.zeros((), device=dev) for k in step.names}
        for it in range(a.iters_per_epoch):
            if it or epoch != start_epoch:
                step.reseed(seed_of(epoch, it))              # queued behind the running step on the same stream
            step()
            for k in step.names:
                acc[k] += step.losses[k]
            if (it + 1) % a.disp_interval == 0:
                vals = {k: float(v) / a.disp_interval for k, v in acc.items()}      # the only host synchronisation of the loop
                for v in acc.values():
                    v.zero_()
                if rank == 0:
                    dt = time.time() - t0
                    print("[session %d][epoch %2d][iter %4d/%4d] loss: %.4f, lr: %.2e, %.1f frames/s" % (
                        a.session, epoch, it + 1, a.iters_per_epoch, vals["total"], lr,
                        world * 2 * a.batch_size * a.disp_interval / dt))
                    print("\t\t\tdet %.4f  dloss s: %.4f dloss t: %.4f dloss s style: %.4f dloss t style: %.4f eta: %.4f" % (
                        vals["det"], vals["dloss_s"], vals["dloss_t"], vals["dloss_s_style"], vals["dloss_t_style"], a.eta))
                t0 = time.time()
        if not a.no_save and rank == 0:
            print("save model: %s" % save_checkpoint(a, net, step.opt, epoch))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
