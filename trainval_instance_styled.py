#!/usr/bin/env python3
"""Detector + instance / style discriminator training loop with the reference's CLI surface and data path
(trainval_net_instance_styleD_bilinear.py:35-120 flags, :73-97 loaders, :225-436 loop), on the HIP path: two
``combined_roidb -> roibatchLoader -> DataLoader(sampler)`` chains (source and target domain) feed
``InstanceStyleDStep.stage_batch`` with exactly what the reference loop copies into its holders (:258-291).  Minibatches
differ in size (each loader pads a batch to its own aspect ratio); the step captures one HIP graph per (source size,
target size) pair the first time it meets it.

No dataset is reachable offline: ``--imdb_name`` / ``--imdb_name_target`` default to seeded synthetic imdbs whose frames
come in five resolutions (roi_data_layer.roidb.SyntheticImdb); a real imdb registered with ``register_imdb`` runs through
the same loop.  Flags keep the reference names (lib/model/utils/parser_func.py): --dataset, --dataset_t, --net, --bs, --nw,
--start_epoch, --epochs, --lr, --lr_decay_step, --lr_decay_gamma, --eta, --eta_style, --style_lambda, --ic, --gc, --cr,
--cag, --s, --r, --load_name, --disp_interval, --save_dir.
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
    p.add_argument("--dataset", default="synthetic")
    p.add_argument("--dataset_t", default="synthetic_t")
    p.add_argument("--imdb_name", default="synthetic_64_v", help="source roidb (combined_roidb name; a+b concatenates)")
    p.add_argument("--imdb_name_target", default="synthetic_40_v_7", help="target roidb")
    p.add_argument("--net", default="res101", choices=["res101", "res50"])
    p.add_argument("--bs", dest="batch_size", type=int, default=4, help="source frames per step and GPU (as many target frames)")
    p.add_argument("--nw", dest="num_workers", type=int, default=0)
    p.add_argument("--start_epoch", type=int, default=1)
    p.add_argument("--epochs", dest="max_epochs", type=int, default=1)
    p.add_argument("--iters_per_epoch", type=int, default=0, help="0: train_size / bs as the reference (:200)")
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
    p.add_argument("--o", dest="optimizer", default="sgd", choices=["sgd", "adam"],
                   help="adam: torch.optim.Adam's update on the same param groups (train.FusedAdam); no fusion into the "
                        "filter-gradient kernels")
    p.add_argument("--cuda", action="store_true", default=True)
    p.add_argument("--disp_interval", type=int, default=5)
    p.add_argument("--scale", type=int, default=0, help="shorter image side (cfg.TRAIN.SCALES; 0: the yml's 600)")
    p.add_argument("--save_dir", default="models", help="checkpoints go to <save_dir>/<net>/<dataset> (reference layout)")
    p.add_argument("--s", dest="session", type=int, default=1)
    p.add_argument("--r", dest="resume", action="store_true", help="resume from --load_name (or --checksession / --checkepoch)")
    p.add_argument("--load_name", default="")
    p.add_argument("--checksession", type=int, default=1)
    p.add_argument("--checkepoch", type=int, default=1)
    p.add_argument("--no-save", action="store_true")
    p.add_argument("--device_prep", action="store_true",
                   help="the loaders hand over uint8 frames as decoded; BGR swap, mean subtraction, resize and batch padding run on "
                        "the GPU (roibatchLoader(device_prep=True) + stage_batch_u8).  Minibatches the reference would crop to a "
                        "square (target ratio exactly 1) are skipped in this mode")
    p.add_argument("--no-graph", action="store_true", help="eager launches (host-side target sampling from np.random, the "
                                                           "reference's RNG contract) instead of the captured step")
    p.add_argument("--set", dest="set_cfgs", nargs=argparse.REMAINDER, default=None)
    return p.parse_args(argv)


def checkpoint_name(a, session, epoch):
    """trainval_net_instance_styleD_bilinear.py:421-426: <save_dir>/<net>/<dataset>/instance_pixel_styleD_bilinear_cr_{cr}_
    source_{dataset}_target_{dataset_t}_session_{s}_lr_{lr}_epoch_{e}_bs_{bs}_mscoco.pth"""
    return os.path.join(a.save_dir, a.net, a.dataset,
                        "instance_pixel_styleD_bilinear_cr_%s_source_%s_target_%s_session_%d_lr_%s_epoch_%d_bs_%d_mscoco.pth" % (
                            a.cr, a.dataset, a.dataset_t, session, a.lr, epoch, a.batch_size))


def save_checkpoint(a, net, opt, epoch):
    """The reference's per-epoch dict (:427-434; net_utils.py:119-120): ``epoch`` holds the NEXT epoch, the model sits under
    the reference's state_dict keys, the optimizer state in torch.optim.SGD's layout."""
    path = checkpoint_name(a, a.session, epoch)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    from i2vsgg_amd.model.utils.config import cfg
    torch.save({"session": a.session, "epoch": epoch + 1, "model": {k: v.detach().cpu() for k, v in net.state_dict().items()},
                "optimizer": opt.state_dict(), "pooling_mode": cfg.POOLING_MODE, "class_agnostic": a.class_agnostic}, path)
    return path


def load_checkpoint(path, net, opt):
    """:186-197: model, optimizer (its rates carry every decay applied so far) -> the epoch to start at."""
    ck = torch.load(path, map_location="cpu")
    from i2vsgg_amd.model.faster_rcnn.layers import load_reference_state
    load_reference_state(net, ck["model"], strict=False)
    opt.load_state_dict(ck["optimizer"])
    opt.bump()
    return int(ck["epoch"])


WO_PARAMETER = ("netD_pixel", "RPN_cls_score", "RPN_bbox_pred", "RCNN_cls_score", "RCNN_bbox_pred")


def init_from_detector(path, net):
    """trainval_net_instance_styleD_bilinear.py:153-183: initialise from a plain detector checkpoint (the reference takes this
    branch when 'faster_rcnn' is in --load_name).  Allow-list = the model's own keys that contain none of ``WO_PARAMETER``
    (:153-161: the instance discriminator and the four class- / anchor-count-dependent output layers keep their
    initialisation); of the file's ``model`` dict exactly the allow-listed keys are taken (:172), keys the file lacks
    (netD_style.* of a plain Faster R-CNN) keep their initialisation -- no KeyError.  Session, epoch and optimizer state are
    not used (:176-180 are commented out), and ``pooling_mode`` is looked up in the MODEL dict (:181, after ``checkpoint`` was
    rebound to it at :166), i.e. never found: the file's pooling mode is not taken in this mode.  -> loaded keys"""
    from i2vsgg_amd.model.faster_rcnn.layers import load_reference_state
    model = torch.load(path, map_location="cpu")["model"]
    own = net.state_dict()
    allow = [k for k in own if not any(tag in k for tag in WO_PARAMETER)]
    loaded = {k: v for k, v in model.items() if k in allow}
    for k, v in loaded.items():
        if tuple(v.shape) != tuple(own[k].shape):
            raise ValueError("init_from_detector: %s is %s in %s, %s here" % (k, tuple(v.shape), path, tuple(own[k].shape)))
    load_reference_state(net, loaded, strict=False)
    return sorted(loaded)


def main(argv=None):
    a = parse_args(argv)
    from i2vsgg_amd import parallel, train
    from i2vsgg_amd.model.utils import config as c
    from i2vsgg_amd.model.utils.net_utils import sampler
    from i2vsgg_amd.roi_data_layer.roibatchLoader import collate_device_prep, roibatchLoader
    from i2vsgg_amd.roi_data_layer.roidb import combined_roidb
    rank, world, dev = parallel.init_from_env()
    c.cfg_from_file(c.default_cfg_file(a.net))
    c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30"])   # parser_func.py:198-199
    if a.scale:
        c.cfg_from_list(["TRAIN.SCALES", "(%d,)" % a.scale])
    if a.set_cfgs:
        c.cfg_from_list(a.set_cfgs)
    np.random.seed(c.cfg.RNG_SEED + rank)
    torch.manual_seed(c.cfg.RNG_SEED + rank)

    # ---- the data path of the reference loop (:70-97): one loader per domain
    c.cfg.TRAIN.USE_FLIPPED = True
    imdb, roidb, ratio_list, ratio_index = combined_roidb(a.imdb_name)
    imdb_t, roidb_t, ratio_list_t, ratio_index_t = combined_roidb(a.imdb_name_target)
    train_size, train_size_t = len(roidb), len(roidb_t)
    if rank == 0:
        print("%d source roidb entries\n%d target roidb entries" % (train_size, train_size_t))
    pin = dev.type == "cuda"
    mk = lambda rdb, rl, ri, n, seed: torch.utils.data.DataLoader(
        roibatchLoader(rdb, rl, ri, a.batch_size, imdb.num_classes, training=True, device_prep=a.device_prep), batch_size=a.batch_size,
        sampler=sampler(n, a.batch_size, rank=rank, world=world, seed=seed), num_workers=a.num_workers, pin_memory=pin,
        collate_fn=collate_device_prep if a.device_prep else None)
    dataloader_s = mk(roidb, ratio_list, ratio_index, train_size, c.cfg.RNG_SEED)
    dataloader_t = mk(roidb_t, ratio_list_t, ratio_index_t, train_size_t, c.cfg.RNG_SEED + 1)
    iters_per_epoch = a.iters_per_epoch or (train_size // a.batch_size // world)

    net = train.build_instance_styled_net(101 if a.net == "res101" else 50, n_cls=imdb.num_classes, device=dev, ic=a.ic, gc=a.gc,
                                          class_agnostic=a.class_agnostic)
    step = train.InstanceStyleDStep(net, a.batch_size, lr=a.lr, eta=a.eta, eta_style=a.eta_style, style_lambda=a.style_lambda,
                                    device=dev, cr=a.cr, stage_synthetic=False, optimizer=a.optimizer)
    start_epoch = a.start_epoch
    if a.resume and "faster_rcnn" in a.load_name:            # :163: model initialisation with an object-detection checkpoint
        if rank == 0:
            print("loading checkpoint %s" % a.load_name)
        loaded = init_from_detector(a.load_name, net)
        step.opt.bump()
        if rank == 0:
            print("loaded checkpoint %s (%d tensors through the allow-list)" % (a.load_name, len(loaded)))
    elif a.resume:                                           # :186: resume a run of this script
        path = a.load_name or checkpoint_name(a, a.checksession, a.checkepoch)
        start_epoch = load_checkpoint(path, net, step.opt)
        if rank == 0:
            print("loaded checkpoint %s (start epoch %d)" % (path, start_epoch))
    lr = step.opt.lr_of("RCNN_rpn.RPN_Conv.weight")          # :194: lr = optimizer.param_groups[0]['lr']
    if start_epoch > a.max_epochs:
        return

    iters = {"s": iter(dataloader_s), "t": iter(dataloader_t)}

    def draw(which, loader):
        try:
            return next(iters[which])
        except StopIteration:
            iters[which] = iter(loader)
            return next(iters[which])

    def stage_next():
        """The next (source, target) pair of minibatches the reference loop would train on (:239-256)."""
        for _ in range(4 * len(dataloader_s) + 4):
            ds_, dt_ = draw("s", dataloader_s), draw("t", dataloader_t)
            if (step.stage_batch_u8(ds_, dt_) if a.device_prep else step.stage_batch(ds_, dt_)):
                return
        raise SystemExit("the data loaders yield no trainable minibatch")

    stage_next()
    graphed = False
    if not a.no_graph and dev.type == "cuda":
        # the loop runs the benchmarked step: both forwards, the backward, the gradient exchange and the update as ONE HIP
        # graph with device-side target sampling; the capture's warm-up steps leave no trace in parameters / momentum / RNG
        graphed = step.capture(warmup=2, restore=True)
    if rank == 0:
        print("step: %s" % ("HIP graph" if graphed else "eager (%s)" % (step.graph_error or "--no-graph")))
    first = True
    for epoch in range(start_epoch, a.max_epochs + 1):
        if epoch > 1 and (epoch - 1) % a.lr_decay_step == 0:
            step.opt.scale_lr(a.lr_decay_gamma)              # adjust_learning_rate (net_utils.py:113-116), :232-234
            lr = step.opt.lr_of("RCNN_rpn.RPN_Conv.weight")
            if graphed:
                graphed = step.capture(warmup=0)             # the rates live in the captured kernel arguments
                if not graphed and rank == 0:
                    print("re-capture failed, eager launches from here: %s" % step.graph_error)
        t0 = time.time()
        acc = {k: torch.zeros((), device=dev) for k in step.names}
        for it in range(iters_per_epoch):
            if not first:
                stage_next()                                 # queued behind the running step on the same stream
            first = False
            step()
            for k in step.names:
                acc[k] += step.losses[k]
            if (it + 1) % a.disp_interval == 0:
                vals = {k: float(v) / a.disp_interval for k, v in acc.items()}      # the only host synchronisation of the loop
                for v in acc.values():
                    v.zero_()
                if rank == 0:
                    dt = time.time() - t0
                    print("[session %d][epoch %2d][iter %4d/%4d] loss: %.4f, lr: %.2e, %.1f frames/s (%d graphs)" % (
                        a.session, epoch, it + 1, iters_per_epoch, vals["total"], lr,
                        world * 2 * a.batch_size * a.disp_interval / dt, sum(1 for d in step.sets.values() if d.graph)))
                    print("\t\t\tdet %.4f  dloss s: %.4f dloss t: %.4f dloss s style: %.4f dloss t style: %.4f eta: %.4f" % (
                        vals["det"], vals["dloss_s"], vals["dloss_t"], vals["dloss_s_style"], vals["dloss_t_style"], a.eta))
                t0 = time.time()
        if not a.no_save and rank == 0:
            print("save model: %s" % save_checkpoint(a, net, step.opt, epoch))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
