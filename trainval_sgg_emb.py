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
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")     # before HIP initialises: i2vsgg_amd/__init__.py

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
    p.add_argument("--save_dir", default="models", help="checkpoints go to <save_dir>/<net>/<dataset> (reference layout)")
    p.add_argument("--dataset", default="synthetic")
    p.add_argument("--s", dest="session", type=int, default=1)
    p.add_argument("--r", dest="resume", action="store_true", help="resume from --checksession / --checkepoch")
    p.add_argument("--checksession", type=int, default=1)
    p.add_argument("--checkepoch", type=int, default=1)
    p.add_argument("--no-save", action="store_true")
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of the captured step")
    p.add_argument("--set", dest="set_cfgs", nargs=argparse.REMAINDER, default=None)
    return p.parse_args()


def checkpoint_name(a, session, epoch):
    return os.path.join(a.save_dir, a.net, a.dataset, "SGG_emb_%d_%d.pth" % (session, epoch))


def save_checkpoint(a, net, opt, epoch, rank):
    """The reference's per-epoch dict (trainval_net_instance_styleD_bilinear.py:421-434; net_utils.py:119-120) with
    the model under the reference's state_dict keys.  A column-parallel fc6 is reassembled first (every rank takes part
    in the gather, rank 0 writes), so the file loads on any number of GPUs and under the reference's layer shapes."""
    w6, b6 = net.vrd.gather_fc6()
    if rank != 0:
        return None
    model = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    model["vrd.fc6.fc.weight"], model["vrd.fc6.fc.bias"] = w6.detach().cpu(), b6.detach().cpu()
    path = checkpoint_name(a, a.session, epoch)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"session": a.session, "epoch": epoch, "model": model, "optimizer": opt.state_dict(),
                "pooling_mode": "pool", "class_agnostic": False}, path)
    return path


def load_checkpoint(a, net, opt, dev):
    path = checkpoint_name(a, a.checksession, a.checkepoch)
    ck = torch.load(path, map_location="cpu")
    model = dict(ck["model"])
    if net.vrd.tp is not None:                   # this run cuts fc6 by columns: keep this rank's shard
        rk, world = net.vrd.tp
        n = model["vrd.fc6.fc.weight"].shape[0] // world
        model["vrd.fc6.fc.weight"] = model["vrd.fc6.fc.weight"][rk * n:(rk + 1) * n]
        model["vrd.fc6.fc.bias"] = model["vrd.fc6.fc.bias"][rk * n:(rk + 1) * n]
    from i2vsgg_amd.model.faster_rcnn.layers import load_reference_state
    load_reference_state(net, model, strict=False)
    sd = ck["optimizer"]
    if net.vrd.tp is not None:
        rk, world = net.vrd.tp
        for i, g in enumerate(sd["param_groups"]):
            if g.get("name") in ("vrd.fc6.fc.weight", "vrd.fc6.fc.bias"):
                m = sd["state"][g["params"][0]]["momentum_buffer"]
                n = m.shape[0] // world
                sd["state"][g["params"][0]]["momentum_buffer"] = m[rk * n:(rk + 1) * n]
    opt.load_state_dict(sd)
    return ck["epoch"], path


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
    seed_of = lambda epoch, it: 1000 * epoch + it * world + rank           # the data layer's role is played by reseed()
    step = train.SGGEmbStep(net, a.batch_size, vrd_lr=vrd_lr, seed=seed_of(1, 0), device=dev, h=a.height, w=a.width,
                            use_graph=not a.no_graph and dev.type == "cuda")
    start_epoch = 1
    if a.resume:
        done, path = load_checkpoint(a, net, step.opt, dev)
        start_epoch = done + 1
        for _ in range(done // a.lr_decay_step):
            vrd_lr *= a.lr_decay_gamma
        if rank == 0:
            print("resumed %s (epoch %d)" % (path, done))
    # batches in training order; the first one is staged before the capture (the overlapped pipeline is primed with it)
    seeds = [seed_of(e, it) for e in range(start_epoch, a.max_epochs + 1) for it in range(a.iters_per_epoch)]
    if not seeds:
        return
    step.reseed(seeds[0])
    # the step the loop runs IS the benchmarked one: one captured graph per step, fused wgrad+SGD, head of batch k beside
    # the backbone of batch k+1.  The warm-up steps of the capture leave no trace in parameters, momentum or RNG state
    graphed = step.capture(warmup=2, restore=True)
    if rank == 0:
        print("step: %s" % ("HIP graph, overlapped" if graphed and step.overlap else "HIP graph" if graphed
                            else "eager (%s)" % step.graph_error))
    lag = 1 if (graphed and step.overlap) else 0        # overlapped: call k trains batch k while the backbone runs batch k+1
    pos = 0
    for epoch in range(start_epoch, a.max_epochs + 1):
        if epoch > 1 and (epoch - 1) % a.lr_decay_step == 0 and epoch != start_epoch:
            vrd_lr *= a.lr_decay_gamma                       # adjust_learning_rate (net_utils.py:113-116)
            step.opt.scale_lr(a.lr_decay_gamma)
            if graphed:
                graphed = step.capture(warmup=0)             # rates live in the captured kernel arguments
        t0, acc = time.time(), torch.zeros((), device=dev)
        for it in range(a.iters_per_epoch):
            ahead = pos + lag                                # the batch this call's backbone pass works on
            if ahead < len(seeds) and ahead > 0:
                step.reseed(seeds[ahead])                    # queued behind the running step, no host synchronisation
            acc += step()                                    # loss of batch ``pos``
            pos += 1
            if (it + 1) % a.disp_interval == 0:
                loss = float(acc) / a.disp_interval          # the only host synchronisation of the loop
                acc.zero_()
                if rank == 0:
                    dt = time.time() - t0
                    print("[epoch %2d][iter %4d/%4d] loss: %.4f, vrd_lr: %.2e, %.1f frames/s" % (
                        epoch, it + 1, a.iters_per_epoch, loss, vrd_lr, world * a.batch_size * a.disp_interval / dt))
                t0 = time.time()
        if not a.no_save:
            path = save_checkpoint(a, net, step.opt, epoch, rank)
            if rank == 0:
                print("save model: %s" % path)
    step.opt.unfuse()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
