#!/usr/bin/env python3
"""Relation-head training loop with the reference's CLI surface and data path (trainval_net_SGG_emb.py:73-320), on the HIP
path: ``combined_roidb -> roibatchLoader(path_return=True) -> DataLoader(sampler)`` feeds the step exactly as the
reference loop is fed (:77-91, :204-217); the step is the captured, overlapped ``train.SGGEmbStep`` (one HIP graph per
frame size, boxes / pairs padded to a capacity), so minibatches may differ in image size and in boxes / pairs per frame.

No dataset is reachable offline: ``--imdb_name synthetic_<n>_v`` (the default) is a seeded imdb whose frames come in five
resolutions and whose annotations (``imdb.gt_rels``: the ``source_gt_rels`` pickle's layout) vary from frame to frame; a
real imdb registered with ``roi_data_layer.roidb.register_imdb`` plus ``--source_gt_rels_path`` runs through the same loop.
Flags keep the reference names (lib/model/utils/parser_func.py): --dataset, --net, --bs, --nw, --start_epoch, --epochs,
--lr, --vrd_lr, --lr_decay_step, --lr_decay_gamma, --o, --num_classes, --num_relations, --s, --r, --load_name,
--adaptation, --save_dir, --disp_interval.
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
    p = argparse.ArgumentParser(description="Train the SGG_emb relation head (pre_det) on MI355X")
    p.add_argument("--dataset", default="synthetic")
    p.add_argument("--imdb_name", default="synthetic_64_v", help="roidb to train on (combined_roidb name; a+b concatenates)")
    p.add_argument("--net", default="res101", choices=["res101", "res50"])
    p.add_argument("--bs", dest="batch_size", type=int, default=2, help="frames per step (per GPU)")
    p.add_argument("--nw", dest="num_workers", type=int, default=0)
    p.add_argument("--start_epoch", type=int, default=1)
    p.add_argument("--epochs", dest="max_epochs", type=int, default=1)
    p.add_argument("--iters_per_epoch", type=int, default=0, help="0: train_size / bs as the reference (:187)")
    p.add_argument("--lr", type=float, default=1.0)
    p.add_argument("--vrd_lr", type=float, default=1e-4)
    p.add_argument("--lr_decay_step", type=int, default=1)
    p.add_argument("--lr_decay_gamma", type=float, default=0.9)
    p.add_argument("--o", dest="optimizer", default="sgd", choices=["sgd", "adam"],
                   help="adam: torch.optim.Adam's update on the same param groups (train.FusedAdam); no fusion into the "
                        "filter-gradient kernels")
    p.add_argument("--num_classes", type=int, default=16)
    p.add_argument("--num_relations", type=int, default=62)
    p.add_argument("--vrd_task", default="pre_det")
    # parser_func.py:155-163,182.  The reference declares the first two ``type=bool`` (any value given on its command line
    # parses to True, i.e. spatial_type 1: SURVEY.md A15); here they are integers and select what they name.  The defaults
    # (1, 2, 300) are what every reference script runs and what the captured step packs; the other variants of the relation
    # head train through the model's own forward on eager launches
    p.add_argument("--use_obj_visual", type=int, default=1, choices=[0, 1], help="subject / object visual embeddings in the fusion")
    p.add_argument("--spatial_type", type=int, default=2, choices=[0, 1, 2], help="0 none, 1 relative location (8-d), 2 dual 32x32 masks")
    p.add_argument("--emb_dim", type=int, default=300, help="dimension of the embedding space (the word vectors' dimension)")
    p.add_argument("--adaptation", default="adap")
    p.add_argument("--source_gt_rels_path", default="")
    p.add_argument("--cuda", action="store_true", default=True)
    p.add_argument("--disp_interval", type=int, default=5)
    p.add_argument("--scale", type=int, default=0, help="shorter image side (cfg.TRAIN.SCALES; 0: the yml's 600)")
    p.add_argument("--save_dir", default="models", help="checkpoints go to <save_dir>/<net>/<dataset> (reference layout)")
    p.add_argument("--s", dest="session", type=int, default=1)
    p.add_argument("--r", dest="resume", action="store_true",
                   help="the reference's --r (trainval_net_SGG_emb.py:155-173): INITIALISE the detector half from the stage-1 "
                        "checkpoint --load_name (a trainval_instance_styled.py / reference detector file): every key without "
                        "'vrd' is taken from it, vrd.* keeps its initialisation, training starts at --start_epoch with a fresh "
                        "optimizer")
    p.add_argument("--resume_train", action="store_true",
                   help="continue an interrupted run of THIS script: model, optimizer and epoch from --load_name (or "
                        "--checksession / --checkepoch).  The reference has no such mode for this script (its --r is the "
                        "detector hand-off above)")
    p.add_argument("--load_name", default="")
    p.add_argument("--checksession", type=int, default=1)
    p.add_argument("--checkepoch", type=int, default=1)
    p.add_argument("--no-save", action="store_true")
    p.add_argument("--device_prep", action="store_true",
                   help="the loaders hand over uint8 frames as decoded; BGR swap, mean subtraction, resize and batch padding run on "
                        "the GPU (roibatchLoader(device_prep=True) + stage_batch_u8).  Minibatches the reference would crop to a "
                        "square (target ratio exactly 1) are skipped in this mode")
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of the captured step")
    p.add_argument("--set", dest="set_cfgs", nargs=argparse.REMAINDER, default=None)
    return p.parse_args(argv)


def checkpoint_name(a, session, epoch, step):
    """trainval_net_SGG_emb.py:293-300: <save_dir>/<net>/<dataset>/SGG_emb_p_prior_{adaptation}_{dataset}_pre_det_
    session_{s}_epoch_{e}_step_{last step index}_un.pth"""
    return os.path.join(a.save_dir, a.net, a.dataset, "SGG_emb_p_prior_%s_%s_%s_session_%d_epoch_%d_step_%d_un.pth" % (
        a.adaptation, a.dataset, a.vrd_task, session, epoch, step))


def save_checkpoint(a, net, opt, epoch, last_step, rank):
    """The reference's per-epoch dict (:310-317; net_utils.py:119-120): ``epoch`` holds the NEXT epoch, the model sits
    under the reference's state_dict keys.  A column-parallel fc6 is reassembled first (every rank takes part in the
    gather, rank 0 writes), so the file loads on any number of GPUs and under the reference's layer shapes."""
    opt.flush_pending()               # (a no-op since round 6: no update is deferred any more)
    w6, b6 = net.vrd.gather_fc6()
    if rank != 0:
        return None
    from i2vsgg_amd.model.utils.config import cfg
    model = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    model["vrd.fc6.fc.weight"], model["vrd.fc6.fc.bias"] = w6.detach().cpu(), b6.detach().cpu()
    path = checkpoint_name(a, a.session, epoch, last_step)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"session": a.session, "epoch": epoch + 1, "model": model, "optimizer": opt.state_dict(),
                "pooling_mode": cfg.POOLING_MODE, "class_agnostic": False}, path)
    return path


def load_checkpoint(path, net, opt):
    """-> the epoch to start at (the file's ``epoch`` entry, as trainval_net_instance_styleD_bilinear.py:190-191 reads it)."""
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
                st = sd["state"][g["params"][0]]
                for key in ("momentum_buffer", "exp_avg", "exp_avg_sq"):       # SGD / Adam state of the rows this rank keeps
                    if st.get(key) is not None:
                        n = st[key].shape[0] // world
                        st[key] = st[key][rk * n:(rk + 1) * n]
    opt.load_state_dict(sd)
    opt.bump()
    return int(ck["epoch"])


def init_from_detector(path, net, log=print):
    """trainval_net_SGG_emb.py:155-173, the stage-1 -> stage-2 hand-off of the method: every key of the model that does not
    contain 'vrd' is taken from the detector checkpoint's ``model`` dict (a key the file lacks is printed and keeps its
    initialisation, :160-163), ``vrd.*`` is left alone, ``cfg.POOLING_MODE`` follows the file (:171-172).  Optimizer state,
    session and epoch of the file are NOT used (":154 resume only for load faster rcnn model, not for other model status").
    Keys the file holds beyond the model's (netD_pixel.*, netD_style.* of an instance_styleD detector) are ignored, as the
    reference's loop over ``state_dict.keys()`` ignores them.  -> (loaded keys, missing keys)"""
    from i2vsgg_amd.model.faster_rcnn.layers import load_reference_state
    from i2vsgg_amd.model.utils.config import cfg
    ck = torch.load(path, map_location="cpu")
    model = ck["model"]
    own = net.state_dict()
    loaded, missing = {}, []
    for k in own:
        if "vrd" in k:
            continue
        if k not in model:
            log(k)                                            # :161 print(k)
            missing.append(k)
            continue
        if tuple(model[k].shape) != tuple(own[k].shape):
            raise ValueError("init_from_detector: %s is %s in %s, %s here (other --net / --num_classes?)" % (
                k, tuple(model[k].shape), path, tuple(own[k].shape)))
        loaded[k] = model[k]
    load_reference_state(net, loaded, strict=False)
    if "pooling_mode" in ck:
        cfg.POOLING_MODE = ck["pooling_mode"]
    return sorted(loaded), missing


def main(argv=None):
    a = parse_args(argv)
    from i2vsgg_amd import parallel, train
    from i2vsgg_amd.model.utils import config as c
    from i2vsgg_amd.model.utils.net_utils import sampler
    from i2vsgg_amd.roi_data_layer.roibatchLoader import collate_device_prep, roibatchLoader
    from i2vsgg_amd.roi_data_layer.roidb import combined_roidb
    rank, world, dev = parallel.init_from_env()
    c.cfg_from_file(c.default_cfg_file(a.net))
    c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30"])
    if a.scale:
        c.cfg_from_list(["TRAIN.SCALES", "(%d,)" % a.scale])
    if a.set_cfgs:
        c.cfg_from_list(a.set_cfgs)
    np.random.seed(c.cfg.RNG_SEED)
    torch.manual_seed(c.cfg.RNG_SEED)

    # ---- the data path of the reference loop (:73-91)
    c.cfg.TRAIN.USE_FLIPPED = False
    imdb, roidb, ratio_list, ratio_index = combined_roidb(a.imdb_name)
    train_size = len(roidb)
    if rank == 0:
        print("%d source roidb entries" % train_size)
    sampler_batch = sampler(train_size, a.batch_size, rank=rank, world=world, seed=c.cfg.RNG_SEED)
    dataset_s = roibatchLoader(roidb, ratio_list, ratio_index, a.batch_size, imdb.num_classes, training=True, path_return=True,
                               device_prep=a.device_prep)
    dataloader_s = torch.utils.data.DataLoader(dataset_s, batch_size=a.batch_size, sampler=sampler_batch,
                                               num_workers=a.num_workers, pin_memory=dev.type == "cuda",
                                               collate_fn=collate_device_prep if a.device_prep else None)
    iters_per_epoch = a.iters_per_epoch or (train_size // a.batch_size // world)

    net = train.build_sgg_net(101 if a.net == "res101" else 50, a.num_relations, a.num_classes, device=dev, emb_dim=a.emb_dim,
                              use_obj_visual=a.use_obj_visual, spatial_type=a.spatial_type)
    if a.source_gt_rels_path:
        import pickle
        with open(a.source_gt_rels_path, "rb") as f:
            net.vrd.source_gt_rels = pickle.load(f, encoding="bytes")
    elif hasattr(imdb, "gt_rels"):
        net.vrd.source_gt_rels = imdb.gt_rels(a.num_relations)
    else:
        raise SystemExit("no relation annotations: pass --source_gt_rels_path")
    if a.emb_dim != 300 and a.use_obj_visual:
        # resnet_SGG_emb.py:86,97: so_vis_embeddings is FC(4096, emb_dim) but fc_so is FC(300*2, 256) -- the reference's own
        # head only composes at emb_dim 300 while the visual embeddings are on (kept: checkpoints carry that shape)
        raise SystemExit("--emb_dim %d needs --use_obj_visual 0: fc_so is FC(300*2, 256) in the reference (resnet_SGG_emb.py:97)" % a.emb_dim)
    if not (a.use_obj_visual == 1 and a.spatial_type == 2):
        return train_variant(a, net, dataloader_s, iters_per_epoch, dev, rank, world)
    step = train.SGGEmbStep(net, a.batch_size, vrd_lr=a.vrd_lr, device=dev, use_graph=not a.no_graph and dev.type == "cuda",
                            stage_synthetic=False, optimizer=a.optimizer)
    start_epoch = a.start_epoch
    if a.resume and a.resume_train:
        raise SystemExit("--r initialises from a detector checkpoint, --resume_train continues a run of this script: pick one")
    if a.resume:                                             # the reference's --r: detector weights in, vrd.* untouched
        if not a.load_name:
            raise SystemExit("--r needs --load_name <detector checkpoint> (trainval_net_SGG_emb.py:156)")
        if rank == 0:
            print("loading checkpoint %s" % a.load_name)
        loaded, missing = init_from_detector(a.load_name, net, log=print if rank == 0 else (lambda *_: None))
        step.opt.bump()                                      # filters changed under every cache derived from them
        if rank == 0:
            print("loaded checkpoint %s (%d detector tensors, %d not in the file)" % (a.load_name, len(loaded), len(missing)))
    if a.resume_train:
        path = a.load_name or checkpoint_name(a, a.checksession, a.checkepoch, iters_per_epoch - 1)
        start_epoch = load_checkpoint(path, net, step.opt)
        if rank == 0:
            print("loaded checkpoint %s (start epoch %d)" % (path, start_epoch))
    vrd_lr = step.opt.lr_of("vrd.fc7.fc.weight")        # the rate the optimizer holds (a resumed one carries its decays)
    if start_epoch > a.max_epochs:
        return

    data_iter = [iter(dataloader_s)]

    def stage_next():
        """The next minibatch the reference loop would train on (:204-217: batches it skips are skipped here)."""
        for _ in range(4 * len(dataloader_s) + 4):
            try:
                data = next(data_iter[0])
            except StopIteration:
                data_iter[0] = iter(dataloader_s)
                data = next(data_iter[0])
            if (step.stage_batch_u8(data) if a.device_prep else step.stage_batch(data)):
                return
        raise SystemExit("the data loader yields no trainable minibatch")

    total = (a.max_epochs - start_epoch + 1) * iters_per_epoch
    stage_next()
    # the step the loop runs IS the benchmarked one: one captured graph per step (and frame size), fused wgrad+SGD, head of
    # batch k beside the backbone of batch k+1.  The warm-up steps of the capture leave no trace in parameters, momentum or
    # RNG state
    graphed = step.capture(warmup=2, restore=True)
    if rank == 0:
        print("step: %s" % ("HIP graph, overlapped" if graphed and step.overlap else "HIP graph" if graphed
                            else "eager (%s)" % (step.graph_error or "--no-graph")))
    pos, staged = 0, 1                  # minibatches trained / handed to the step so far (the first one before the capture)
    for epoch in range(start_epoch, a.max_epochs + 1):
        if epoch > 1 and (epoch - 1) % a.lr_decay_step == 0:
            step.opt.scale_lr(a.lr_decay_gamma)              # adjust_learning_rate (net_utils.py:113-116), :196-199
            vrd_lr = step.opt.lr_of("vrd.fc7.fc.weight")
            if graphed:
                graphed = step.capture(warmup=0)             # rates live in the captured kernel arguments
                if not graphed and rank == 0:
                    print("re-capture failed, eager launches from here: %s" % step.graph_error)
        lag = step.lag                      # overlapped: call k trains batch k while the backbone runs batch k + lag (1 or 2)
        t0, acc = time.time(), torch.zeros((), device=dev)
        for it in range(iters_per_epoch):
            # overlapped: the batch staged now is the backbone's in this call and the head's ``lag`` calls later (none is staged
            # for the last ``lag`` calls: their backbone branches have nothing new to do).  Sequential: the batch of this call
            while True:
                if staged < min(total, pos + lag + 1):
                    stage_next()                             # queued behind the running step, no host synchronisation
                    staged += 1
                bubble = step.bubble                         # backbone cut by stage: the second call of a run fills the pipeline
                loss_k = step()                              # (both backbone halves, no head: its return value is stale)
                if not bubble:
                    break
            acc += loss_k                                    # loss of batch ``pos``
            pos += 1
            if (it + 1) % a.disp_interval == 0:
                loss = float(acc) / a.disp_interval          # the only host synchronisation of the loop
                acc.zero_()
                if rank == 0:
                    dt = time.time() - t0
                    print("[session %d][epoch %2d][iter %4d/%4d] loss: %.4f, lr: %.2e, vrd_lr: %.2e, %.1f frames/s" % (
                        a.session, epoch, it + 1, iters_per_epoch, loss, a.lr, vrd_lr,
                        world * a.batch_size * a.disp_interval / dt))
                t0 = time.time()
        if not a.no_save:
            path = save_checkpoint(a, net, step.opt, epoch, iters_per_epoch - 1, rank)
            if rank == 0:
                print("save model: %s" % path)
    step.opt.unfuse()
    if world > 1:
        torch.distributed.destroy_process_group()


def train_variant(a, net, loader, iters_per_epoch, dev, rank, world):
    """The non-default variants of the relation head (--use_obj_visual 0, --spatial_type 0 / 1): the reference's loop as written
    (trainval_net_SGG_emb.py:204-255) -- forward of the model on the minibatch, backward, optimizer step -- on eager launches of
    the same kernels (the captured step packs the default head's inputs only)."""
    from i2vsgg_amd import parallel, train
    if a.device_prep or a.resume or a.resume_train:
        raise SystemExit("--use_obj_visual 0 / --spatial_type 0|1 run the plain eager loop: no --device_prep, --r, --resume_train")
    opt = train.make_optimizer(a.optimizer, [(n, p) for n, p in net.named_parameters() if n.startswith("vrd.")], a.vrd_lr)
    it_data = iter(loader)
    for epoch in range(a.start_epoch, a.max_epochs + 1):
        if epoch > 1 and (epoch - 1) % a.lr_decay_step == 0:
            opt.scale_lr(a.lr_decay_gamma)
        t0, acc, n_acc = time.time(), 0.0, 0
        for it in range(iters_per_epoch):
            try:
                data = next(it_data)
            except StopIteration:
                it_data = iter(loader)
                data = next(it_data)
            loss = None
            if isinstance(data, (list, tuple)) and len(data) >= 5:          # the loop skips items that are not lists (:206-217)
                # annotations are looked up by the last path component (trainval_net_SGG_emb.py:217)
                im, info, gt, nb = data[0].to(dev), data[1].to(dev), data[2].to(dev), data[3].to(dev)
                paths = [str(q).split("/")[-1] for q in data[4]]
                loss = net(im, info, gt, nb, paths)
                if not torch.is_tensor(loss):                               # no annotated relation in the minibatch (:177-183)
                    loss = None
            # The reference's `continue` is a single-process one.  With several ranks the exchange is a collective: a rank
            # without a loss still brings (zero) gradients, and the step is skipped only when NO rank has one -- every rank
            # learns that from the same all-reduced count, so all of them take the same branch (round-5 advice)
            if not parallel.any_rank(loss is not None, dev):
                continue
            opt.zero_grad()
            if loss is not None:
                (loss / world).backward()
            parallel.all_reduce_grads(opt.params())
            opt.step()
            if loss is None:
                continue
            acc, n_acc = acc + float(loss.detach()), n_acc + 1
            if (it + 1) % a.disp_interval == 0 and rank == 0 and n_acc:
                print("[session %d][epoch %2d][iter %4d/%4d] loss: %.4f, vrd_lr: %.2e, %.1f frames/s (eager, variant head)" % (
                    a.session, epoch, it + 1, iters_per_epoch, acc / n_acc, opt.lr_of("vrd.fc7.fc.weight"),
                    world * a.batch_size * a.disp_interval / (time.time() - t0)))
                t0, acc, n_acc = time.time(), 0.0, 0
        if not a.no_save:
            path = save_checkpoint(a, net, opt, epoch, iters_per_epoch - 1, rank)
            if rank == 0:
                print("save model: %s" % path)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
