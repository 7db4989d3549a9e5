"""Training steps of the two reference loops, as reusable objects.

``SGGEmbStep``      one step of trainval_net_SGG_emb.py:189-255 (pre_det): backbone forward (no grad,
                    the reference detaches it), relation head forward + backward, SGD(momentum) update
                    of the ``vrd.*`` parameters with the reference's param groups (:129-150).
``InstanceStyleDStep`` one D+G adversarial step of trainval_net_instance_styleD_bilinear.py:262-341.

Both keep their inputs resident on the device, can be captured into a HIP graph (the step is launch
bound otherwise: ~300 small kernels), and all-reduce gradients over RCCL when world_size > 1.
"""
import numpy as np
import torch

from . import ops, parallel
from . import synthetic as syn
from .model.utils.config import cfg


class FusedSGD:
    """SGD(momentum) with the reference's param groups (bias: lr x2 and no weight decay when
    cfg.TRAIN.DOUBLE_BIAS / not BIAS_DECAY) on the fused HIP kernel; one launch per tensor."""

    def __init__(self, named_params, lr, momentum=None, weight_decay=None):
        T = cfg.TRAIN
        self.momentum = T.MOMENTUM if momentum is None else momentum
        wd = T.WEIGHT_DECAY if weight_decay is None else weight_decay
        self.items = []
        self._fused_keys = []
        for name, p in named_params:
            if not p.requires_grad:
                continue
            is_bias = "bias" in name
            self.items.append(dict(
                name=name, p=p, m=torch.zeros_like(p),
                lr=lr * ((T.DOUBLE_BIAS + 1) if is_bias else 1),
                wd=(wd if T.BIAS_DECAY else 0.0) if is_bias else wd))

    def fuse_wgrad(self, min_numel=1 << 24):
        """Fuse the update of large filters into their wgrad epilogue (single-GPU only: with data
        parallelism the gradient must be all-reduced before the update).  Returns the fused names."""
        names = []
        for it in self.items:
            p = it["p"]
            if parallel.exchange_enabled() and not parallel.is_local(p):
                continue              # its gradient has to cross the ranks first
            if p.dim() >= 2 and p.numel() >= min_numel:
                ops.FUSED_SGD[p.data_ptr()] = (it["m"], it["lr"], self.momentum, it["wd"])
                self._fused_keys.append(p.data_ptr())
                names.append(it["name"])
        return names

    def unfuse(self):
        for k in self._fused_keys:
            ops.FUSED_SGD.pop(k, None)
        self._fused_keys = []

    def __del__(self):
        try:
            self.unfuse()
        except Exception:
            pass

    def params(self):
        return [it["p"] for it in self.items]

    def zero_grad(self):
        for it in self.items:
            it["p"].grad = None

    def scale_lr(self, k):
        for it in self.items:
            it["lr"] *= k

    MULTI_BELOW = 1 << 20       # tensors under 1 Mi elements share one launch

    @torch.no_grad()
    def step(self):
        small = []
        for it in self.items:
            p, g = it["p"], it["p"].grad
            if g is None:
                continue
            if g.stride() != p.stride():
                g = torch.empty_like(p).copy_(g)
            if p.numel() < self.MULTI_BELOW:
                small.append((p, g, it))
            else:
                ops.sgd_momentum_(p, g, it["m"], it["lr"], self.momentum, it["wd"])
        if small:
            ops.sgd_momentum_multi_([p for p, _, _ in small], [g for _, g, _ in small], [it["m"] for _, _, it in small],
                                    [it["lr"] for _, _, it in small], [it["wd"] for _, _, it in small], self.momentum)


def synthetic_sgg_batch(seed, n_frames, n_boxes=32, n_pairs=32, n_rel=62, n_cls=16, h=600, w=1000):
    """frames + per-frame annotation dicts (keys ``f0..``) + im_info."""
    im, info = syn.frames(seed, n_frames, h, w)
    annos = {"f%d" % i: syn.relation_annotation(seed * 1000 + i, n_boxes, n_pairs, n_rel, n_cls, h, w)
             for i in range(n_frames)}
    return im, info, annos


class SGGEmbStep:
    def __init__(self, net, n_frames, vrd_lr=1e-4, seed=1, device="cuda:0", h=600, w=1000, n_boxes=32, n_pairs=32,
                 use_graph=True, fuse_sgd=True, zero_arena=True):
        self.net, self.dev, self.n_frames = net, torch.device(device), n_frames
        self.world = parallel.world_size()
        self.geom = (h, w, n_boxes, n_pairs)
        self.reseed(seed)
        # data parallelism for everything except vrd.fc6, which is cut by output columns (parallel.py): its 822 MB
        # gradient -- 91 % of the exchange -- stays local and its SGD update stays fused into the wgrad epilogue
        import os
        self.tp = parallel.exchange_enabled() and os.environ.get("I2V_TP_FC6", "1") != "0" and \
            net.vrd.fc6.fc.weight.shape[0] % max(self.world, 1) == 0
        if self.tp and net.vrd.tp is None:
            net.vrd.enable_fc6_tp(parallel.rank(), self.world)
        self.opt = FusedSGD([(n, p) for n, p in net.named_parameters() if n.startswith("vrd.")], vrd_lr)
        self.fused = self.opt.fuse_wgrad() if fuse_sgd else []
        self.loss = torch.zeros((), device=self.dev)
        self.graph = None
        self.use_graph = use_graph
        self.arena = ops.ZeroArena(1024, self.dev) if zero_arena else None    # sized after the first step
        self.arena_bb = ops.ZeroArena(1024, self.dev) if zero_arena else None
        self.fmap = None
        self.fmap_head = None
        self.pipelined = False
        import os as _os
        self.overlap = _os.environ.get("I2V_OVERLAP", "1") != "0" and use_graph

    def reseed(self, seed):
        """(Re)generate the synthetic minibatch: frames, pair tables, masks, labels -> static device inputs (the
        data layer's job; resident before the timed region).  Shapes do not depend on the seed."""
        from .model.faster_rcnn.faster_rcnn_SGG_emb import build_pair_tables, rasterize_masks
        h, w, n_boxes, n_pairs = self.geom
        head = self.net.vrd
        im, info, annos = synthetic_sgg_batch(seed, self.n_frames, n_boxes, n_pairs, head.n_rel, head.n_obj, h, w)
        head.source_gt_rels = annos
        self.paths = sorted(annos, key=lambda s: int(s[1:]))
        boxes, relb, bounds, labels, ixs, ixo, counts, off = [], [], [], [], [], [], [], 0
        for f, path in enumerate(self.paths):
            gt, union, bnd, lab, s, o = build_pair_tables(annos[path], float(info[f][2]), float(info[f][0]),
                                                          float(info[f][1]), head.n_rel)
            b5 = np.zeros((gt.shape[0], 5), np.float32); b5[:, 0] = f; b5[:, 1:] = gt
            r5 = np.zeros((union.shape[0], 5), np.float32); r5[:, 0] = f; r5[:, 1:] = union
            boxes.append(b5); relb.append(r5); bounds.append(bnd); labels.append(lab)
            ixs.append(s + off); ixo.append(o + off); counts.append(lab.shape[0]); off += gt.shape[0]
        t = lambda a, dt=torch.float32: torch.from_numpy(np.concatenate(a)).to(self.dev, dt)
        # frames in the layout the device front-end emits (ops.image_prep): NHWC with the stem's zero fourth channel
        im4 = torch.zeros((im.shape[0], 4) + tuple(im.shape[2:]), device=self.dev).contiguous(memory_format=torch.channels_last)
        im4[:, :3] = torch.from_numpy(im).to(self.dev)
        new = dict(im=im4, info=torch.from_numpy(info).to(self.dev), boxes=t(boxes),
                   relb=t(relb), labels=t(labels), ixs=t(ixs, torch.long), ixo=t(ixo, torch.long),
                   masks=torch.nn.functional.pad(rasterize_masks(np.concatenate(bounds), self.dev), (0, 0, 0, 0, 0, 2)),  # 2 zero channels: the float4 pad of conv_lo.0, done once by the data side
                   wrow=torch.cat([torch.full((c,), 1.0 / (c * len(counts))) for c in counts]).to(self.dev))
        for k, v in new.items():
            cur = getattr(self, k, None)
            if cur is not None and cur.shape == v.shape:
                cur.copy_(v)                 # keep addresses: a captured graph stays valid
            else:
                setattr(self, k, v)
        self.n_rows = int(self.boxes.shape[0] + self.relb.shape[0])

    # The step in two halves.  The backbone is frozen in this loop (the reference detaches it), so its
    # forward does not depend on the head's weights: with data parallelism the backbone pass of the NEXT
    # minibatch runs while the gradients of this one are exchanged (see __call__).
    def _backbone(self):
        ops.ARENA = self.arena_bb
        try:
            if self.arena_bb is not None:
                self.arena_bb.reset()
            with torch.no_grad():
                fmap = self.net.RCNN_base(self.im)
            if self.fmap is None:
                self.fmap = torch.empty_like(fmap)
            self.fmap.copy_(fmap)           # static address for the head's graph; 20 MB, ~8 us
        finally:
            ops.ARENA = None

    def _head(self):
        ops.ARENA = self.arena
        try:
            if self.arena is not None:
                self.arena.reset()          # one clear for every atomically accumulated output of this half
            fmap = self.fmap_head if self.overlap else self.fmap
            score, _ = self.net.vrd.forward_device(fmap, self.boxes, self.relb, self.masks, self.ixs, self.ixo)
            loss = ops.bce_rows(score, self.labels, self.wrow)     # sum_r wrow[r] * mean_c BCE: one kernel each way
            self.opt.zero_grad()
            (loss / self.world).backward()
            self.loss.copy_(loss.detach())
        finally:
            ops.ARENA = None

    def _body(self):
        self._backbone()
        if self.overlap:
            if self.fmap_head is None:
                self.fmap_head = torch.empty_like(self.fmap)
            self.fmap_head.copy_(self.fmap)
        self._head()
        parallel.all_reduce_grads(self.opt.params())
        self.opt.step()

    def _size_arenas(self):
        for name in ("arena", "arena_bb"):
            a = getattr(self, name)
            if a is not None and a.wanted * 4 > a.buf.numel() * 4:
                setattr(self, name, ops.ZeroArena(int(a.wanted * 4 * 1.05) + 4096, self.dev))

    def capture(self, warmup=2):
        """Warm up eagerly on a side stream, then capture the step into HIP graphs.

        world == 1: one graph for the whole step.  world > 1 (or I2V_SPLIT_GRAPH=1): three graphs --
        backbone forward / head forward+backward / SGD update -- with the RCCL all-reduce of the
        (graph-static) gradient tensors launched eagerly between them: the collective stays outside
        capture, the ~330 compute launches do not go through Python."""
        import os
        s = torch.cuda.Stream(self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            for i in range(warmup):
                self._body()
                if i == 0:
                    self._size_arenas()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        if not self.use_graph:
            return False
        self.s_main = torch.cuda.Stream(self.dev)       # every replay runs here (see __call__)
        self.pipelined = parallel.exchange_enabled() or os.environ.get("I2V_SPLIT_GRAPH") == "1"
        try:
            if self.overlap:
                self._capture_overlapped()
                self._tune_for(torch.cuda.current_stream(self.dev))
            elif not self.pipelined:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._body()
                self.graph = (g,)
            else:
                gbb, gh, gs = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                # The backbone graph gets its OWN memory pool: it is replayed between the head graph and
                # the SGD graph, and graphs that share a pool may reuse each other's freed blocks -- its
                # activations would land on the gradients the update is about to read.
                with torch.cuda.graph(gbb):
                    self._backbone()
                with torch.cuda.graph(gh):
                    self._head()
                self._grads = [p.grad for p in self.opt.params()]      # static tensors owned by the head graph's pool
                with torch.cuda.graph(gs, pool=gh.pool()):
                    self.opt.step()
                self.graph = (gbb, gh, gs)
                self.s_main.wait_stream(torch.cuda.current_stream(self.dev))
                with torch.cuda.stream(self.s_main):
                    gbb.replay()            # feature map of the first timed step
            return True
        except Exception as e:      # report, fall back to eager launches
            self.graph = None
            self.pipelined = self.overlap = False
            self.graph_error = repr(e)
            torch.cuda.synchronize(self.dev)
            return False

    def _capture_overlapped(self):
        """Two streams.  The backbone is frozen, so the backbone pass of the NEXT minibatch does not depend on this
        step's update: it runs on its own stream BESIDE this step's head (whose long kernels -- the fused fc6
        wgrad+SGD, ROI pooling -- are HBM- or latency-bound, and whose ~100 small kernels leave most CUs idle) and
        beside the gradient exchange, instead of after them.  The feature map is handed over through a copy at the
        top of each step.  The graphs pin different split-K workspace slabs (they are captured on one stream but
        replayed concurrently).  Single GPU: head graph = head fwd+bwd + SGD.  Multi-GPU: head graph (with the
        column-parallel fc6 collectives captured), eager all-reduce of the remaining gradients, SGD graph."""
        from ._lib import lib
        self.s_bb = torch.cuda.Stream(self.dev)          # capture() -> _tune_side_stream() settles its priority
        self.ev_bb, self.ev_copy = torch.cuda.Event(), torch.cuda.Event()
        gbb, gh = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        gs = torch.cuda.CUDAGraph() if self.pipelined else None
        try:
            lib.i2v_conv_set_split_slot(1)
            with torch.cuda.graph(gbb):
                self._backbone()
            lib.i2v_conv_set_split_slot(0)
            with torch.cuda.graph(gh):
                self._head()
                if not self.pipelined:
                    self.opt.step()
            if self.pipelined:
                self._grads = [p.grad for p in self.opt.params()]      # static tensors owned by the head graph's pool
                with torch.cuda.graph(gs, pool=gh.pool()):
                    self.opt.step()
        finally:
            lib.i2v_conv_set_split_slot(-1)
        self.graph = (gbb, gh, gs) if self.pipelined else (gbb, gh)

    def _host_stream(self, caller):
        """The stream that carries the head graph for this caller: the caller's own, unless that is HIP's legacy
        default stream (see __call__)."""
        return self.s_main if caller == torch.cuda.default_stream(self.dev) else caller

    def _tune_for(self, caller):
        """Settle the side stream for the stream the replays will run on: which priority interleaves depends on the
        PAIR of streams (measured: the same high-priority side stream gives 5.1 ms beside one stream and 10.3 beside
        another), so the measurement runs on the hosting stream itself and is repeated when a later call arrives on a
        different one.  Snapshot, steps and restore all run there -- never on the default stream."""
        host = self._host_stream(caller)
        self._tuned_for = host                      # the tuning steps below go through __call__
        if host != caller:
            torch.cuda.synchronize(self.dev)
            host.wait_stream(caller)
        with torch.cuda.stream(host):
            self._tune_side_stream()
        if host != caller:
            caller.wait_stream(host)
            torch.cuda.synchronize(self.dev)

    def _prime(self, stream):
        """Backbone pass of the first step on ``stream`` (the feature map the first head replay consumes)."""
        self.s_bb = stream
        torch.cuda.synchronize(self.dev)
        with torch.cuda.stream(stream):
            self.graph[0].replay()
            self.ev_bb.record(stream)

    def _tune_side_stream(self):
        """Which stream carries the backbone graph is settled by measurement.  How HIP maps streams to hardware queues
        decides whether the two graphs interleave at all, and it depends on things this object does not control.
        Measured on MI355X / ROCm 7.2 (ms per step): without a process group a normal-priority side stream gives 5.1
        and a high-priority one 10.3; with an RCCL process group alive it is 6.1 (the side stream shares an in-order
        queue: no interleaving) against 5.1.  So: run three steps with each, keep the faster, and put parameters,
        momentum and the RNG state back exactly as they were -- the tuning steps leave no trace in the trajectory.
        I2V_BB_PRIORITY=0/-1 pins the choice."""
        import os
        import time
        pin = os.environ.get("I2V_BB_PRIORITY")
        if pin is not None:
            self.bb_priority = int(pin)
            self._prime(torch.cuda.Stream(self.dev, priority=self.bb_priority))
            return
        state = [it["p"].data for it in self.opt.items] + [it["m"] for it in self.opt.items]
        saved = [t.clone() for t in state]
        rng = torch.cuda.get_rng_state(self.dev)
        timing = {}
        for prio in (0, -1):
            stream = torch.cuda.Stream(self.dev, priority=prio)
            self._prime(stream)
            self()                                    # settle
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            for _ in range(3):
                self()
            torch.cuda.synchronize(self.dev)
            timing[prio] = (time.perf_counter() - t0, stream)
        # normal priority unless the high-priority stream is clearly faster (it is by ~17 % when an RCCL process group is
        # alive): when the two measure alike, the high-priority stream has been seen to fall into the 10 ms mode later
        self.bb_priority = -1 if timing[-1][0] < 0.92 * timing[0][0] else 0
        self.bb_tuning_ms = {k: v[0] / 3 * 1e3 for k, v in timing.items()}
        with torch.no_grad():
            for t, sv in zip(state, saved):
                t.copy_(sv)
        torch.cuda.set_rng_state(rng, self.dev)
        self._prime(timing[self.bb_priority][1])

    def _call_overlapped(self):
        gbb, gh = self.graph[0], self.graph[1]
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(self.ev_bb)          # backbone(i) done
        self.fmap_head.copy_(self.fmap)     # hand-off: 20 MB
        self.ev_copy.record(cur)
        with torch.cuda.stream(self.s_bb):
            self.s_bb.wait_event(self.ev_copy)
            gbb.replay()                    # backbone(i+1), beside ...
            self.ev_bb.record(self.s_bb)
        gh.replay()                         # ... head(i) fwd + bwd (+ SGD on one GPU)
        if self.pipelined:
            parallel.all_reduce_grads(self.opt.params())      # the exchange also runs beside the backbone stream
            self.graph[2].replay()
        return self.loss

    def __call__(self):
        """One step = one backbone pass, one head pass, one exchange (world > 1), one update.  Default: the backbone
        pass is the NEXT minibatch's, on its own stream beside this step's head (``_capture_overlapped``; the frames
        are static here, in a training loop that is where the next batch goes).  I2V_OVERLAP=0: everything on one
        stream -- one graph on one GPU; head -> [exchange || backbone] -> SGD with world > 1."""
        if self.graph is None:
            self._body()
            return self.loss
        # Graph replays never go to HIP's legacy default stream.  Measured (tools/loss_trace.py, bench.py): with the
        # two-stream step replayed back to back on the default stream, about half of the 23-step runs end with a
        # different, an all-zero-logit or a NaN loss; a host-side synchronize of that stream before every step removes
        # it, and so does replaying on an ordinary stream -- there every run reproduces the one-graph / eager
        # trajectory to the last bit or two (tests/test_gpu_models.py::test_sgg_step_back_to_back_replays_are_ordered).
        # A caller that is on the default stream gets the replays on a stream of this object, ordered against its own
        # stream on both sides and with a device synchronize before each step (event waits alone were not enough once
        # the side stream had a priority): correct, but it costs the overlap across steps (6.0 instead of 5.1 ms per
        # step), so bench.py and the training script set an ordinary current stream.
        caller = torch.cuda.current_stream(self.dev)
        host = self._host_stream(caller)
        detour = host != caller
        if self.overlap and getattr(self, "_tuned_for", None) != host:
            self._tune_for(caller)                  # first call from this stream
        if detour:
            torch.cuda.synchronize(self.dev)        # nothing of the previous step in flight: the conservative form
            self.s_main.wait_stream(caller)
        with torch.cuda.stream(host):
            if self.overlap:
                self._call_overlapped()
            elif len(self.graph) == 1:
                self.graph[0].replay()
            else:
                gbb, gh, gs = self.graph
                gh.replay()
                token = parallel.all_reduce_grads_start(self.opt.params())
                gbb.replay()
                parallel.all_reduce_grads_finish(token)
                gs.replay()
        if detour:
            caller.wait_stream(self.s_main)
        return self.loss


class InstanceStyleDStep:
    """One D+G adversarial step (trainval_net_instance_styleD_bilinear.py:262-341): source forward with
    detection + RPN losses and 0.5*mean(d^2) for both discriminators, target forward with
    0.5*mean((1-d)^2), style terms weighted by style_lambda, ONE backward through the gradient-reversal
    layers, one SGD step over every trainable parameter.  Eager (the target layers sample on the host
    with the reference's np.random stream, which needs two small D2H copies per step)."""

    def __init__(self, net, n_frames, lr=5e-4, eta=0.1, eta_style=0.001, style_lambda=1.0, seed=3, device="cuda:0",
                 h=600, w=1000, n_gt=8, cr=False):
        self.net, self.dev = net, torch.device(device)
        self.cr = cr                  # --cr: consistency regularisation between instance- and image-level D (:299-312)
        self.world = parallel.world_size()
        self.eta, self.eta_style, self.style_lambda = eta, eta_style, style_lambda
        ims, info = syn.frames(seed, n_frames, h, w)
        imt, _ = syn.frames(seed + 100, n_frames, h, w)
        gt, nb = syn.gt_boxes(seed, n_frames, n_gt, net.n_classes, cfg.MAX_NUM_GT_BOXES, h, w)
        to = lambda a: torch.from_numpy(a).to(self.dev)
        self.im_s, self.im_t, self.info, self.gt, self.nb = to(ims), to(imt), to(info), to(gt), to(nb)
        self.gt_t = torch.zeros((n_frames, 1, 5), device=self.dev)
        self.nb_t = torch.zeros((n_frames,), device=self.dev)
        self.opt = FusedSGD(list(net.named_parameters()), lr)
        self.losses = {}

    def __call__(self):
        net = self.net
        out = net(self.im_s, self.info, self.gt, self.nb, target=False, eta=self.eta, eta_style=self.eta_style)
        _, _, _, l_rpn_cls, l_rpn_box, l_cls, l_box, _, d_inst, d_style = out
        loss = l_rpn_cls.mean() + l_rpn_box.mean() + l_cls.mean() + l_box.mean()
        dloss_s = 0.5 * torch.mean(d_inst ** 2)
        dloss_s_style = 0.5 * torch.mean(d_style ** 2)
        d_inst_t, d_style_t = net(self.im_t, self.info, self.gt_t, self.nb_t, target=True, eta=self.eta,
                                  eta_style=self.eta_style)
        dloss_t = 0.5 * torch.mean((1 - d_inst_t) ** 2)
        dloss_t_style = 0.5 * torch.mean((1 - d_style_t) ** 2)
        total = loss + dloss_s + dloss_t + self.style_lambda * (dloss_s_style + dloss_t_style)
        cst = {}
        if self.cr:
            cst = consistency_terms(d_inst, d_style, d_inst_t, d_style_t)
            total = total + cst["source_adv_cst"] + cst["target_adv_cst"]
        self.opt.zero_grad()
        (total / self.world).backward()
        parallel.all_reduce_grads(self.opt.params())
        self.opt.step()
        self.losses = dict(total=total.detach(), det=loss.detach(), dloss_s=dloss_s.detach(), dloss_t=dloss_t.detach(),
                           dloss_s_style=dloss_s_style.detach(), dloss_t_style=dloss_t_style.detach(),
                           **{k: v.detach() for k, v in cst.items()})
        return self.losses["total"]


def consistency_terms(d_inst, d_style, d_inst_t, d_style_t):
    """trainval_net_instance_styleD_bilinear.py:299-311 (``--cr``): MSE between the per-ROI mean of the instance
    discriminator map and the (detached) image-level discriminator output repeated once per ROI.  The reference
    hard-codes 128 ROIs per image (``repeat(1,128)``); here the repeat count is the actual
    number of ROIs per frame, which is the same thing at TRAIN.BATCH_SIZE = 128."""
    out = {}
    for name, di, ds in (("source_adv_cst", d_inst, d_style), ("target_adv_cst", d_inst_t, d_style_t)):
        per_roi = di.mean(3).mean(2)                                  # (B*R, 1)
        rois_per_frame = per_roi.shape[0] // ds.shape[0]
        prob = ds.reshape(ds.shape[0], -1)[:, :1].repeat(1, rois_per_frame).view(-1, 1)
        out[name] = torch.nn.functional.mse_loss(per_roi, prob.detach())
    return out


def build_instance_styled_net(layers=101, n_cls=16, seed=0, device="cuda:0"):
    from .model.faster_rcnn.resnet_instance_styleD_bilinear import resnet
    torch.manual_seed(seed)
    net = resnet(tuple(range(n_cls)), layers)
    net.create_architecture()
    _randomise_bn(net, seed + 1)
    return net.to(device).train()


def _randomise_bn(net, seed):
    """Frozen-BN statistics as a trained checkpoint would have them (gamma < 1 on the block outputs keeps
    activations O(1) through 33 residual blocks)."""
    g = torch.Generator().manual_seed(seed)
    for name, m in net.named_modules():
        if m.__class__.__name__ == "FrozenBN":
            c = m.weight.numel()
            hi = 0.5 if name.endswith("bn3") else 1.0
            m.weight.data.copy_(torch.rand(c, generator=g) * (hi - 0.2) + 0.2)
            m.bias.data.copy_(torch.rand(c, generator=g) * 0.2 - 0.1)
            m.running_mean.copy_(torch.rand(c, generator=g) * 0.2 - 0.1)
            m.running_var.copy_(torch.rand(c, generator=g) + 0.5)
            m.invalidate()


def build_sgg_net(layers=101, n_rel=62, n_cls=16, seed=0, device="cuda:0"):
    """Random-init SGG_emb model of the reference architecture (no checkpoint is reachable)."""
    import argparse
    from .model.faster_rcnn.resnet_SGG_emb import resnet
    torch.manual_seed(seed)
    args = argparse.Namespace(num_relations=n_rel, num_classes=n_cls, emb_dim=300, use_obj_visual=True,
                              spatial_type=2, vrd_task="pre_det")
    net = resnet(tuple(range(n_cls)), args, layers, obj_vecs=syn.word_vectors(22, n_cls),
                 prd_vecs=syn.word_vectors(21, n_rel))
    net.create_architecture()
    _randomise_bn(net, seed + 1)
    return net.to(device).train()


# FLOPs of the implicit-GEMM launches of one SGGEmbStep (algorithmic: 2*M*N*K per launch) -------------
def conv_flops_backbone(n_frames, h=600, w=1000, blocks=(3, 4, 23)):
    """2*MAC of conv1..layer3 for n_frames frames (166.1 GFLOP/frame at 600x1000).
    Returns (flops, launches)."""
    def out(n, k, s, p):
        return (n + 2 * p - k) // s + 1
    fl, n = 0, 0
    H, W = out(h, 7, 2, 3), out(w, 7, 2, 3)
    fl += 2 * H * W * 64 * 49 * 3; n += 1
    H, W = -(-(H - 3) // 2) + 1, -(-(W - 3) // 2) + 1
    cin = 64
    for planes, nb, stride in ((64, blocks[0], 1), (128, blocks[1], 2), (256, blocks[2], 2)):
        for i in range(nb):
            s = stride if i == 0 else 1
            Ho, Wo = out(H, 1, s, 0), out(W, 1, s, 0)
            fl += 2 * Ho * Wo * planes * cin                       # conv1 1x1 (strided)
            fl += 2 * Ho * Wo * planes * planes * 9                # conv2 3x3
            fl += 2 * Ho * Wo * planes * 4 * planes                # conv3 1x1
            n += 3
            if i == 0:
                fl += 2 * Ho * Wo * planes * 4 * cin               # downsample
                n += 1
            H, W, cin = Ho, Wo, planes * 4
    return fl * n_frames, n
