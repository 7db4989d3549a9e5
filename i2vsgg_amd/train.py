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
            p._i2v_trained = True        # updated through raw device pointers: caches keyed on p._version also key on ops.PARAM_EPOCH
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

    @staticmethod
    def bump():
        """The parameters changed (an eager ``step()``, a fused wgrad+SGD epilogue or a graph replay that contains
        them): whatever is derived from trained parameters and cached (Winograd-domain filters) is stale."""
        ops.PARAM_EPOCH += 1

    def state_dict(self):
        """torch.optim.SGD's layout (param_groups + state[i]['momentum_buffer']) in named_parameters order, so that a
        checkpoint written here resumes under torch.optim.SGD and vice versa."""
        return {"state": {i: {"momentum_buffer": it["m"].detach().clone()} for i, it in enumerate(self.items)},
                "param_groups": [{"lr": it["lr"], "momentum": self.momentum, "weight_decay": it["wd"], "params": [i],
                                  "name": it["name"]} for i, it in enumerate(self.items)]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        flat = [pi for g in groups for pi in g["params"]]
        if len(flat) != len(self.items):
            raise ValueError("optimizer state holds %d parameters, this optimizer %d" % (len(flat), len(self.items)))
        by_param = {pi: g for g in groups for pi in g["params"]}
        for i, it in enumerate(self.items):
            g = by_param[flat[i]]
            it["lr"], it["wd"] = float(g["lr"]), float(g.get("weight_decay", it["wd"]))
            self.momentum = float(g.get("momentum", self.momentum))
            st = sd["state"].get(flat[i], sd["state"].get(str(flat[i])))
            if st is not None and st.get("momentum_buffer") is not None:
                it["m"].copy_(st["momentum_buffer"].reshape(it["m"].shape))
            else:
                it["m"].zero_()
        for k in list(self._fused_keys):       # fused entries hold (momentum, lr, ...) by value
            for it in self.items:
                if it["p"].data_ptr() == k:
                    ops.FUSED_SGD[k] = (it["m"], it["lr"], self.momentum, it["wd"])

    def zero_grad(self):
        for it in self.items:
            it["p"].grad = None

    def scale_lr(self, k):
        for it in self.items:
            it["lr"] *= k
            ent = ops.FUSED_SGD.get(it["p"].data_ptr())
            if ent is not None:                 # fused entries hold the rate by value (a captured graph holds it too:
                ops.FUSED_SGD[it["p"].data_ptr()] = (ent[0], it["lr"], ent[2], ent[3])   # re-capture after a decay)

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
        self.bump()


def synthetic_sgg_batch(seed, n_frames, n_boxes=32, n_pairs=32, n_rel=62, n_cls=16, h=600, w=1000):
    """frames + per-frame annotation dicts (keys ``f0..``) + im_info."""
    im, info = syn.frames(seed, n_frames, h, w)
    annos = {"f%d" % i: syn.relation_annotation(seed * 1000 + i, n_boxes, n_pairs, n_rel, n_cls, h, w)
             for i in range(n_frames)}
    return im, info, annos


class _Slot:
    """One minibatch worth of head inputs packed into ONE device buffer (256-B aligned fields), so that moving a
    batch between pipeline stages is a single copy whatever the number of fields."""

    def __init__(self, fields, device):
        self.spec, off = [], 0
        for name, t in fields.items():
            nbytes = t.numel() * t.element_size()
            self.spec.append((name, off, nbytes, t.dtype, tuple(t.shape)))
            off += (nbytes + 255) // 256 * 256
        self.buf = torch.zeros(max(off, 256), dtype=torch.uint8, device=device)
        self.views = {name: self.buf[o:o + n].view(dt).view(shape) for name, o, n, dt, shape in self.spec}

    def same_layout(self, fields):
        return [(n, dt, sh) for n, _, _, dt, sh in self.spec] == [(n, t.dtype, tuple(t.shape)) for n, t in fields.items()]

    def write(self, fields):
        for name, t in fields.items():
            self.views[name].copy_(t)


class SGGEmbStep:
    """One step of trainval_net_SGG_emb.py:189-255 (pre_det) as a replayable object.

    A step = one backbone pass (no grad: the reference detaches the feature map, faster_rcnn_SGG_emb.py:148), one
    relation-head forward + backward, one gradient exchange (world > 1), one SGD(momentum) update of ``vrd.*``.

    Schedule (``overlap``, the default with HIP graphs): the backbone is frozen in this loop, so the backbone pass of
    the NEXT minibatch does not depend on this step's update.  The whole step is ONE captured graph with two branches
    between a fork and a join: [head fwd + bwd (+ exchange) + SGD of batch k] beside [backbone of batch k+1].  The
    branches own disjoint device state (``ops.LaunchContext``: zero arena, split-K workspace, scratch) and meet only at
    graph edges: the feature-map hand-off (one 20 MB copy before the fork) and the join.  There is one graph launch
    per step on the caller's stream, no side stream, no event and no priority for a caller to get wrong.

    Minibatches move through a three-slot pipeline so that ``stage()`` may be called at any time between steps:
    ``stage(b)`` writes the frames of b (read by the NEXT call's backbone branch) and its head inputs into the ``in``
    slot; each call starts with cur <- nxt, nxt <- in (two small copies inside the graph) and then runs head(cur) beside
    backbone(frames).  A batch staged before call k is therefore consumed by the backbone in call k and by the head in
    call k+1 -- features and boxes / labels of one batch always meet.  ``overlap=False`` (and eager mode): backbone and
    head of the staged batch in the same call."""

    def __init__(self, net, n_frames, vrd_lr=1e-4, seed=1, device="cuda:0", h=600, w=1000, n_boxes=32, n_pairs=32,
                 use_graph=True, fuse_sgd=True, zero_arena=True, overlap=None):
        import os
        self.net, self.dev, self.n_frames = net, torch.device(device), n_frames
        self.world = parallel.world_size()
        self.geom = (h, w, n_boxes, n_pairs)
        # data parallelism for everything except vrd.fc6, which is cut by output columns (parallel.py): its 822 MB
        # gradient -- 91 % of the exchange -- stays local and its SGD update stays fused into the wgrad epilogue
        self.tp = parallel.exchange_enabled() and os.environ.get("I2V_TP_FC6", "1") != "0" and \
            net.vrd.fc6.fc.weight.shape[0] % max(self.world, 1) == 0
        if self.tp and net.vrd.tp is None:
            net.vrd.enable_fc6_tp(parallel.rank(), self.world)
        self.opt = FusedSGD([(n, p) for n, p in net.named_parameters() if n.startswith("vrd.")], vrd_lr)
        self.fused = self.opt.fuse_wgrad() if fuse_sgd else []
        self.loss = torch.zeros((), device=self.dev)
        self.graph = None
        self.graph_error = None
        self.use_graph = use_graph
        if overlap is None:
            overlap = os.environ.get("I2V_OVERLAP", "1") != "0"
        self.overlap = bool(overlap) and use_graph
        self.ctx_bb = ops.LaunchContext(self.dev, arena=zero_arena)        # backbone branch
        self.bb_split = os.environ.get("I2V_BB_SPLIT", "1") == "1" and use_graph and n_frames > 1
        self.ctx_frames = [ops.LaunchContext(self.dev, arena=zero_arena) for _ in range(n_frames)] if self.bb_split else []
        self._frame_streams = [torch.cuda.Stream(self.dev) for _ in range(n_frames)] if self.bb_split else []
        self.ctx_head = ops.LaunchContext(self.dev, arena=zero_arena)      # head branch
        self.im = self.fmap = self.fmap_head = None
        self.cur = self.nxt = self.inp = None
        self.primed = False
        self.reseed(seed)

    # ------------------------------------------------------------------ data side
    def _synthetic(self, seed):
        """SURVEY.md 8d config 2 as device tensors: frames in the layout the device front-end emits (ops.image_prep:
        NHWC with the stem's zero fourth channel) + the head inputs of faster_rcnn_SGG_emb.py:170-245."""
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
        im4 = torch.zeros((im.shape[0], 4) + tuple(im.shape[2:]), device=self.dev).contiguous(memory_format=torch.channels_last)
        im4[:, :3] = torch.from_numpy(im).to(self.dev)
        fields = dict(boxes=t(boxes), relb=t(relb), labels=t(labels), ixs=t(ixs, torch.long), ixo=t(ixo, torch.long),
                      # 2 zero channels: the float4 pad of conv_lo.0, done once by the data side
                      masks=torch.nn.functional.pad(rasterize_masks(np.concatenate(bounds), self.dev), (0, 0, 0, 0, 0, 2)),
                      wrow=torch.cat([torch.full((c,), 1.0 / (c * len(counts))) for c in counts]).to(self.dev))
        return im4, torch.from_numpy(info).to(self.dev), fields

    def stage(self, im4, info, fields):
        """Hand the NEXT minibatch to the step: frames (N,4,H,W) channels_last + head inputs.  Ordered on the caller's
        stream like everything else: it may be called right after ``__call__`` returns, the copies queue behind the
        step that is still running.  Shapes must match the first staged batch once a graph has been captured."""
        if self.inp is None or (self.graph is None and not self.inp.same_layout(fields)):
            self.cur, self.nxt, self.inp = (_Slot(fields, self.dev) for _ in range(3))
            for k, v in self.cur.views.items():
                setattr(self, k, v)                      # the head reads the ``cur`` slot
            self.im, self.info = im4.clone(memory_format=torch.preserve_format), info.clone()
            self.n_rows = int(fields["boxes"].shape[0] + fields["relb"].shape[0])
            for slot in (self.cur, self.nxt, self.inp):
                slot.write(fields)
            self.primed = False
            return
        if not self.inp.same_layout(fields) or im4.shape != self.im.shape:
            raise ValueError("SGGEmbStep.stage: the captured graph is bound to the shapes of the first staged batch")
        if self.tp and self.graph is None:
            parallel.assert_same_rows(fields["boxes"].shape[0] + fields["relb"].shape[0], "boxes + pairs")
        self.im.copy_(im4)
        self.info.copy_(info)
        self.inp.write(fields)

    def reseed(self, seed):
        """Stage the synthetic minibatch of ``seed`` (the data layer's job; resident before the timed region)."""
        self.stage(*self._synthetic(seed))

    # ------------------------------------------------------------------ the two halves of a step
    def _rotate(self):
        if self.overlap:
            self.cur.buf.copy_(self.nxt.buf)
            self.nxt.buf.copy_(self.inp.buf)
        else:
            self.cur.buf.copy_(self.inp.buf)

    def _backbone(self):
        with self.ctx_bb:
            with torch.no_grad():
                fmap = self.net.RCNN_base(self.im)
            if self.fmap is None:
                self.fmap = torch.empty_like(fmap)
            self.fmap.copy_(fmap)           # static address across replays; 20 MB, ~8 us

    def _backbone_per_frame(self, join=True):
        """Captured form with ``bb_split``: the frames of the minibatch are independent chains of ~100 short kernels each
        (a layer3 GEMM of one frame runs ~15 us, a quarter of it set-up, first-load latency and the store tail with the
        matrix pipe idle).  One graph branch per frame: the kernels of the two chains are co-resident on every CU, the
        fixed phases of one lie under the K loops of the other."""
        main = torch.cuda.current_stream(self.dev)
        n = self.im.shape[0]
        for f in range(n):
            st = self._frame_streams[f]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                with self.ctx_frames[f]:
                    with torch.no_grad():
                        fm = self.net.RCNN_base(self.im[f:f + 1])
                    self.fmap[f:f + 1].copy_(fm)
        if join:
            for f in range(n):
                main.wait_stream(self._frame_streams[f])

    def _head(self):
        with self.ctx_head:
            fmap = self.fmap_head if self.overlap else self.fmap
            score, _ = self.net.vrd.forward_device(fmap, self.boxes, self.relb, self.masks, self.ixs, self.ixo)
            loss = ops.bce_rows(score, self.labels, self.wrow)     # sum_r wrow[r] * mean_c BCE: one kernel each way
            self.opt.zero_grad()
            (loss / self.world).backward()
            self.loss.copy_(loss.detach())
            parallel.all_reduce_grads(self.opt.params())           # world > 1: RCCL, captured with the rest of the branch
            self.opt.step()

    def _body(self):
        """Eager form, and what the sequential graph captures: backbone, then head, of the staged batch."""
        self._rotate()
        self._backbone()
        self._head()

    def _body_overlapped(self):
        """What the overlapped graph captures.  Fork / join through the capturing stream: everything the side stream
        does lies between ``side.wait_stream(main)`` and ``main.wait_stream(side)``, i.e. inside the graph."""
        main = torch.cuda.current_stream(self.dev)
        self._rotate()
        self.fmap_head.copy_(self.fmap)     # hand-off: features of the batch now in ``cur`` (computed by the previous call)
        if self.bb_split:                   # one branch per frame, forked from the capturing stream itself (a fork inside a
            self._backbone_per_frame(join=False)      # forked branch crashes hipStreamEndCapture on ROCm 7.2)
            self._head()
            for st in self._frame_streams:
                main.wait_stream(st)
            return
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            self._backbone()                # batch k+1
        self._head()                        # batch k
        main.wait_stream(self._side)

    def prime(self):
        """Overlapped schedule only: backbone pass of the batch staged first, so that the first call's head finds its
        features (``nxt`` <- ``in`` as a call would have done)."""
        if self.overlap and not self.primed:
            self.nxt.buf.copy_(self.inp.buf)
            self._backbone()
            if self.fmap_head is None:
                self.fmap_head = torch.empty_like(self.fmap)
        self.primed = True

    def capture(self, warmup=2, restore=False):
        """Warm up eagerly (sizes the arenas, fills the allocator), then capture the step into ONE HIP graph.
        Returns False (and keeps the eager form, ``graph_error`` says why) when graphs are off or capture fails.
        ``restore``: the warm-up steps are real training steps on the staged batch; put parameters, momentum and the RNG
        state back afterwards (a training loop that must not see them).  ``warmup=0`` re-captures (after a learning-rate
        change: rates live in the captured kernel arguments)."""
        saved = None
        if restore and warmup:
            state = [it["p"].data for it in self.opt.items] + [it["m"] for it in self.opt.items]
            saved = (state, [t.clone() for t in state], torch.cuda.get_rng_state(self.dev))
        try:
            return self._capture(warmup)
        finally:
            if saved is not None:
                torch.cuda.synchronize(self.dev)
                with torch.no_grad():
                    for t, sv in zip(saved[0], saved[1]):
                        t.copy_(sv)
                torch.cuda.set_rng_state(saved[2], self.dev)
                self.opt.bump()

    def _capture(self, warmup):
        if self.tp:
            parallel.assert_same_rows(self.n_rows, "boxes + pairs")
        ov, self.overlap = self.overlap, False          # the warm-up steps are sequential eager steps
        s = torch.cuda.Stream(self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            for i in range(warmup):
                self._body()
                if i == 0:
                    self.ctx_bb.fit()
                    self.ctx_head.fit()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        self.overlap = ov
        if not self.use_graph:
            return False
        if self.bb_split:                               # size the per-frame arenas: two eager per-frame passes
            for _rep in range(2):
                for f, ctx in enumerate(self.ctx_frames):
                    with ctx:
                        with torch.no_grad():
                            self.net.RCNN_base(self.im[f:f + 1])
                    ctx.fit()
            torch.cuda.synchronize(self.dev)
        self.graph = None
        try:
            g = torch.cuda.CUDAGraph()
            if self.overlap:
                self.prime()
                torch.cuda.synchronize(self.dev)
                self._side = torch.cuda.Stream(self.dev)
                with torch.cuda.graph(g):
                    self._body_overlapped()
            else:
                with torch.cuda.graph(g):
                    self._body()
            self.graph = g
            return True
        except Exception as e:      # report, fall back to eager launches
            self.graph = None
            self.overlap = False
            self.graph_error = repr(e)
            torch.cuda.synchronize(self.dev)
            return False

    def __call__(self):
        """One step on the caller's current stream (any stream, the legacy default stream included: see
        ``_graph_launch_guard``).  Returns the device scalar holding the loss of the batch the head just processed."""
        if self.graph is None:
            self._body()
        else:
            _graph_launch_guard()
            self.graph.replay()
        self.opt.bump()
        return self.loss

    def flush(self):
        """Overlapped schedule: run the head of the batch staged last (its backbone pass ran in the previous call)."""
        return self()


def _graph_launch_guard():
    """ROCm 7.2's HIP runtime replays a graph through pre-built AQL packet batches (``DEBUG_CLR_GRAPH_PACKET_CAPTURE``,
    on by default).  On the LEGACY DEFAULT stream that path loses the order between a graph's nodes and the stream's
    other work while a second stream is busy (losses off by 2e-2 from the second step on, NaN
    weights; same graphs correct on any created stream, and correct on the default stream with the packet path off).
    i2vsgg_amd/__init__.py switches the path off before the runtime initialises; if the process had already initialised
    HIP with it on, replaying on the default stream is refused rather than risked."""
    import os
    if os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1") != "0" and \
            torch.cuda.current_stream() == torch.cuda.default_stream():
        raise RuntimeError("HIP graph replay on the legacy default stream with DEBUG_CLR_GRAPH_PACKET_CAPTURE on: set "
                           "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before the first HIP call (importing i2vsgg_amd before "
                           "torch.cuda is initialised does it) or run the step on a stream made with torch.cuda.Stream()")


class _SplitBatch(torch.autograd.Function):
    """(x[:n], x[n:]) along the batch axis as views; the backward writes the two gradients into ONE buffer (autograd's own
    slice backward zero-fills a full-size tensor per slice and adds them)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.shape = n, tuple(x.shape)
        ctx.fmt = torch.channels_last if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) else torch.contiguous_format
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        ref = ga if ga is not None else gb
        out = torch.empty(ctx.shape, device=ref.device, dtype=ref.dtype, memory_format=ctx.fmt)
        for dst, g in ((out[:ctx.n], ga), (out[ctx.n:], gb)):
            if g is None:
                dst.zero_()
            else:
                dst.copy_(g)
        return out, None


class InstanceStyleDStep:
    """One D+G adversarial step (trainval_net_instance_styleD_bilinear.py:262-341): source forward with
    detection + RPN losses and 0.5*mean(d^2) for both discriminators, target forward with
    0.5*mean((1-d)^2), style terms weighted by style_lambda, ONE backward through the gradient-reversal
    layers, one SGD step over every trainable parameter.

    Eager by default with the target layers sampling on the host from the reference's np.random stream (bit parity with
    the reference's RNG contract; two small D2H copies per step).  ``capture()`` switches the target layers to
    device-side sampling (same subsample sizes, torch's generator) and records the whole step -- both forwards, the
    backward, the gradient exchange and the update -- into ONE HIP graph."""

    def __init__(self, net, n_frames, lr=5e-4, eta=0.1, eta_style=0.001, style_lambda=1.0, seed=3, device="cuda:0",
                 h=600, w=1000, n_gt=8, cr=False):
        self.net, self.dev = net, torch.device(device)
        self.cr = cr                  # --cr: consistency regularisation between instance- and image-level D (:299-312)
        self.world = parallel.world_size()
        self.eta, self.eta_style, self.style_lambda = eta, eta_style, style_lambda
        self.geom = (h, w)
        self.n_frames, self.n_gt = n_frames, n_gt
        import os
        self.batched = os.environ.get("I2V_ISD_BATCHED", "1") != "0"
        # filter gradients of the bottleneck nodes on a side branch of the step (ops.WGRAD_STREAM)
        # (only with the one-pass backbone: a filter met twice in one backward would have its two gradients added on the main
        # stream while the side branch may still be writing the first)
        self.wgrad_branch = os.environ.get("I2V_WGRAD_BRANCH", "0") == "1" and self.dev.type == "cuda" and self.batched
        self._wgrad_stream = torch.cuda.Stream(self.dev) if self.wgrad_branch else None
        # the captured step: source and target as two branches of the graph (_body_branches)
        self.branches = os.environ.get("I2V_ISD_BRANCHES", "1") != "0" and self.dev.type == "cuda" and not self.wgrad_branch
        self._branch_streams = [torch.cuda.Stream(self.dev) for _ in range(2)] if self.branches else []
        self.ctx_src = ops.LaunchContext(self.dev) if self.branches else None
        self.ctx_tgt = ops.LaunchContext(self.dev) if self.branches else None
        self.im_s = self.im_t = self.im_st = self.info = self.gt = self.nb = None
        self.reseed(seed)
        self.gt_t = torch.zeros((n_frames, 1, 5), device=self.dev)
        self.nb_t = torch.zeros((n_frames,), device=self.dev)
        self.opt = FusedSGD(list(net.named_parameters()), lr)
        self.ctx = ops.LaunchContext(self.dev)
        self.names = ["total", "det", "dloss_s", "dloss_t", "dloss_s_style", "dloss_t_style"] + \
            (["source_adv_cst", "target_adv_cst"] if cr else [])
        self.losses = {k: torch.zeros((), device=self.dev) for k in self.names}     # static addresses: a captured step writes here
        self.graph = None
        self.graph_error = None
        self._fitted = False

    # ------------------------------------------------------------------ data side
    def stage(self, im_s, info, gt, nb, im_t):
        """Hand the next minibatch to the step: source frames (N,3,H,W) with ``im_info`` (N,3), ``gt_boxes``
        (N,MAX_NUM_GT_BOXES,5) and ``num_boxes`` (N,) -- one roi_data_layer batch -- and N target frames.  Copies into static
        device tensors on the caller's stream (a captured step is bound to their addresses and shapes)."""
        if self.im_s is None:
            t = lambda a: a.to(self.dev, torch.float32).clone()
            self.im_s, self.im_t, self.info, self.gt, self.nb = t(im_s), t(im_t), t(info), t(gt), t(nb)
            self.im_st = torch.cat((self.im_s, self.im_t), 0) if self.batched else None
            return
        if tuple(im_s.shape) != tuple(self.im_s.shape) or tuple(gt.shape) != tuple(self.gt.shape):
            raise ValueError("InstanceStyleDStep.stage: the step is bound to the shapes of the first staged batch")
        self.im_s.copy_(im_s); self.im_t.copy_(im_t); self.info.copy_(info); self.gt.copy_(gt); self.nb.copy_(nb)
        if self.batched:
            n = self.im_s.shape[0]
            self.im_st[:n].copy_(im_s); self.im_st[n:].copy_(im_t)

    def reseed(self, seed):
        """Stage the synthetic minibatch of ``seed`` (frames, 8 GT boxes per source frame)."""
        h, w = self.geom
        ims, info = syn.frames(seed, self.n_frames, h, w)
        imt, _ = syn.frames(seed + 100, self.n_frames, h, w)
        gt, nb = syn.gt_boxes(seed, self.n_frames, self.n_gt, self.net.n_classes, cfg.MAX_NUM_GT_BOXES, h, w)
        f = torch.from_numpy
        self.stage(f(ims), f(info), f(gt), f(nb), f(imt))

    def _body(self):
        net = self.net
        with self.ctx:
            if self.batched:
                # ONE backbone pass over the source and the target frames (the reference makes two, :271 and :293; with
                # frozen BN they are the same arithmetic): half the launches of the trunk's forward, data-gradient and
                # filter-gradient kernels, each over twice the pixels, and no accumulation adds between two backward
                # passes through the same filters
                feat, feat1 = net.extract_feature(self.im_st)
                n = self.im_s.shape[0]
                (fs, ft), (f1s, f1t) = _SplitBatch.apply(feat, n), _SplitBatch.apply(feat1, n)
                out = net.forward_features(fs, f1s, self.info, self.gt, self.nb, False, self.eta, self.eta_style)
            else:
                out = net(self.im_s, self.info, self.gt, self.nb, target=False, eta=self.eta, eta_style=self.eta_style)
            _, _, _, l_rpn_cls, l_rpn_box, l_cls, l_box, _, d_inst, d_style = out
            loss = l_rpn_cls.mean() + l_rpn_box.mean() + l_cls.mean() + l_box.mean()
            dloss_s = 0.5 * torch.mean(d_inst ** 2)
            dloss_s_style = 0.5 * torch.mean(d_style ** 2)
            if self.batched:
                d_inst_t, d_style_t = net.forward_features(ft, f1t, self.info, self.gt_t, self.nb_t, True, self.eta,
                                                           self.eta_style)
            else:
                d_inst_t, d_style_t = net(self.im_t, self.info, self.gt_t, self.nb_t, target=True, eta=self.eta,
                                          eta_style=self.eta_style)
            dloss_t = 0.5 * torch.mean((1 - d_inst_t) ** 2)
            dloss_t_style = 0.5 * torch.mean((1 - d_style_t) ** 2)
            total = loss + dloss_s + dloss_t + self.style_lambda * (dloss_s_style + dloss_t_style)
            vals = dict(det=loss, dloss_s=dloss_s, dloss_t=dloss_t, dloss_s_style=dloss_s_style, dloss_t_style=dloss_t_style)
            if self.cr:
                cst = consistency_terms(d_inst, d_style, d_inst_t, d_style_t)
                total = total + cst["source_adv_cst"] + cst["target_adv_cst"]
                vals.update(cst)
            vals["total"] = total
            self.opt.zero_grad()
            ops.WGRAD_STREAM = self._wgrad_stream if self.wgrad_branch else None
            try:
                (total / self.world).backward()
                ops.join_wgrad_branch()
            finally:
                ops.WGRAD_STREAM = None
            parallel.all_reduce_grads(self.opt.params())
            self.opt.step()
            for k in self.names:
                self.losses[k].copy_(vals[k].detach())

    eager_step = _body

    def _body_branches(self):
        """The captured form by default (``I2V_ISD_BRANCHES=0``: the one-pass form above): the source and the target forward /
        backward as TWO BRANCHES of the step graph -- they are independent until their gradients meet (the reference runs
        them one after the other, :271-296).  An 8-frame layer3 GEMM runs at 102 TF alone; two 4-frame chains side by side
        reach 114 TF (tools/corun_probe.py): one chain's set-up / store phases lie under the other's K loops, with no edge
        between the branches until the join.  Each branch owns a LaunchContext (arena, split-K workspace, scratch) and
        returns its gradients through ``torch.autograd.grad`` (no AccumulateGrad on a shared ``.grad`` from two streams);
        the capturing stream adds them after the join."""
        net, main = self.net, torch.cuda.current_stream(self.dev)
        params = self.opt.params()
        self.opt.zero_grad()
        s_src, s_tgt = self._branch_streams
        vals, grads = {}, {}
        s_src.wait_stream(main)
        s_tgt.wait_stream(main)
        # Each branch's autograd graph is gone before the other branch's forward starts (only detached values leave the
        # block): a parameter's AccumulateGrad node lives as long as a graph references it and remembers the stream it was
        # made on -- shared between the two branches the engine would synchronise their streams with each other.
        def source():
            out = net(self.im_s, self.info, self.gt, self.nb, target=False, eta=self.eta, eta_style=self.eta_style)
            _, _, _, l_rpn_cls, l_rpn_box, l_cls, l_box, _, d_inst, d_style = out
            v = {"det": l_rpn_cls.mean() + l_rpn_box.mean() + l_cls.mean() + l_box.mean(),
                 "dloss_s": 0.5 * torch.mean(d_inst ** 2), "dloss_s_style": 0.5 * torch.mean(d_style ** 2)}
            part = v["det"] + v["dloss_s"] + self.style_lambda * v["dloss_s_style"]
            if self.cr:
                v["source_adv_cst"] = _consistency_term(d_inst, d_style)
                part = part + v["source_adv_cst"]
            g = torch.autograd.grad(part / self.world, params, allow_unused=True)
            v["_src"] = part
            net.RCNN_rpn.rpn_loss_cls = net.RCNN_rpn.rpn_loss_box = 0      # the module keeps its last losses (rpn.py:89-108): they hold the graph
            return {k: t.detach() for k, t in v.items()}, g

        def target():
            d_inst_t, d_style_t = net(self.im_t, self.info, self.gt_t, self.nb_t, target=True, eta=self.eta,
                                      eta_style=self.eta_style)
            v = {"dloss_t": 0.5 * torch.mean((1 - d_inst_t) ** 2), "dloss_t_style": 0.5 * torch.mean((1 - d_style_t) ** 2)}
            part = v["dloss_t"] + self.style_lambda * v["dloss_t_style"]
            if self.cr:
                v["target_adv_cst"] = _consistency_term(d_inst_t, d_style_t)
                part = part + v["target_adv_cst"]
            g = torch.autograd.grad(part / self.world, params, allow_unused=True)
            v["_tgt"] = part
            return {k: t.detach() for k, t in v.items()}, g

        with torch.cuda.stream(s_src), self.ctx_src:
            v, grads["s"] = source()
            vals.update(v)
        with torch.cuda.stream(s_tgt), self.ctx_tgt:
            v, grads["t"] = target()
            vals.update(v)
        main.wait_stream(s_src)
        main.wait_stream(s_tgt)
        both = [(a, b) for a, b in zip(grads["s"], grads["t"]) if a is not None and b is not None]
        if both:
            torch._foreach_add_([a for a, _ in both], [b for _, b in both])
        for p, a, b in zip(params, grads["s"], grads["t"]):
            p.grad = a if a is not None else b
        vals["total"] = vals.pop("_src") + vals.pop("_tgt")
        parallel.all_reduce_grads(params)
        self.opt.step()
        for k in self.names:
            self.losses[k].copy_(vals[k].detach())

    def _device_sampling(self, on):
        h, w = self.geom
        atl = self.net.RCNN_rpn.RPN_anchor_target
        atl.device_sampling = on
        atl.image_size = (h, w) if on else None
        self.net.RCNN_proposal_target.device_sampling = on

    def capture(self, warmup=2, restore=False):
        """Device-side target sampling, eager warm-up, then the whole step as ONE HIP graph.  False (eager form kept,
        ``graph_error`` says why) when capture fails.  ``restore``: the warm-up steps are real training steps on the staged
        batch; put parameters, momentum and the RNG state back afterwards.  ``warmup=0`` re-captures (after a learning-rate
        change: the rates live in the captured kernel arguments)."""
        saved = None
        if restore and warmup:
            state = [it["p"].data for it in self.opt.items] + [it["m"] for it in self.opt.items]
            saved = (state, [t.clone() for t in state], torch.cuda.get_rng_state(self.dev))
        try:
            return self._capture(warmup)
        finally:
            if saved is not None:
                torch.cuda.synchronize(self.dev)
                with torch.no_grad():
                    for t, sv in zip(saved[0], saved[1]):
                        t.copy_(sv)
                torch.cuda.set_rng_state(saved[2], self.dev)
                self.opt.bump()

    def _capture(self, warmup):
        self._device_sampling(True)
        body = self._body_branches if self.branches else self._body
        s = torch.cuda.Stream(self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            for i in range(warmup):
                body()
                if i == 0:
                    for ctx in (self.ctx, self.ctx_src, self.ctx_tgt):
                        if ctx is not None:
                            ctx.fit()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            self.graph = g
            return True
        except Exception as e:
            self.graph = None
            self.graph_error = repr(e)
            torch.cuda.synchronize(self.dev)
            return False

    def __call__(self):
        if self.graph is None:
            self._body()
            if not self._fitted:            # eager use: size the arena of atomically accumulated outputs after the first step
                self.ctx.fit()
                self._fitted = True
        else:
            _graph_launch_guard()
            self.graph.replay()
            self.opt.bump()
        return self.losses["total"]


def _consistency_term(di, ds):
    """One domain's term of ``consistency_terms`` below."""
    per_roi = di.mean(3).mean(2)
    rois_per_frame = per_roi.shape[0] // ds.shape[0]
    prob = ds.reshape(ds.shape[0], -1)[:, :1].repeat(1, rois_per_frame).view(-1, 1)
    return torch.nn.functional.mse_loss(per_roi, prob.detach())


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


def build_instance_styled_net(layers=101, n_cls=16, seed=0, device="cuda:0", ic=False, gc=False, class_agnostic=False):
    from .model.faster_rcnn.resnet_instance_styleD_bilinear import resnet
    torch.manual_seed(seed)
    net = resnet(tuple(range(n_cls)), layers, class_agnostic=class_agnostic, ic=ic, gc=gc)
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
