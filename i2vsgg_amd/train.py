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

from . import _lib, ops, parallel
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
        parallelism the gradient must be all-reduced before the update).  Returns the fused names.
        (Rounds 3-5 carried a second form, ``defer``: the update applied by the NEXT forward on its pass over the filter --
        parity-tested, 1.41 ms against 0.43 + 0.79 as two kernels, never used; it left the tree in round 6, DESIGN_HISTORY.md 5.6.)"""
        names = []
        for it in self.items:
            p = it["p"]
            if parallel.exchange_enabled() and not parallel.is_local(p):
                continue              # its gradient has to cross the ranks first
            if p.dim() >= 2 and p.numel() >= min_numel:
                # keyed by storage pointer (what the autograd node sees); ``owner`` says whose entry it is -- a pointer is
                # reused by the allocator, and an optimizer that is collected late must not remove (or act on) the entry a
                # newer optimizer made for a new filter at the same address
                ops.FUSED_SGD[p.data_ptr()] = ops.FusedEntry(it["m"], it["lr"], self.momentum, it["wd"], self)
                self._fused_keys.append(p.data_ptr())
                names.append(it["name"])
        return names

    def _mine(self, table, k):
        ent = table.get(k)
        return ent if ent is not None and getattr(ent, "owner", None) is self else None

    def flush_pending(self):
        """Nothing is pending: every update is applied inside the step (rounds 3-5 had a deferred form for fc6 / fc7; callers that
        read filters outside the step keep calling this)."""

    def pending_state(self):
        return [], []

    def restore_pending(self, host):
        pass

    def unfuse(self):
        for k in self._fused_keys:
            if self._mine(ops.FUSED_SGD, k) is not None:      # not an entry a newer optimizer made at a reused address
                del ops.FUSED_SGD[k]
        self._fused_keys = []

    def __del__(self):
        try:
            self.unfuse()
        except Exception:
            pass

    def params(self):
        return [it["p"] for it in self.items]

    def state_tensors(self):
        """Every tensor of the optimizer's own state (a step object snapshots / restores them around warm-up steps)."""
        return [it["m"] for it in self.items]

    @staticmethod
    def bump():
        """The parameters changed (an eager ``step()``, a fused wgrad+SGD epilogue or a graph replay that contains
        them): whatever is derived from trained parameters and cached (Winograd-domain filters) is stale."""
        ops.PARAM_EPOCH += 1

    def state_dict(self):
        """torch.optim.SGD's layout (param_groups + state[i]['momentum_buffer']) in named_parameters order, so that a
        checkpoint written here resumes under torch.optim.SGD and vice versa."""
        self.flush_pending()
        return {"state": {i: {"momentum_buffer": it["m"].detach().clone()} for i, it in enumerate(self.items)},
                "param_groups": [{"lr": it["lr"], "momentum": self.momentum, "weight_decay": it["wd"], "params": [i],
                                  "name": it["name"]} for i, it in enumerate(self.items)]}

    def load_state_dict(self, sd):
        self.flush_pending()
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
                if it["p"].data_ptr() == k and self._mine(ops.FUSED_SGD, k) is not None:
                    ops.FUSED_SGD[k] = ops.FusedEntry(it["m"], it["lr"], self.momentum, it["wd"], self)

    def zero_grad(self):
        for it in self.items:
            it["p"].grad = None

    def lr_of(self, name):
        """The learning rate the optimizer holds for parameter ``name`` (what a resumed run shows and decays from)."""
        for it in self.items:
            if it["name"] == name:
                return it["lr"]
        return self.items[0]["lr"]

    def scale_lr(self, k):
        self.flush_pending()            # a pending update belongs to the step that computed it: applied at that step's rate
        for it in self.items:
            it["lr"] *= k
            ent = self._mine(ops.FUSED_SGD, it["p"].data_ptr())
            if ent is not None:                 # fused entries hold the rate by value (a captured graph holds it too:
                ops.FUSED_SGD[it["p"].data_ptr()] = ops.FusedEntry(ent[0], it["lr"], ent[2], ent[3], self)   # re-capture after a decay)

    MULTI_BELOW = 1 << 20       # tensors under 1 Mi elements share one launch

    @torch.no_grad()
    def step(self):
        small = []
        for it in self.items:
            p, g = it["p"], it["p"].grad
            if g is None:
                continue
            if g.stride() != p.stride() and not _same_memory_order(p, g):
                g = torch.empty_like(p).copy_(g)
            if p.numel() < self.MULTI_BELOW:
                small.append((p, g, it))
            else:
                ops.sgd_momentum_(p, g, it["m"], it["lr"], self.momentum, it["wd"])
        if small:
            ops.sgd_momentum_multi_([p for p, _, _ in small], [g for _, g, _ in small], [it["m"] for _, _, it in small],
                                    [it["lr"] for _, _, it in small], [it["wd"] for _, _, it in small], self.momentum)
        self.bump()


class FusedAdam(FusedSGD):
    """torch.optim.Adam with the reference's param groups (``--o adam``: trainval_net_instance_styleD_bilinear.py:143-145,
    trainval_net_SGG_emb.py:146-147) on ``i2v_adam_multi``: same interface as ``FusedSGD`` towards the step objects, no fusion
    into the filter-gradient kernels (the second-moment update needs the finished gradient).  The step count sits in device
    memory, so a captured step replays with the right bias corrections; ``state_dict`` is torch.optim.Adam's layout."""

    def __init__(self, named_params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=None):
        super().__init__(named_params, lr, momentum=0.0, weight_decay=weight_decay)
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        for it in self.items:
            it["v"] = torch.zeros_like(it["p"])
        dev = self.items[0]["p"].device if self.items else "cpu"
        self.t = torch.zeros(1, dtype=torch.int32, device=dev)

    def fuse_wgrad(self, min_numel=1 << 24, defer=None):
        return []

    def state_tensors(self):
        return [it["m"] for it in self.items] + [it["v"] for it in self.items] + [self.t]

    def state_dict(self):
        # torch.optim.Adam holds state only for parameters that have received a gradient (round-3 advice); its per-parameter
        # ``step`` is one shared device counter here -- every trained parameter of the two reference models gets a gradient
        # every step, so the two agree
        step = float(self.t.item())
        return {"state": {i: {"step": torch.tensor(step), "exp_avg": it["m"].detach().clone(), "exp_avg_sq": it["v"].detach().clone()}
                          for i, it in enumerate(self.items) if it.get("seen")},
                "param_groups": [{"lr": it["lr"], "betas": self.betas, "eps": self.eps, "weight_decay": it["wd"], "amsgrad": False,
                                  "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                                  "params": [i], "name": it["name"]} for i, it in enumerate(self.items)]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        flat = [pi for g in groups for pi in g["params"]]
        if len(flat) != len(self.items):
            raise ValueError("optimizer state holds %d parameters, this optimizer %d" % (len(flat), len(self.items)))
        by_param = {pi: g for g in groups for pi in g["params"]}
        step = 0.0
        for i, it in enumerate(self.items):
            g = by_param[flat[i]]
            it["lr"], it["wd"] = float(g["lr"]), float(g.get("weight_decay", it["wd"]))
            self.betas, self.eps = tuple(float(b) for b in g.get("betas", self.betas)), float(g.get("eps", self.eps))
            st = sd["state"].get(flat[i], sd["state"].get(str(flat[i])))
            if st is not None and st.get("exp_avg") is not None:
                it["m"].copy_(st["exp_avg"].reshape(it["m"].shape))
                it["v"].copy_(st["exp_avg_sq"].reshape(it["v"].shape))
                step = max(step, float(st.get("step", 0.0)))
                it["seen"] = True
            else:
                it["m"].zero_()
                it["v"].zero_()
                it["seen"] = False
        self.t.fill_(int(step))

    def scale_lr(self, k):
        for it in self.items:
            it["lr"] *= k

    @torch.no_grad()
    def step(self):
        live = []
        for it in self.items:
            p, g = it["p"], it["p"].grad
            if g is None:                        # torch.optim.Adam skips parameters without a gradient
                continue
            if g.stride() != p.stride() and not _same_memory_order(p, g):
                g = torch.empty_like(p).copy_(g)
            it["seen"] = True
            live.append((p, g, it))
        if not live:
            return
        ops.adam_step_(self.t)
        ops.adam_multi_([p for p, _, _ in live], [g for _, g, _ in live], [it["m"] for _, _, it in live],
                        [it["v"] for _, _, it in live], [it["lr"] for _, _, it in live], [it["wd"] for _, _, it in live],
                        self.betas, self.eps, self.t)
        self.bump()


def make_optimizer(kind, named_params, lr):
    """``--o sgd | adam`` of the reference loops."""
    if kind == "sgd":
        return FusedSGD(named_params, lr)
    if kind == "adam":
        return FusedAdam(named_params, lr)
    raise ValueError("optimizer %r: the reference loops know 'sgd' and 'adam'" % (kind,))


def _same_memory_order(p, g):
    """Two dense tensors of one shape whose strides agree on every axis longer than 1 hold their elements in the same order
    in memory (a (Cout,Cin,1,1) filter gradient in channels_last strides against the parameter's plain strides): the flat
    update kernels may read both as they are.  Without this every 1x1 filter gradient of the trunk was copied once per step
    (45 launches, 0.2 ms of the instance_styleD step: tools/glue_trace.py)."""
    if p.shape != g.shape or p.numel() != g.numel():
        return False
    for n, sp, sg in zip(p.shape, p.stride(), g.stride()):
        if n > 1 and sp != sg:
            return False
    dense = lambda t: t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
    return dense(p) and dense(g)


def synthetic_sgg_batch(seed, n_frames, n_boxes=32, n_pairs=32, n_rel=62, n_cls=16, h=600, w=1000):
    """frames + per-frame annotation dicts (keys ``f0..``) + im_info."""
    im, info = syn.frames(seed, n_frames, h, w)
    annos = {"f%d" % i: syn.relation_annotation(seed * 1000 + i, n_boxes, n_pairs, n_rel, n_cls, h, w)
             for i in range(n_frames)}
    return im, info, annos


def sgg_head_inputs(annos, info, n_rel):
    """Head inputs of one minibatch on the host, exact sizes (faster_rcnn_SGG_emb.py:170-245 for every frame of the batch;
    frames without an annotated relation contribute nothing, :177-183): ``annos`` one annotation dict per frame (unscaled
    pixel boxes, as in the ``source_gt_rels`` pickle), ``info`` (n_frames,3) im_info rows [h, w, scale].
    -> dict of numpy arrays: boxes (nb,5), relb (np,5) [frame index in column 0], labels (np,n_rel), ixs / ixo (np,) rows of
    ``boxes``, bounds (np,2,4) integer bounds of the 32x32 dual masks, wrow (np,) = 1 / (pairs of the frame * frames with
    pairs): sum_r wrow[r] * mean_c BCE is the mean over frames of the per-frame BCE mean."""
    from .model.faster_rcnn.faster_rcnn_SGG_emb import build_pair_tables
    boxes, relb, bounds, labels, ixs, ixo, counts, off = [], [], [], [], [], [], [], 0
    for f, anno in enumerate(annos):
        if anno is None or len(anno["rels"]) < 1:
            continue
        gt, union, bnd, lab, s, o = build_pair_tables(anno, float(info[f][2]), float(info[f][0]), float(info[f][1]), n_rel)
        b5 = np.zeros((gt.shape[0], 5), np.float32); b5[:, 0] = f; b5[:, 1:] = gt
        r5 = np.zeros((union.shape[0], 5), np.float32); r5[:, 0] = f; r5[:, 1:] = union
        boxes.append(b5); relb.append(r5); bounds.append(bnd); labels.append(lab)
        ixs.append(s + off); ixo.append(o + off); counts.append(lab.shape[0]); off += gt.shape[0]
    if not counts:
        return None
    cat = np.concatenate
    return dict(boxes=cat(boxes), relb=cat(relb), labels=cat(labels).astype(np.float32), ixs=cat(ixs).astype(np.int64),
                ixo=cat(ixo).astype(np.int64), bounds=cat(bounds).astype(np.int32),
                wrow=cat([np.full((c,), 1.0 / (c * len(counts)), np.float32) for c in counts]))


def _rasterize_host(bounds, channels=4):
    """(n,2,4) integer [x1,y1,x2,y2) -> (n,channels,32,32) float32 dual masks (resnet_SGG_emb.py:246-256); channels 2.. are
    the zero pad that keeps conv_lo.0's gathers 16-byte wide."""
    n = bounds.shape[0]
    m = np.zeros((n, channels, 32, 32), np.float32)
    ar = np.arange(32)
    b = bounds.reshape(n, 2, 4, 1)
    xs = (ar[None, None, :] >= b[:, :, 0]) & (ar[None, None, :] < b[:, :, 2])          # (n,2,32)
    ys = (ar[None, None, :] >= b[:, :, 1]) & (ar[None, None, :] < b[:, :, 3])
    m[:, :2] = (ys[:, :, :, None] & xs[:, :, None, :]).astype(np.float32)
    return m


class _Slot:
    """One minibatch worth of head inputs packed into ONE device buffer (256-B aligned fields), so that moving a
    batch between pipeline stages is a single copy whatever the number of fields.  ``layout``: name -> (shape, dtype).
    ``host=True`` adds two pinned host mirrors of the same layout: a batch is assembled in one of them and crosses PCIe as
    ONE asynchronous copy."""

    def __init__(self, layout, device, host=False):
        self.layout = {k: (tuple(sh), dt) for k, (sh, dt) in layout.items()}
        self.spec, off = [], 0
        for name, (shape, dt) in self.layout.items():
            nbytes = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
            self.spec.append((name, off, nbytes, dt, shape))
            off += (nbytes + 255) // 256 * 256
        self.nbytes = max(off, 256)
        self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.views = {name: self.buf[o:o + n].view(dt).view(shape) for name, o, n, dt, shape in self.spec}
        self._host, self._turn = [], 0
        if host:
            for _ in range(2):
                hb = torch.zeros(self.nbytes, dtype=torch.uint8).pin_memory()
                hv = {name: hb[o:o + n].view(dt).view(shape).numpy() for name, o, n, dt, shape in self.spec}
                self._host.append((hb, hv, torch.cuda.Event()))

    def same_layout(self, layout):
        return self.layout == {k: (tuple(sh), dt) for k, (sh, dt) in layout.items()}

    def write(self, fields):
        """Device tensors of exactly the slot's shapes (one small copy per field)."""
        for name, t in fields.items():
            self.views[name].copy_(t)

    def write_host(self, fields):
        """numpy arrays, each at most as large as its field along axis 0: zero-padded to the slot's capacity in a pinned
        mirror, then ONE asynchronous H2D copy on the current stream."""
        hb, hv, ev = self._host[self._turn]
        self._turn ^= 1
        ev.synchronize()                         # the copy that last read this mirror has finished (two calls ago)
        for name, a in fields.items():
            dst = hv[name]
            n = a.shape[0] if a.ndim else 0
            if a.ndim and n > dst.shape[0]:
                raise ValueError("field %s: %d rows exceed the slot's capacity %d" % (name, n, dst.shape[0]))
            if a.ndim:
                dst[:n] = a
                dst[n:] = 0
            else:
                dst[...] = a
        self.buf.copy_(hb, non_blocking=True)
        ev.record()


class _Uploader:
    """Host frames -> device on the process's COPY stream (ops.role_stream), two staging buffers and event edges both ways: the
    transfer of minibatch k+1 runs beside the step that is still computing (a 2 x 3 x 600 x 1000 fp32 minibatch is 14.4 MB;
    bench.py --data loader over four alternating frame sizes: 5.00 -> 4.82 ms per step, uint8 frames 4.83 -> 4.77).
    ``I2V_UPLOAD_STREAM=0``: the transfer on the caller's stream, in front of the step (the default of round 3, which had met
    a host segfault in hipGraphLaunch with the copy stream and blamed stream aliasing).  The same file order with pooled
    streams and every alias logged -- the copy stream WAS a captured branch, the side stream WAS torch's capture stream --
    neither crashes (profiles/r04_alias_repro.txt; that record stopped on a bookkeeping KeyError of the test before the numeric
    comparison) nor changes a loss or a weight (profiles/r05_alias_repro.txt: run to the end, 18 passed) once no graph is
    dropped while a replay of it may be in flight (``invalidate_graphs`` synchronises first; stage() grew the head capacity
    and dropped every graph right behind an asynchronous replay).  The aliases cost the overlap, not correctness; they are gone too (ops.role_stream),
    and tests/test_gpu_data_layer.py runs the loader loop both ways, in the order that crashed.
    ``upload`` returns a device tensor that is valid on the caller's CURRENT stream, ``consumed`` marks the point after which
    its buffer may be overwritten."""

    def __init__(self, device):
        self.dev = torch.device(device)
        # Normal priority.  Measured (tools/loader_probe.py, relation step, 14.4 MB of frames per step): the transfer costs the
        # step 0.35-0.4 ms although it is queued a step ahead on its own stream -- it runs as a blit kernel and only gets its
        # turn when the step's branches drain.  A HIGH-priority copy stream (I2V_UPLOAD_PRIORITY=-1) hides it when every
        # minibatch has one size (4.74 -> 4.86 ms instead of 5.15) but doubles the step (8.7-9.5 ms) as soon as the loop
        # alternates between the graphs of two sizes -- so it is not the default.
        import os
        self.enabled = os.environ.get("I2V_UPLOAD_STREAM", "1") == "1"
        # the copy stream exists only when asked for, and is the process's ONE copy stream (ops.role_stream): a handle of the
        # library's own, never an alias of a branch / capture / communicator stream out of torch's pool
        self.stream = ops.role_stream(self.dev, "copy", 0) if self.enabled else None      # normal priority: a high one doubles the step when the loop alternates between the graphs of two sizes
        self.rings = {}

    def upload(self, frames):
        if not self.enabled:                 # the transfer on the caller's stream, in front of the step (the round-2 form)
            return frames.to(self.dev, non_blocking=True), None
        return self._upload(frames)

    def _upload(self, frames):
        # ONE ring of two byte buffers for every shape: upload k+2 waits for the consumer of upload k whatever their shapes, so
        # at most two transfers are ever queued ahead of the step
        nbytes = frames.numel() * frames.element_size()
        ring = self.rings.setdefault("ring", {"i": 0, "buf": [None, None], "free": [None, None]})
        i = ring["i"]
        ring["i"] ^= 1
        cur = torch.cuda.current_stream(self.dev)
        if ring["free"][i] is not None:
            self.stream.wait_event(ring["free"][i])          # the copy that last READ this buffer (two uploads ago) is done
        with torch.cuda.stream(self.stream):
            if ring["buf"][i] is None or ring["buf"][i].numel() < nbytes:
                # allocated ON the copy stream (the caching allocator hands a block only to work ordered behind its previous
                # use on the stream it was allocated for) and known to the consumer's stream, so that a release -- growth
                # here, or the end of the step object -- waits for both
                ring["buf"][i] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=self.dev)
            ring["buf"][i].record_stream(cur)
            dst = ring["buf"][i][:nbytes].view(frames.dtype).view(frames.shape)
            dst.copy_(frames, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        cur.wait_event(done)
        return dst, (ring, i)

    def consumed(self, token):
        if token is None:
            return
        ring, i = token
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        ring["free"][i] = ev


def _place_u8(uploader, frames_u8, meta, dst):
    """The device front-end of a ``roibatchLoader(device_prep=True)`` minibatch: every decoded uint8 frame crosses PCIe as it
    is (copy stream) and ``i2v_image_prep`` writes the mean-subtracted, resized BGR image into its slot of ``dst`` (n,4,H,W)
    channels_last, which is cleared first (the canvas around an image is zero padding, roibatchLoader.py:162-181)."""
    dst.zero_()
    for f, u8 in enumerate(frames_u8):
        flipped, target = bool(meta[f][0]), int(meta[f][4])
        src, token = uploader.upload(u8)
        ops.image_prep(src, cfg.PIXEL_MEANS, target, flipped=flipped, rgb=True, blob=dst[f:f + 1])
        uploader.consumed(token)


class _FrameSet:
    """What the backbone half of the SGG_emb step owns for ONE minibatch size (n, H, W): the staged frames, one launch
    context per frame branch and the captured graph of the step whose backbone half has this size."""

    def __init__(self, key, device, n_ctx, arena):
        n, h, w = key
        self.key = key
        self.im = torch.zeros((n, 4, h, w), device=device).contiguous(memory_format=torch.channels_last)
        self.ctx = [ops.LaunchContext(device, arena=arena) for _ in range(n_ctx)]
        self.fh = self.fw = None      # extent of the C4 map, known after the first pass
        self.graph = None             # None: not captured yet; False: capture failed (eager launches for this size)
        self.fitted = False
        self.tick = 0
        # stage-split schedule (round 6): this size as the FRONT half's (stem .. layer3[:cut]) and as the BACK half's
        # (layer3[cut:]) input; a graph is captured per (front size, back size) pair -- the back half of a call works on the
        # previous minibatch, whose size may differ
        self.ctx_front = ops.LaunchContext(device, arena=arena)
        self.ctx_back = ops.LaunchContext(device, arena=arena)
        self.fitted_front = self.fitted_back = False
        self.graphs = {}              # back size key -> graph (or False)


class _StagePipeline:
    """Host-side twin of the three head-input slots of the stage-split schedule (in -> mid -> cur): which minibatch each holds and
    which one the head trained last.  No device state: ``tests/test_host_logic.py`` drives it on the CPU.

    A minibatch reaches the head two calls after it was staged.  ``prime()`` fills every slot with the minibatch staged first, so a
    resident minibatch is trained by every call; once a NEWER minibatch waits behind one the head has already trained, the next
    call runs no head (``bubble``) -- in a loop that stages before every call that is exactly its second call."""

    def __init__(self):
        self.inp = self.mid = self.cur = 0
        self.last_trained = -1

    def staged(self):
        self.inp += 1

    def prime(self):
        self.mid = self.cur = self.inp
        self.last_trained = -1

    @property
    def bubble(self):
        return self.cur <= self.last_trained and max(self.mid, self.inp) > self.cur

    def call(self):
        """One call of the step: -> the minibatch its head trains, or None for a bubble; the slots advance."""
        head = None if self.bubble else self.cur
        if head is not None:
            self.last_trained = head
        self.cur, self.mid = self.mid, self.inp
        return head


class SGGEmbStep:
    """One step of trainval_net_SGG_emb.py:189-255 (pre_det) as a replayable object.

    A step = one backbone pass (no grad: the reference detaches the feature map, faster_rcnn_SGG_emb.py:148), one
    relation-head forward + backward, one gradient exchange (world > 1), one SGD(momentum) update of ``vrd.*``.

    Schedule (``overlap``, the default with HIP graphs): the backbone is frozen in this loop, so the backbone pass of
    the NEXT minibatch does not depend on this step's update.  The whole step is ONE captured graph with branches
    between a fork and a join: [head fwd + bwd (+ exchange) + SGD of batch k] beside [backbone of batch k+1, one branch per
    frame].  The branches own disjoint device state (``ops.LaunchContext``: zero arena, split-K workspace, scratch) and
    meet only at graph edges: the feature-map hand-off (one copy before the fork) and the join.  There is one graph launch
    per step on the caller's stream, no side stream, no event and no priority for a caller to get wrong.

    Round 6 (``bb_split="stage"``, the default): the backbone half is cut by STAGE, not by frame -- [head of batch k-2] beside
    [stem .. layer3[:cut] of batch k, all frames] beside [layer3[cut:] of batch k-1, all frames]; a minibatch reaches the head two
    calls after its ``stage()`` (``lag`` 2, three slots in -> mid -> cur; ``bubble`` / ``run_staged`` for loops).  The paragraphs
    below describe the one-call pipeline of ``bb_split="frame"`` / ``"none"``; the stage form differs only in the extra slot.

    Minibatches move through a two-slot pipeline so that ``stage()`` may be called at any time between steps:
    ``stage(b)`` writes the frames of b (read by the NEXT call's backbone branches) and its head inputs into the ``in``
    slot; a call runs head(``cur``) beside backbone(frames) and ends with cur <- in (one small copy inside the graph, behind
    the head).  A batch staged before call k is therefore consumed by the backbone in call k and by the head in call k+1 --
    features and boxes / labels of one batch always meet.  ``overlap=False`` (and eager mode): backbone and head of the
    staged batch in the same call.

    Minibatches of a data loader differ in size (the loader pads every batch to its own aspect ratio,
    roibatchLoader.py:162-190) and in the number of boxes and pairs per frame.  The captured step takes them all:
      * the head half is size-free: boxes / pairs are zero-padded to a CAPACITY (``n_boxes`` / ``n_pairs`` per frame; pad
        rows carry loss weight 0, so their gradient is exactly zero), the feature maps sit packed at the start of a
        capacity buffer and the ROI op reads their extent from device memory (``ops.PackedMaps``; the extent travels
        through the slots with the batch);
      * the backbone half is captured once per frame size: ``shapes[(n, H, W)]`` holds the staged frames, the per-frame
        launch contexts (arenas sized for that size) and the graph [rotate, hand-off, fork, head | backbone(n, H, W), join];
        a call replays the graph of the size staged last.  At most ``max_graphs`` graphs are kept (least recently used
        goes first); they share one memory pool, since no two of them ever run at the same time.
    """

    def __init__(self, net, n_frames, vrd_lr=1e-4, seed=1, device="cuda:0", h=600, w=1000, n_boxes=32, n_pairs=32,
                 use_graph=True, fuse_sgd=True, zero_arena=True, overlap=None, max_graphs=16, trace_rows=0, stage_synthetic=True,
                 optimizer="sgd", bb_split=None):
        import os
        self.net, self.dev, self.n_frames = net, torch.device(device), n_frames
        if not (net.vrd.use_obj_visual and net.vrd.spatial_type == 2):
            raise ValueError("SGGEmbStep packs the inputs of the reference's default relation head (use_obj_visual=True, "
                             "spatial_type=2: what every reference script runs); the other vrd variants run through "
                             "vrd.forward / forward_device")
        self.world = parallel.world_size()
        self.geom = (h, w, n_boxes, n_pairs)
        # data parallelism for everything except vrd.fc6, which is cut by output columns (parallel.py): its 822 MB
        # gradient -- 91 % of the exchange -- stays local and its SGD update stays fused into the wgrad epilogue
        self.tp = parallel.exchange_enabled() and os.environ.get("I2V_TP_FC6", "1") != "0" and \
            net.vrd.fc6.fc.weight.shape[0] % max(self.world, 1) == 0
        if self.tp and net.vrd.tp is None:
            net.vrd.enable_fc6_tp(parallel.rank(), self.world)
        self.opt = make_optimizer(optimizer, [(n, p) for n, p in net.named_parameters() if n.startswith("vrd.")], vrd_lr)
        self.fused = self.opt.fuse_wgrad() if fuse_sgd else []
        self.loss = torch.zeros((), device=self.dev)
        self._seed = torch.full((), 1.0 / max(self.world, 1), device=self.dev)
        self.graph_error = None
        self.use_graph = use_graph
        if overlap is None:
            overlap = os.environ.get("I2V_OVERLAP", "1") != "0"
        self.overlap = bool(overlap) and use_graph       # the schedule asked for; in force once capture() has succeeded
        self._pipelined = False                          # (until then a call is the sequential eager step)
        self._graphs_on = False
        self.zero_arena = zero_arena
        # How the backbone half is cut into graph branches beside the head (I2V_BB_SPLIT / ``bb_split``):
        #   "stage" (default, round 6): by STAGE -- [stem .. layer3[:cut] of batch k+2] and [layer3[cut:] of batch k+1], every
        #           kernel over ALL frames of its minibatch; a batch reaches the head two calls after it was staged (``lag`` 2);
        #   "frame" (rounds 2-5): one branch per frame of batch k+1 (``lag`` 1);
        #   "none": one branch for the whole backbone of batch k+1.
        mode = bb_split if bb_split is not None else os.environ.get("I2V_BB_SPLIT", "stage")
        mode = {"1": "frame", "0": "none", "2": "stage"}.get(str(mode), str(mode))
        if mode not in ("stage", "frame", "none"):
            raise ValueError("SGGEmbStep: bb_split must be 'stage', 'frame' or 'none', got %r" % (mode,))
        self.stage_split = mode == "stage" and use_graph
        self.bb_split = mode == "frame" and use_graph and n_frames > 1
        # blocks of layer3 in the front half: 6-7 of ResNet-101's 23 balance the chains (tools/stage_split_probe.py); never more
        # than a third of the stage (ResNet-50 has 6)
        n3 = len(net.RCNN_base[6])
        self.cut = max(1, min(int(os.environ.get("I2V_STAGE_CUT", "6")), max(n3 // 3, 1), n3 - 1))
        self._frame_streams = [ops.role_stream(self.dev, ("frame", f)) for f in range(max(n_frames, 2))] \
            if (self.bb_split or self.stage_split) else []
        # host-side twins of the three head-input slots (which minibatch each holds) and of what the head has trained: with
        # the backbone cut by stage a batch reaches the head two calls after its stage(); in a loop that stages a new batch
        # before every call the second call would find the FIRST batch in ``cur`` again -- that call runs no head (``bubble``)
        self._pipe, self.n_bubbles = _StagePipeline(), 0
        self._mid_key = None          # key of the frame set whose front-half output ``mid_next`` holds
        self.mid_next_flat = self.mid_cur_flat = None
        self.mid_slot = None
        self._side = None
        self.ctx_head = ops.LaunchContext(self.dev, arena=zero_arena, ordered=True)      # head branch: bit-reproducible sums
        self.ctx_bb = ops.LaunchContext(self.dev, arena=zero_arena)        # eager backbone passes (any size)
        self.shapes, self.max_graphs, self._tick, self._pool = {}, int(max_graphs), 0, None
        self.cap_boxes, self.cap_pairs = n_frames * n_boxes, n_frames * n_pairs      # rows of the padded head inputs
        self.cap_cells = 0
        self.fmap_flat = self.fmap_head_flat = None
        self.cur = self.inp = None    # head inputs: ``inp`` is written by stage(), ``cur`` read by the head
        self._uploader = None
        self._staged = None           # key of the frame set staged last
        self._fmap_key = None         # key of the frame set whose features ``fmap_flat`` holds
        self.primed = False
        self.trace = torch.zeros((trace_rows, len(self.TRACE_COLS)), device=self.dev, dtype=torch.float64) if trace_rows else None
        self._trace_i = torch.zeros((1,), device=self.dev, dtype=torch.long)
        if stage_synthetic:           # a loop fed by a data loader stages its own first minibatch (``stage_batch``)
            self.reseed(seed)

    @property
    def lag(self):
        """How many calls after its ``stage()`` a minibatch reaches the head: 0 sequential, 1 overlapped with the backbone cut by
        frame (or not at all), 2 with the backbone cut by stage.  After ``capture()`` the first ``lag - 1`` calls run no head
        (``bubble``: the pipeline fills); a loop ends with ``lag`` calls without a new ``stage()`` (``flush``)."""
        return (2 if self.stage_split else 1) if self._pipelined else 0

    @property
    def bubble(self):
        """True when the NEXT call will run no head (backbone cut by stage): the head's slot holds a minibatch it has already
        trained while a newer one waits behind it in the pipeline -- the call advances the backbone halves only and its return
        value is stale (``n_bubbles`` counts them).  In a loop that stages before every call that is the second call; a resident
        minibatch (nothing newer staged) is trained by every call."""
        return bool(self._pipelined and self.stage_split and self._pipe.bubble)

    # ------------------------------------------------------------------ compatibility views
    @property
    def graph(self):
        """The captured graph of the frame size staged last (None when that size runs on eager launches)."""
        fs = self.shapes.get(self._staged)
        return (fs.graph or None) if fs is not None else None

    @property
    def im(self):
        return self.shapes[self._staged].im

    @property
    def fmap(self):
        """Feature maps of the last backbone pass as a (n,1024,h,w) channels_last view."""
        fs = self.shapes[self._fmap_key]
        n, c = fs.key[0], self._channels
        return self.fmap_flat[:n * fs.fh * fs.fw * c].view(n, fs.fh, fs.fw, c).permute(0, 3, 1, 2)

    # ------------------------------------------------------------------ data side
    def _layout(self, nb, npair):
        f32, i64 = torch.float32, torch.long
        # boxes and union boxes in ONE roi table (the head pools them in one pass), subject and object indices in one index
        # vector: what the head would otherwise concatenate every step
        return {"rois": ((nb + npair, 5), f32), "labels": ((npair, self.net.vrd.n_rel), f32), "ix12": ((2 * npair,), i64),
                "masks": ((npair, 4, 32, 32), f32), "wrow": ((npair,), f32), "extent": ((2,), torch.int32)}

    def _synthetic(self, seed):
        """frames in the layout the device front-end emits (ops.image_prep: NHWC with the stem's
        zero fourth channel) + the host-side head inputs of faster_rcnn_SGG_emb.py:170-245."""
        h, w, n_boxes, n_pairs = self.geom
        head = self.net.vrd
        im, info, annos = synthetic_sgg_batch(seed, self.n_frames, n_boxes, n_pairs, head.n_rel, head.n_obj, h, w)
        head.source_gt_rels = annos
        self.paths = sorted(annos, key=lambda s: int(s[1:]))
        fields = sgg_head_inputs([annos[p] for p in self.paths], info, head.n_rel)
        im4 = torch.zeros((im.shape[0], 4) + tuple(im.shape[2:]), device=self.dev).contiguous(memory_format=torch.channels_last)
        im4[:, :3] = torch.from_numpy(im).to(self.dev)
        return im4, info, fields

    def _frames(self, key):
        fs = self.shapes.get(key)
        if fs is None:
            fs = self.shapes[key] = _FrameSet(key, self.dev, key[0] if self.bb_split else 1, self.zero_arena)
        self._tick += 1
        fs.tick = self._tick
        return fs

    def _geom_dev(self, fs):
        if getattr(fs, "_geom", None) is None:
            fs._geom = torch.tensor([fs.fh, fs.fw], dtype=torch.int32, device=self.dev)
        return fs._geom

    def _grow(self, nb, npair):
        """Padded head inputs too small for this batch: new capacity (multiples of 16 rows), every graph is stale."""
        if self.tp:
            raise ValueError("SGGEmbStep: %d boxes / %d pairs exceed the capacity %d / %d; with fc6 cut across the ranks every "
                             "rank must hold the same number of rows -- construct the step with larger n_boxes / n_pairs"
                             % (nb, npair, self.cap_boxes, self.cap_pairs))
        up = lambda n, cap: max(cap, (n + 15) // 16 * 16)
        self.cap_boxes, self.cap_pairs = up(nb, self.cap_boxes), up(npair, self.cap_pairs)
        self.invalidate_graphs()

    def stage(self, frames, info, fields, size=None):
        """Hand the NEXT minibatch to the step.  ``frames``: (n,4,H,W) channels_last device tensor (the device front-end's
        blob), or (n,3,H,W) float frames on the device or the host (a roibatchLoader batch; pinned host memory crosses
        asynchronously), or -- with ``size`` = (n, H, W) -- a callable that writes the frames into the frame set it is given
        (``stage_batch_u8``).  ``info``: (n,3) im_info rows.  ``fields``: ``sgg_head_inputs`` of the batch.  Ordered on the
        caller's stream like everything else: it may be called right after ``__call__`` returns, the copies queue behind
        the step that is still running."""
        if fields is None:
            raise ValueError("SGGEmbStep.stage: a minibatch without an annotated relation (the reference loop skips it, "
                             "faster_rcnn_SGG_emb.py:177-183)")
        placer = None
        if callable(frames):
            if size is None:
                raise ValueError("SGGEmbStep.stage: a frame-writing callable needs size=(n, H, W)")
            placer = frames
            n, H, W = size
        else:
            n, _, H, W = frames.shape
        if int(n) != self.n_frames:
            raise ValueError("SGGEmbStep.stage: %d frames, the step was built for %d" % (n, self.n_frames))
        key = (int(n), int(H), int(W))
        fs = self._frames(key)
        if placer is not None:
            placer(fs)
        elif frames.shape[1] == 4 and frames.is_cuda:
            fs.im.copy_(frames)
        elif frames.is_cuda:
            fs.im[:, :3].copy_(frames)                           # NCHW3 -> NHWC4 (channel 3 stays zero): one strided copy
        else:                                                    # host frames: PCIe on the copy stream, beside the running step
            if self._uploader is None:
                self._uploader = _Uploader(self.dev)
            src, token = self._uploader.upload(frames)
            fs.im[:, :3].copy_(src)
            self._uploader.consumed(token)
        self.info = np.asarray(info, np.float32).reshape(-1, 3) if not torch.is_tensor(info) else info.detach().cpu().numpy().reshape(-1, 3)
        nb, npair = fields["boxes"].shape[0], fields["relb"].shape[0]
        self.n_rows = nb + npair
        host_masks = _rasterize_host(fields["bounds"])
        if not self.use_graph:
            # eager launches: exact sizes, nothing is padded
            lay = self._layout(nb, npair)
            if self.inp is None or not self.inp.same_layout(lay):
                self.cur = _Slot(lay, self.dev)
                self.mid_slot = _Slot(lay, self.dev)
                self.inp = _Slot(lay, self.dev, host=True)
                self._bind(nb, npair)
            if self.tp:
                parallel.assert_same_rows(self.n_rows, "boxes + pairs")
        else:
            if nb > self.cap_boxes or npair > self.cap_pairs:
                self._grow(nb, npair)
            lay = self._layout(self.cap_boxes, self.cap_pairs)
            if self.inp is None or not self.inp.same_layout(lay):
                old, old_caps = (self.cur if self.inp is not None else None), getattr(self, "_caps", None)
                old_mid = self.mid_slot if self.inp is not None else None
                self.cur = _Slot(lay, self.dev)
                self.mid_slot = _Slot(lay, self.dev)             # stage-split schedule: the batch between ``in`` and ``cur``
                self.inp = _Slot(lay, self.dev, host=True)
                self._bind(self.cap_boxes, self.cap_pairs)
                if old is not None and self._pipelined:          # the batches in flight move to the larger slots
                    ob, op = old_caps
                    for src, dst in ((old, self.cur), (old_mid, self.mid_slot)):
                        if src is None:
                            continue
                        o, n = src.views, dst.views
                        n["rois"][:ob].copy_(o["rois"][:ob]); n["rois"][self.cap_boxes:self.cap_boxes + op].copy_(o["rois"][ob:])
                        n["ix12"][:op].copy_(o["ix12"][:op]); n["ix12"][self.cap_pairs:self.cap_pairs + op].copy_(o["ix12"][op:])
                        for name in ("labels", "masks", "wrow"):
                            n[name][:op].copy_(o[name])
                        n["extent"].copy_(o["extent"])
        cb, cp = (nb, npair) if not self.use_graph else (self.cap_boxes, self.cap_pairs)
        rois = np.zeros((cb + cp, 5), np.float32)
        rois[:nb], rois[cb:cb + npair] = fields["boxes"], fields["relb"]
        ix12 = np.zeros((2 * cp,), np.int64)
        ix12[:npair], ix12[cp:cp + npair] = fields["ixs"], fields["ixo"]
        host = {"rois": rois, "labels": fields["labels"], "ix12": ix12, "masks": host_masks, "wrow": fields["wrow"],
                "extent": np.zeros((2,), np.int32)}                  # (h, w) of the C4 maps; filled by _measure() for a new size
        if fs.fh is not None:
            host["extent"][:] = (fs.fh, fs.fw)
        self.inp.write_host(host)
        self._pipe.staged()
        self._staged = key
        first = not self.primed and self.cur is not None and not self._pipelined
        if first and not getattr(self, "_filled", False):
            # the very first batch: the head's slot starts out holding it
            self.cur.buf.copy_(self.inp.buf)
            self._filled = True

    def _bind(self, nb, npair):
        """The head reads the ``cur`` slot: its fields, and the box / pair halves of the packed ones, as attributes."""
        v = self.cur.views
        self.rois, self.ix12, self.labels, self.masks, self.wrow = v["rois"], v["ix12"], v["labels"], v["masks"], v["wrow"]
        self.boxes, self.relb = v["rois"][:nb], v["rois"][nb:]
        self.ixs, self.ixo = v["ix12"][:npair], v["ix12"][npair:]
        self._caps = (nb, npair)
        self._filled = False

    def stage_batch(self, data):
        """One roibatchLoader(path_return=True) minibatch as the DataLoader collates it -- (im_data (n,3,H,W), im_info (n,3),
        gt_boxes, num_boxes, paths) -- looked up in ``vrd.source_gt_rels`` by the last path component as the reference loop
        does (trainval_net_SGG_emb.py:211-217).  Returns False for the batches the reference skips (a frame outside the
        aspect-ratio range, :209-210; no annotated relation in any frame, faster_rcnn_SGG_emb.py:177-183)."""
        if not isinstance(data, (list, tuple)) or len(data) <= 2:
            return False
        info = data[1].detach().cpu().numpy().reshape(-1, 3)
        rels = self.net.vrd.source_gt_rels
        fields = sgg_head_inputs([rels.get(str(p).split("/")[-1]) for p in data[4]], info, self.net.vrd.n_rel)
        if fields is None:
            return False
        self.stage(data[0], info, fields)
        return True

    def stage_batch_u8(self, data):
        """One ``roibatchLoader(device_prep=True, path_return=True)`` minibatch (``collate_device_prep``): uint8 frames as
        decoded + meta; the image work runs on the device (``_place_u8``), a quarter of the float blob's bytes cross PCIe.
        Same contract as ``stage_batch`` otherwise; a minibatch the device front-end does not take (the square trim) -> False."""
        if not isinstance(data, (list, tuple)) or len(data) <= 2:
            return False
        frames, meta = data[0], data[1].numpy()
        if int(meta[0][1]) <= 0 or int(meta[0][2]) <= 0:
            return False
        hc, wc = int(meta[0][1]), int(meta[0][2])
        info = np.array([[hc, wc, m[3]] for m in meta], np.float32)
        rels = self.net.vrd.source_gt_rels
        fields = sgg_head_inputs([rels.get(str(p).split("/")[-1]) for p in data[4]], info, self.net.vrd.n_rel)
        if fields is None:
            return False
        if self._uploader is None:
            self._uploader = _Uploader(self.dev)
        self.stage(lambda fs: _place_u8(self._uploader, frames, meta, fs.im), info, fields, size=(len(frames), hc, wc))
        return True

    def reseed(self, seed):
        """Stage the synthetic minibatch of ``seed`` (the data layer's job; resident before the timed region)."""
        self.stage(*self._synthetic(seed))

    # ------------------------------------------------------------------ feature-map buffers
    _channels = 1024

    def _reserve_cells(self, cells):
        if cells <= self.cap_cells:
            return
        n = self.n_frames * self._channels * cells
        new, new_h = torch.zeros(n, device=self.dev), torch.zeros(n, device=self.dev)
        if self.fmap_flat is not None:
            new[:self.fmap_flat.numel()].copy_(self.fmap_flat)
            new_h[:self.fmap_head_flat.numel()].copy_(self.fmap_head_flat)
        if self.stage_split:          # the front half's output (layer3[cut - 1]: the C4 map's shape) of this call and of the last
            m_next, m_cur = torch.zeros(n, device=self.dev), torch.zeros(n, device=self.dev)
            if self.mid_next_flat is not None:
                m_next[:self.mid_next_flat.numel()].copy_(self.mid_next_flat)
                m_cur[:self.mid_cur_flat.numel()].copy_(self.mid_cur_flat)
            self.mid_next_flat, self.mid_cur_flat = m_next, m_cur
        self.fmap_flat, self.fmap_head_flat, self.cap_cells = new, new_h, cells
        self.invalidate_graphs()                                 # they hold the old buffers' addresses

    def reserve(self, h, w):
        """Size the feature-map buffers for frames of up to (h, w) pixels (a loop that knows its largest minibatch avoids
        re-capturing when it arrives)."""
        self._reserve_cells(((h + 15) // 16 + 1) * ((w + 15) // 16 + 1))

    def _fmap_dst(self, fs, f=None):
        """Where the feature maps of frame set ``fs`` (all frames, or frame f) live in the packed buffer: the backbone's last
        layer writes there (``RCNN_base(..., out=)``: no copy afterwards)."""
        n, c = fs.key[0], self._channels
        dst = self.fmap_flat[:n * fs.fh * fs.fw * c].view(n, fs.fh, fs.fw, c).permute(0, 3, 1, 2)
        return dst if f is None else dst[f:f + 1]

    def _measure(self, fs):
        """First sight of a frame size: one eager pass tells the extent of its C4 map (and sizes the eager arena)."""
        if fs.fh is None:
            with self.ctx_bb, torch.no_grad():
                fm = self.net.RCNN_base(fs.im[:1])
            self._channels = int(fm.shape[1])
            fs.fh, fs.fw = int(fm.shape[2]), int(fm.shape[3])
            self._reserve_cells(fs.fh * fs.fw)
            if self._staged == fs.key and self.inp is not None:      # the batch staged for this size carries its extent
                g = self._geom_dev(fs)
                self.inp.views["extent"].copy_(g)
                if not self._pipelined:
                    self.cur.views["extent"].copy_(g)

    # ------------------------------------------------------------------ the two halves of a step
    def _rotate(self):
        """Sequential schedule: the head of this call works on the batch staged last."""
        if not self._pipelined:
            self.cur.buf.copy_(self.inp.buf)

    def _rotate_after_head(self):
        """Overlapped schedule: the head of call k reads ``cur`` = batch k-1 while the backbone branches work on the frames of
        batch k; once the head is done, batch k's head inputs (still in ``inp``: the next stage() is ordered behind this call)
        move to ``cur`` for call k+1.  ONE small copy per step."""
        if self._pipelined and self.stage_split:     # three slots: in (batch k) -> mid (k-1) -> cur (k-2, the head's)
            self.cur.buf.copy_(self.mid_slot.buf)
            self.mid_slot.buf.copy_(self.inp.buf)
        elif self._pipelined:
            self.cur.buf.copy_(self.inp.buf)

    def _mid_view(self, flat, fs):
        n, c = fs.key[0], self._channels
        return flat[:n * fs.fh * fs.fw * c].view(n, fs.fh, fs.fw, c).permute(0, 3, 1, 2)

    def _front(self, fs, ctx):
        """stem .. layer3[:cut] of the frames staged last, all frames in one pass -> ``mid_next``."""
        with ctx:
            with torch.no_grad():
                self.net.RCNN_base.forward_front(fs.im, self.cut, out=self._mid_view(self.mid_next_flat, fs))

    def _back(self, fs, ctx):
        """layer3[cut:] of the previous call's front output (``mid_cur``, a minibatch of size ``fs``) -> the packed feature maps."""
        with ctx:
            with torch.no_grad():
                self.net.RCNN_base.forward_back(self._mid_view(self.mid_cur_flat, fs), self.cut, out=self._fmap_dst(fs))

    def _backbone(self, fs):
        with self.ctx_bb:
            with torch.no_grad():
                self.net.RCNN_base(fs.im, out=self._fmap_dst(fs))          # static address across replays
        self._fmap_key = fs.key

    def _backbone_per_frame(self, fs, join=True):
        """Captured form with ``bb_split``: the frames of the minibatch are independent chains of ~100 short kernels each
        (a layer3 GEMM of one frame runs ~15 us, a quarter of it set-up, first-load latency and the store tail with the
        matrix pipe idle).  One graph branch per frame: the kernels of the two chains are co-resident on every CU, the
        fixed phases of one lie under the K loops of the other."""
        main = torch.cuda.current_stream(self.dev)
        n = fs.key[0]
        for f in range(n):
            st = self._frame_streams[f]
            with ops.branch(st, main):
                with fs.ctx[f]:
                    with torch.no_grad():
                        self.net.RCNN_base(fs.im[f:f + 1], out=self._fmap_dst(fs, f))
        if join:
            ops.join(main, *self._frame_streams[:n])
        self._fmap_key = fs.key

    TRACE_COLS = ("loss", "features", "scores", "embedding", "rng_canary", "fc7_weight", "fc6_weight_head", "boxes", "labels")

    def _head(self, fs=None):
        with self.ctx_head:
            if self.use_graph:        # packed maps, extent from the batch's slot: the launches are the same for every size
                src = self.fmap_head_flat if self._pipelined else self.fmap_flat
                fmap = ops.PackedMaps(src, self.n_frames, self._channels, self.cur.views["extent"])
            else:
                fmap = self.fmap
            score, x = self.net.vrd.forward_device(fmap, self.boxes, self.relb, self.masks, self.ixs, self.ixo,
                                                   rois=self.rois, ix12=self.ix12)
            loss = ops.bce_rows(score, self.labels, self.wrow)     # sum_r wrow[r] * mean_c BCE: one kernel each way
            if self.trace is not None:
                self._record(loss, fmap, score, x)
            self.opt.zero_grad()
            loss.backward(self._seed)                              # d(loss / world): a resident scalar, no division node
            self.loss.copy_(loss.detach())
            parallel.all_reduce_grads(self.opt.params())           # world > 1: RCCL, captured with the rest of the branch
            self.opt.step()
        self._rotate_after_head()

    @torch.no_grad()
    def _record(self, loss, fmap, score, x):
        """Diagnostic rows (``trace_rows``): per step, checksums of everything the head forward read and produced, written by
        the step itself (inside the graph) so that two trajectories can be compared step by step and tensor by tensor
        without putting anything but the graph launch on the caller's stream."""
        v = self.net.vrd
        d = lambda t: torch.linalg.vector_norm(t.detach().reshape(-1).float(), 2, dtype=torch.float64)    # accumulated in double
        feats = fmap.buf if isinstance(fmap, ops.PackedMaps) else fmap
        row = torch.stack([loss.detach().double(), d(feats), d(score), d(x), torch.rand((), device=self.dev).double(),
                           d(v.fc7.fc.weight), d(v.fc6.fc.weight.view(-1)[:1 << 22]), d(self.boxes), d(self.labels)])
        i = self._trace_i.clamp(max=self.trace.shape[0] - 1)
        self.trace.index_copy_(0, i, row.view(1, -1))
        self._trace_i.add_(1)

    def _body(self):
        """Sequential form (eager, and what the sequential graph captures): backbone, then head, of the staged batch."""
        fs = self.shapes[self._staged]
        self._measure(fs)
        self._rotate()
        self._backbone(fs)
        self._head()

    def _body_overlapped(self, fs):
        """What the overlapped graph of frame set ``fs`` captures.  Fork / join through the capturing stream: everything a
        side stream does lies between ``side.wait_stream(main)`` and ``main.wait_stream(side)``, i.e. inside the graph."""
        main = torch.cuda.current_stream(self.dev)
        self._rotate()
        self.fmap_head_flat.copy_(self.fmap_flat)   # hand-off: features of the batch now in ``cur`` (computed by the previous call)
        if self.stage_split:
            # Round 6.  [head of batch k-2] | [front half of batch k: stem .. layer3[:cut]] | [back half of batch k-1].  Three
            # chains as with one branch per frame, but every backbone kernel runs over BOTH frames: fuller launches, half as
            # many of them, the Winograd-domain filters read once per pair of frames.  The step is bound by the sum of its
            # kernels' alone-times with all chains ending together, and 2-frame kernels have less of it:
            # 4.40 -> 4.08-4.12 ms per step (tools/stage_split_probe.py).  The price is one more call of latency (``lag`` 2).
            fb = self.shapes[self._mid_key]         # the minibatch whose front half ran in the previous call
            self.mid_cur_flat.copy_(self.mid_next_flat)
            sa, sb = self._frame_streams[0], self._frame_streams[1]
            with ops.branch(sa, main):
                self._front(fs, fs.ctx_front)
            with ops.branch(sb, main):
                self._back(fb, fb.ctx_back)
            self._head()
            ops.join(main, sa, sb)
            return
        if self.bb_split:                   # one branch per frame, forked from the capturing stream itself (a fork inside a
            self._backbone_per_frame(fs, join=False)      # forked branch crashes hipStreamEndCapture on ROCm 7.2)
            self._head()
            ops.join(main, *self._frame_streams[:fs.key[0]])
            return
        with ops.branch(self._side, main):
            self._backbone(fs)              # batch k+1
        self._head()                        # batch k
        ops.join(main, self._side)

    def prime(self):
        """Overlapped schedule only: backbone pass of the batch staged first, so that the first call's head finds its
        features (``cur`` <- ``in`` as the end of a call would have done)."""
        if self.overlap and not self.primed:
            fs = self.shapes[self._staged]
            self._measure(fs)
            if self.stage_split:
                # the batch staged first fills EVERY stage of the pipeline: both backbone halves run here, all three slots hold
                # it.  A resident minibatch is then trained by every call from the first; a staging loop trains it in its first
                # call, runs no head in the second (``bubble``: the slot still holds the first batch) and is in step from the third
                self.cur.buf.copy_(self.inp.buf)
                self.mid_slot.buf.copy_(self.inp.buf)
                self._front(fs, self.ctx_bb)
                self.mid_cur_flat.copy_(self.mid_next_flat)
                self._back(fs, self.ctx_bb)
                self._mid_key = self._fmap_key = fs.key
                self._pipe.prime()
            else:
                self.cur.buf.copy_(self.inp.buf)
                self._backbone(fs)
        self.primed = True

    def _fill_call(self, fs):
        """A call while the stage pipeline fills: the graph's two backbone chains on eager launches, the slots advance, no head."""
        fb = self.shapes[self._mid_key]
        self.mid_cur_flat.copy_(self.mid_next_flat)
        self._back(fb, self.ctx_bb)
        self._front(fs, self.ctx_bb)
        self.cur.buf.copy_(self.mid_slot.buf)
        self.mid_slot.buf.copy_(self.inp.buf)
        self._fmap_key, self._mid_key = fb.key, fs.key

    # ------------------------------------------------------------------ capture
    def invalidate_graphs(self):
        """Drop every captured graph (a learning-rate change -- rates live in the captured kernel arguments --, a capacity
        change, new feature-map buffers).  They are captured again on first use."""
        dropped = any(fs.graph for fs in self.shapes.values())
        if dropped:
            torch.cuda.synchronize(self.dev)      # a replay may still be running: its executable graph goes only after it
        for fs in self.shapes.values():
            fs.graph = None
            fs.graphs.clear()
        if dropped:
            import gc
            gc.collect()
            torch.cuda.synchronize(self.dev)
        self._pool = None             # the allocator releases a pool with its last graph: the next capture opens a new one

    def capture(self, warmup=2, restore=False):
        """Warm up eagerly (sizes the arenas, fills the allocator), then capture the step for the staged frame size into ONE
        HIP graph (other sizes are captured when they are first staged).  Returns False (and keeps the eager form,
        ``graph_error`` says why) when graphs are off or capture fails.  ``restore``: the warm-up steps are real training
        steps on the staged batch; put parameters, momentum and the RNG state back afterwards (a training loop that must
        not see them).  ``warmup=0`` re-captures (after a learning-rate change: rates live in the captured kernel
        arguments)."""
        saved = None
        if restore and warmup:
            saved = self._snapshot()
        try:
            return self._capture(warmup)
        finally:
            if saved is not None:
                self._restore(saved)

    def _snapshot(self):
        pend, host = self.opt.pending_state()
        state = [it["p"].data for it in self.opt.items] + self.opt.state_tensors() + pend
        return (state, [t.clone() for t in state], torch.cuda.get_rng_state(self.dev), host)

    def _restore(self, saved):
        torch.cuda.synchronize(self.dev)
        with torch.no_grad():
            for t, sv in zip(saved[0], saved[1]):
                t.copy_(sv)
        torch.cuda.set_rng_state(saved[2], self.dev)
        self.opt.restore_pending(saved[3])
        self.opt.bump()

    def _capture(self, warmup):
        if self.tp:
            parallel.assert_same_rows(self.cap_boxes + self.cap_pairs if self.use_graph else self.n_rows, "boxes + pairs")
        if warmup == 0:
            self.invalidate_graphs()
        fs = self.shapes[self._staged]
        self._measure(fs)
        pipelined, self._pipelined = self._pipelined, False     # the warm-up steps are sequential eager steps
        s = ops.role_stream(self.dev, "warmup")
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            for i in range(warmup):
                self._body()
                if i == 0:
                    self.ctx_bb.fit()
                    self.ctx_head.fit()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        self._pipelined = pipelined
        if not self.use_graph:
            return False
        if self.overlap and not self._pipelined:
            self._pipelined = True
            self.prime()
            torch.cuda.synchronize(self.dev)
        ok = self._capture_frames(fs)
        if ok:
            self._graphs_on = True
        elif not self._graphs_on:
            self._pipelined = False      # nothing was ever captured: plain sequential eager steps (the documented fall-back)
            self.overlap = False
        return ok

    def _capture_frames(self, fs):
        """Capture the step whose backbone half works on frame set ``fs``.  Nothing here changes the training state: the
        per-frame arenas are sized by eager backbone passes into scratch outputs (the backbone is frozen), the head half is
        recorded, not run."""
        self._measure(fs)
        staged = self._pipelined and self.stage_split
        fb = self.shapes[self._mid_key] if staged else None       # the minibatch the back half of this graph works on
        if self._pipelined and self.bb_split and not fs.fitted:
            for _rep in range(2):
                for f, ctx in enumerate(fs.ctx):
                    with ctx:
                        with torch.no_grad():
                            self.net.RCNN_base(fs.im[f:f + 1])
                    ctx.fit()
            torch.cuda.synchronize(self.dev)
        if staged and not (fs.fitted_front and fb.fitted_back):
            # arenas of the two half contexts, sized by eager passes into SCRATCH outputs (mid_next / the feature maps hold live
            # pipeline state: the fitting passes must not write them; what they read may be stale)
            for _rep in range(2):
                if not fs.fitted_front:
                    with fs.ctx_front:
                        with torch.no_grad():
                            self.net.RCNN_base.forward_front(fs.im, self.cut)
                    fs.ctx_front.fit()
                if not fb.fitted_back:
                    with fb.ctx_back:
                        with torch.no_grad():
                            self.net.RCNN_base.forward_back(self._mid_view(self.mid_cur_flat, fb), self.cut)
                    fb.ctx_back.fit()
            fs.fitted_front = fb.fitted_back = True
            torch.cuda.synchronize(self.dev)
        fs.fitted = True
        if staged:
            live = [(f, k) for f in self.shapes.values() for k, g in f.graphs.items() if g]
            if len(live) >= self.max_graphs:                      # least recently used pair goes
                torch.cuda.synchronize(self.dev)
                f, k = min(live, key=lambda fk: fk[0].tick)
                f.graphs[k] = None
                if f.graph is not None and not any(f.graphs.values()):
                    f.graph = None
        else:
            live = [f for f in self.shapes.values() if f.graph]
            if len(live) >= self.max_graphs:                          # least recently used goes (after whatever is still running)
                torch.cuda.synchronize(self.dev)
                min(live, key=lambda f: f.tick).graph = None
        fmap_key = self._fmap_key
        try:
            g = torch.cuda.CUDAGraph()
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            torch.cuda.synchronize(self.dev)
            parallel.wait_for_collectives(self.dev)
            if self._pipelined:
                if self._side is None:
                    self._side = ops.role_stream(self.dev, "side")
                with torch.cuda.graph(g, pool=self._pool, **parallel.capture_kwargs()):
                    self._body_overlapped(fs)
            else:
                with torch.cuda.graph(g, pool=self._pool, **parallel.capture_kwargs()):
                    self._rotate()
                    self._backbone(fs)
                    self._head()
            fs.graph = g
            if staged:
                fs.graphs[fb.key] = g
            return True
        except Exception as e:      # report, fall back to eager launches for this size
            fs.graph = False
            if staged:
                fs.graphs[fb.key] = False
            self.graph_error = repr(e)
            ops.reset_branches()
            torch.cuda.synchronize(self.dev)
            return False
        finally:
            self._fmap_key = fmap_key        # recording is not running: the buffer still holds what it held

    def __call__(self):
        """One step on the caller's current stream (any stream, the legacy default stream included: see
        ``replay_graph``).  Returns the device scalar holding the loss of the batch the head just processed.
        Before a successful ``capture()`` a call is the sequential eager step."""
        fs = self.shapes[self._staged]
        try:
            if self._graphs_on and self._pipelined and self.stage_split:
                self._measure(fs)
                if self._pipe.call() is None:
                    self._fill_call(fs)              # the pipeline fills: both backbone halves, no head
                    self.n_bubbles += 1
                else:
                    fb_key = self._mid_key
                    if fs.graphs.get(fb_key) is None:
                        self._capture_frames(fs)     # first sight of this (front size, back size) pair
                    g = fs.graphs.get(fb_key)
                    self._tick += 1
                    fs.tick = self._tick
                    fs.graph = g if g else fs.graph
                    if g:
                        replay_graph(g, self.dev)
                    else:
                        self._body_overlapped(fs)    # this pair could not be captured: the same schedule on eager launches
                    self._fmap_key, self._mid_key = fb_key, fs.key
            elif self._graphs_on:
                if fs.graph is None:
                    self._capture_frames(fs)         # first sight of this frame size (or the graphs were invalidated)
                self._tick += 1
                fs.tick = self._tick
                if fs.graph:
                    replay_graph(fs.graph, self.dev)
                    self._fmap_key = fs.key
                elif self._pipelined:
                    self._body_overlapped(fs)        # this size could not be captured: the same schedule on eager launches
                else:
                    self._body()
            else:
                self._body()
        except BaseException:
            # an eager body that raised between a branch and its join leaves process-wide role streams marked open
            # (ops._FORKED): every later step object on this device would be refused its branches
            ops.reset_branches()
            raise
        self.opt.bump()
        return self.loss

    def flush(self):
        """Overlapped schedule: one more call without a new ``stage()`` -- the batches already in the pipeline advance one stage
        (``lag`` such calls end a loop: the last one runs the head of the batch staged last)."""
        return self()


def run_staged(step, stagers, keep):
    """Drive ``step`` (any schedule) over ``1 + len(stagers)`` minibatches: the first one is staged (and the step captured)
    already, ``stagers[k]()`` stages the next.  One minibatch is staged before every call while there are any; calls go on until
    the head has trained every minibatch; ``keep[j]`` (a device vector) receives the loss of minibatch j right behind its call
    (no host synchronisation).  Calls that run no head (``SGGEmbStep.bubble``) are not counted."""
    trained, nxt, total = 0, 0, len(stagers) + 1
    while trained < total:
        if nxt < len(stagers):
            stagers[nxt]()
            nxt += 1
        bubble = getattr(step, "bubble", False)
        loss = step()
        if not bubble:
            keep[trained].copy_(loss)
            trained += 1
    return keep


_REPLAY_STREAMS = {}
REDIRECT_DEFAULT_STREAM = True      # tools/graph_order_probe.py studies the runtime's default path and switches this off


def replay_graph(graph, device=None):
    """``graph.replay()`` on the caller's current stream -- except on the LEGACY DEFAULT stream, where the replay runs on a
    private stream between two event edges (the caller's stream order is kept).  ROCm 7.2's HIP runtime replays a graph through
    pre-built AQL packet batches (``DEBUG_CLR_GRAPH_PACKET_CAPTURE``, on by default); on the legacy default stream that path loses
    the order between a graph's nodes and the stream's other work while a second stream is busy (losses
    off by 2e-2 from the second step on, NaN weights; the same graphs are correct on any created stream).
    i2vsgg_amd/__init__.py switches the path off before the runtime initialises WHEN IT CAN -- but what the runtime actually
    read cannot be told from os.environ (``torch.cuda.is_available()`` / ``device_count()`` bring the runtime up without setting
    ``torch.cuda.is_initialized()``; a script may set the variable after its first HIP call), so the variable is not trusted
    as proof (round-3 advice): every replay asked for on the default stream is redirected.  Cost: two event edges per step."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    cur = torch.cuda.current_stream(dev)
    if not REDIRECT_DEFAULT_STREAM or cur.cuda_stream != torch.cuda.default_stream(dev).cuda_stream:
        graph.replay()
        return
    s = _REPLAY_STREAMS.get(dev.index)
    if s is None:
        s = _REPLAY_STREAMS[dev.index] = ops.role_stream(dev, "replay")
    s.wait_stream(cur)
    with torch.cuda.stream(s):
        graph.replay()
    cur.wait_stream(s)


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


class _DomainSet:
    """What the instance_styleD step owns for ONE pair of minibatch sizes (source (n,Hs,Ws), target (n,Ht,Wt)) and one
    ``im_info[0]`` (the anchor-target layer's inside test follows it, anchor_target_layer.py:85-86): the staged frames,
    the launch contexts and the captured graph."""

    def __init__(self, key, device, batched, branches):
        n, hs, ws, ht, wt = key[:5]
        # frames live in the layout the stem consumes (NHWC, zero fourth channel: ops.image_prep's blob): stage() does the
        # NCHW3 -> NHWC4 placement once, outside the step, instead of a fill + a copy per branch inside it
        z = lambda b, h, w: torch.zeros((b, 4, h, w), device=device).contiguous(memory_format=torch.channels_last)
        self.key = key
        self.im_s, self.im_t = z(n, hs, ws), z(n, ht, wt)
        self.im_st = z(2 * n, hs, ws) if (batched and (hs, ws) == (ht, wt)) else None
        # ordered=True (round 6): every split reduction of the detector step in a fixed order -- bit-reproducible steps at
        # +0.3 % (round 5: +4 %, most of it ONE unsplit 18-row filter gradient)
        self.ctx = ops.LaunchContext(device, ordered=True)
        self.ctx_src = ops.LaunchContext(device, ordered=True) if branches else None
        self.ctx_tgt = ops.LaunchContext(device, ordered=True) if branches else None
        self.graph = None             # None: not captured yet; False: capture failed (eager launches for this key)
        self.fitted = False
        self.tick = 0


class InstanceStyleDStep:
    """One D+G adversarial step (trainval_net_instance_styleD_bilinear.py:262-341): source forward with
    detection + RPN losses and 0.5*mean(d^2) for both discriminators, target forward with
    0.5*mean((1-d)^2), style terms weighted by style_lambda, ONE backward through the gradient-reversal
    layers, one SGD step over every trainable parameter.

    Eager by default with the target layers sampling on the host from the reference's np.random stream (bit parity with
    the reference's RNG contract; two small D2H copies per step).  ``capture()`` switches the target layers to
    device-side sampling (same subsample sizes, torch's generator) and records the whole step -- both forwards, the
    backward, the gradient exchange and the update -- into ONE HIP graph.

    The two data loaders of the loop pad every minibatch to its own aspect ratio (roibatchLoader.py:162-190), so source
    and target frames come in several sizes.  ``stage()`` takes any; a graph is captured the first time a (source size,
    target size) pair is met -- after one eager step at that size that sizes the arenas and fills the host-built caches
    (anchor grids), whose effect on parameters, momentum and the RNG stream is undone -- and replayed from then on; at most
    ``max_graphs`` are kept, in one shared memory pool (no two of them ever run at the same time)."""

    def __init__(self, net, n_frames, lr=5e-4, eta=0.1, eta_style=0.001, style_lambda=1.0, seed=3, device="cuda:0",
                 h=600, w=1000, n_gt=8, cr=False, max_graphs=32, stage_synthetic=True, optimizer="sgd"):
        self.net, self.dev = net, torch.device(device)
        self.cr = cr                  # --cr: consistency regularisation between instance- and image-level D (:299-312)
        self.world = parallel.world_size()
        self.eta, self.eta_style, self.style_lambda = eta, eta_style, style_lambda
        self.geom = (h, w)
        self.n_frames, self.n_gt = n_frames, n_gt
        import os
        self.batched = True
        # filter gradients of the bottleneck nodes on a side branch of the step (ops.WGRAD_STREAM)
        # (only with the one-pass backbone: a filter met twice in one backward would have its two gradients added on the main
        # stream while the side branch may still be writing the first)
        self.wgrad_branch = False      # filter gradients on a side branch: -1 % of the step for +2.4 GB; a test sets it
        self._wgrad_stream = ops.role_stream(self.dev, "wgrad") if self.wgrad_branch else None
        # the captured step: source and target as two branches of the graph (_body_branches)
        self.branches = os.environ.get("I2V_ISD_BRANCHES", "1") != "0" and self.dev.type == "cuda" and not self.wgrad_branch
        self._branch_streams = [ops.role_stream(self.dev, ("domain", i)) for i in range(2)] if self.branches else []
        self.sets, self.max_graphs, self._tick, self._pool = {}, int(max_graphs), 0, None
        self._cur = None              # the _DomainSet staged last
        self._uploader = None
        self._graphs_on = False
        mg = int(cfg.MAX_NUM_GT_BOXES)
        self.info = torch.zeros((n_frames, 3), device=self.dev)        # shared by every size: a captured step reads them
        self.info_t = torch.zeros((n_frames, 3), device=self.dev)
        self.gt = torch.zeros((n_frames, mg, 5), device=self.dev)
        self.nb = torch.zeros((n_frames,), device=self.dev)
        self.gt_t = torch.zeros((n_frames, 1, 5), device=self.dev)
        self.nb_t = torch.zeros((n_frames,), device=self.dev)
        if stage_synthetic:           # a loop fed by data loaders stages its own first minibatch (``stage_batch``)
            self.reseed(seed)
        self.opt = make_optimizer(optimizer, list(net.named_parameters()), lr)
        # the gradient exchange in buckets (parallel.exchange_in_buckets; only when an exchange exists): bucket of every
        # parameter, and a hook per parameter that marks, on the stream its gradient was produced on, where its bucket stands
        ids, _ = parallel.detector_buckets([it["name"] for it in self.opt.items])
        used = sorted(set(ids))
        self._buckets = [[i for i, b in enumerate(ids) if b == k] for k in used]
        self._marks = None            # the parallel.BucketMarks of the branch whose backward is being queued
        for i, p in enumerate(self.opt.params()):
            p.register_hook(lambda g, k=used.index(ids[i]): self._bucket_hit(k))
        # total, det = the four detection losses summed, the four discriminator terms (trainval_net_instance...:276-296), then the
        # four detection losses one by one (rpn_loss_cls, rpn_loss_box, RCNN_loss_cls, RCNN_loss_bbox: :276-279)
        self.names = ["total", "det", "dloss_s", "dloss_t", "dloss_s_style", "dloss_t_style"] + \
            (["source_adv_cst", "target_adv_cst"] if cr else []) + ["rpn_cls", "rpn_box", "rcnn_cls", "rcnn_box"]
        self._loss_buf = torch.zeros((len(self.names),), device=self.dev)           # static addresses: a captured step writes here
        self.losses = {k: self._loss_buf[i] for i, k in enumerate(self.names)}      # (views: one stack + one copy per step)
        self.graph_error = None

    # ------------------------------------------------------------------ compatibility views
    @property
    def graph(self):
        return (self._cur.graph or None) if self._cur is not None else None

    im_s = property(lambda self: self._cur.im_s)
    im_t = property(lambda self: self._cur.im_t)
    im_st = property(lambda self: self._cur.im_st)
    ctx = property(lambda self: self._cur.ctx)
    ctx_src = property(lambda self: self._cur.ctx_src)
    ctx_tgt = property(lambda self: self._cur.ctx_tgt)

    # ------------------------------------------------------------------ data side
    def stage(self, im_s, info, gt, nb, im_t, info_t=None, placers=None):
        """Hand the next minibatch to the step: source frames (N,3,H,W) with ``im_info`` (N,3), ``gt_boxes``
        (N,MAX_NUM_GT_BOXES,5) and ``num_boxes`` (N,) -- one roi_data_layer batch -- and N target frames with their
        ``im_info`` (default: the source's).  Host tensors cross PCIe here (pinned ones asynchronously); everything is
        copied into static device tensors on the caller's stream (a captured step is bound to their addresses)."""
        n, _, hs, ws = im_s.shape
        ht, wt = int(im_t.shape[2]), int(im_t.shape[3])
        if int(n) != self.n_frames or int(im_t.shape[0]) != self.n_frames:
            raise ValueError("InstanceStyleDStep.stage: %d source / %d target frames, the step was built for %d"
                             % (n, im_t.shape[0], self.n_frames))
        if tuple(gt.shape[1:]) != tuple(self.gt.shape[1:]):
            raise ValueError("InstanceStyleDStep.stage: gt_boxes must be (N, MAX_NUM_GT_BOXES = %d, 5)" % self.gt.shape[1])
        info_t = info if info_t is None else info_t
        info0 = info[0].tolist()                    # a loader's im_info is a host tensor: no synchronisation
        key = (int(n), int(hs), int(ws), ht, wt, int(info0[0]), int(info0[1]))
        ds = self.sets.get(key)
        if ds is None:
            ds = self.sets[key] = _DomainSet(key, self.dev, self.batched, self.branches)
        self._tick += 1
        ds.tick = self._tick
        self._cur = ds
        self.info0 = (int(info0[0]), int(info0[1]))
        cp = lambda dst, src: dst.copy_(src, non_blocking=True)
        for k, (dst, src) in enumerate(((ds.im_s, im_s), (ds.im_t, im_t))):
            if placers is not None:                              # the device front-end writes the frames itself (_place_u8)
                placers[k](dst)
            elif src.is_cuda:
                dst[:, :3].copy_(src)
            else:                                                # host frames: PCIe on the copy stream, beside the running step
                if self._uploader is None:
                    self._uploader = _Uploader(self.dev)
                dev3, token = self._uploader.upload(src.float() if src.dtype != torch.float32 else src)
                dst[:, :3].copy_(dev3)
                self._uploader.consumed(token)
        cp(self.info, info); cp(self.info_t, info_t); cp(self.gt, gt); cp(self.nb, nb)
        if ds.im_st is not None:
            ds.im_st[:n].copy_(ds.im_s); ds.im_st[n:].copy_(ds.im_t)

    def stage_batch_u8(self, data_s, data_t):
        """The same with ``roibatchLoader(device_prep=True)`` minibatches (``collate_device_prep``): uint8 frames, the image work
        on the device.  False when either minibatch cannot go that way (the square trim) or would be skipped."""
        ok = lambda d: isinstance(d, (list, tuple)) and len(d) >= 4 and int(d[1][0][1]) > 0 and int(d[1][0][2]) > 0
        if not ok(data_s) or not ok(data_t):
            return False
        ms, mt = data_s[1].numpy(), data_t[1].numpy()
        info = torch.tensor([[m[1], m[2], m[3]] for m in ms], dtype=torch.float32)
        info_t = torch.tensor([[m[1], m[2], m[3]] for m in mt], dtype=torch.float32)
        n = len(data_s[0])
        key_shapes = (torch.empty((n, 0, int(ms[0][1]), int(ms[0][2]))), torch.empty((n, 0, int(mt[0][1]), int(mt[0][2]))))
        if self._uploader is None:
            self._uploader = _Uploader(self.dev)
        self.stage(key_shapes[0], info, data_s[2], data_s[3], key_shapes[1], info_t,
                   placers=(lambda dst: _place_u8(self._uploader, data_s[0], ms, dst),
                            lambda dst: _place_u8(self._uploader, data_t[0], mt, dst)))
        return True

    def stage_batch(self, data_s, data_t):
        """One minibatch of each of the loop's two ``roibatchLoader`` iterators as the DataLoader collates them
        (trainval_net_instance_styleD_bilinear.py:236-291): source (im_data, im_info, gt_boxes, num_boxes), target
        (im_data, im_info, ...) whose boxes are not used.  Returns False for the batches the reference skips (:241-252)."""
        if not isinstance(data_s, (list, tuple)) or not isinstance(data_t, (list, tuple)) or len(data_s) < 4 or len(data_t) < 2:
            return False
        self.stage(data_s[0], data_s[1], data_s[2], data_s[3], data_t[0], data_t[1])
        return True

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
            batched = self.batched and self.im_st is not None      # one pass needs source and target frames of one size
            if batched:
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
            parts = dict(rpn_cls=_mean1(l_rpn_cls), rpn_box=_mean1(l_rpn_box), rcnn_cls=_mean1(l_cls), rcnn_box=_mean1(l_box))
            loss = parts["rpn_cls"] + parts["rpn_box"] + parts["rcnn_cls"] + parts["rcnn_box"]
            dloss_s = ops.half_mse(d_inst)                  # 0.5 * mean(d^2) (:276-277), one kernel each way
            dloss_s_style = ops.half_mse(d_style)
            if batched:
                d_inst_t, d_style_t = net.forward_features(ft, f1t, self.info_t, self.gt_t, self.nb_t, True, self.eta,
                                                           self.eta_style)
            else:
                d_inst_t, d_style_t = net(self.im_t, self.info_t, self.gt_t, self.nb_t, target=True, eta=self.eta,
                                          eta_style=self.eta_style)
            dloss_t = ops.half_mse(d_inst_t, 1.0)           # 0.5 * mean((1 - d)^2) (:294-295)
            dloss_t_style = ops.half_mse(d_style_t, 1.0)
            total = loss + dloss_s + dloss_t + self.style_lambda * (dloss_s_style + dloss_t_style)
            vals = dict(det=loss, dloss_s=dloss_s, dloss_t=dloss_t, dloss_s_style=dloss_s_style, dloss_t_style=dloss_t_style, **parts)
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
            self._loss_buf.copy_(torch.stack([vals[k].detach().reshape(()) for k in self.names]))

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
        # Each branch's autograd graph is gone before the other branch's forward starts (only detached values leave the
        # block): a parameter's AccumulateGrad node lives as long as a graph references it and remembers the stream it was
        # made on -- shared between the two branches the engine would synchronise their streams with each other.
        def source():
            out = net(self.im_s, self.info, self.gt, self.nb, target=False, eta=self.eta, eta_style=self.eta_style)
            _, _, _, l_rpn_cls, l_rpn_box, l_cls, l_box, _, d_inst, d_style = out
            v = dict(rpn_cls=_mean1(l_rpn_cls), rpn_box=_mean1(l_rpn_box), rcnn_cls=_mean1(l_cls), rcnn_box=_mean1(l_box))
            v.update({"det": v["rpn_cls"] + v["rpn_box"] + v["rcnn_cls"] + v["rcnn_box"],
                      "dloss_s": ops.half_mse(d_inst), "dloss_s_style": ops.half_mse(d_style)})
            part = v["det"] + v["dloss_s"] + self.style_lambda * v["dloss_s_style"]
            if self.cr:
                v["source_adv_cst"] = _consistency_term(d_inst, d_style)
                part = part + v["source_adv_cst"]
            g = torch.autograd.grad(part / self.world, params, allow_unused=True)
            v["_src"] = part
            net.RCNN_rpn.rpn_loss_cls = net.RCNN_rpn.rpn_loss_box = 0      # the module keeps its last losses (rpn.py:89-108): they hold the graph
            return {k: t.detach() for k, t in v.items()}, g

        def target():
            d_inst_t, d_style_t = net(self.im_t, self.info_t, self.gt_t, self.nb_t, target=True, eta=self.eta,
                                      eta_style=self.eta_style)
            v = {"dloss_t": ops.half_mse(d_inst_t, 1.0), "dloss_t_style": ops.half_mse(d_style_t, 1.0)}
            part = v["dloss_t"] + self.style_lambda * v["dloss_t_style"]
            if self.cr:
                v["target_adv_cst"] = _consistency_term(d_inst_t, d_style_t)
                part = part + v["target_adv_cst"]
            g = torch.autograd.grad(part / self.world, params, allow_unused=True)
            v["_tgt"] = part
            return {k: t.detach() for k, t in v.items()}, g

        exchange = parallel.exchange_enabled()
        marks = {}
        try:
            with ops.branch(s_src, main), self.ctx_src:
                self._marks = marks["s"] = parallel.BucketMarks(True) if exchange else None
                v, grads["s"] = source()
                vals.update(v)
            with ops.branch(s_tgt, main), self.ctx_tgt:
                self._marks = marks["t"] = parallel.BucketMarks(True) if exchange else None
                v, grads["t"] = target()
                vals.update(v)
        finally:
            self._marks = None
        if exchange:
            # Round 6: the exchange runs on the CAPTURING stream, which has nothing else to do between the fork of the two domain
            # branches and their join.  Bucket k (backward order: heads + layer4 + RPN, layer3 in three slices, the early layers)
            # is summed over the two domains and all-reduced as soon as BOTH branches have queued its last gradient -- an event
            # edge from each -- so the 80 MB of bucket 0 cross xGMI beside the backward of layer3, each layer3 slice beside the
            # next, and only the last bucket (layer2 / layer1 / netD_style, ~10 MB) is exposed.  Rounds 1-5: one all-reduce of
            # all 202 MB after the join.  (Not a third forked branch: RCCL runs a collective on a stream of its own forked from
            # the issuing one, and a fork inside a forked branch ends hipStreamEndCapture in a host segfault -- DESIGN.md 5.1;
            # the first form of this schedule died exactly so.)
            tokens = parallel.exchange_in_buckets(params, self._buckets, [grads["s"], grads["t"]], (marks["s"], marks["t"]), main)
            parallel.finish_buckets(tokens)
            ops.join(main, s_src, s_tgt)
        else:
            ops.join(main, s_src, s_tgt)
            both = [(a, b) for a, b in zip(grads["s"], grads["t"]) if a is not None and b is not None]
            if both:
                torch._foreach_add_([a for a, _ in both], [b for _, b in both])
            for p, a, b in zip(params, grads["s"], grads["t"]):
                p.grad = a if a is not None else b
        vals["total"] = vals.pop("_src") + vals.pop("_tgt")
        self.opt.step()
        self._loss_buf.copy_(torch.stack([vals[k].detach().reshape(()) for k in self.names]))

    def _bucket_hit(self, k):
        m = self._marks
        if m is not None:
            m.hit(k)

    def _device_sampling(self, on):
        atl = self.net.RCNN_rpn.RPN_anchor_target
        atl.device_sampling = on
        # the inside-anchor test follows im_info[0] (anchor_target_layer.py:85-86); its value is known on the host from the
        # staged batch, so the layer needs no device read -- and a captured step is keyed by it (``_DomainSet``)
        atl.image_size = self.info0 if on else None
        self.net.RCNN_proposal_target.device_sampling = on

    def _snapshot(self):
        pend, host = self.opt.pending_state()
        state = [it["p"].data for it in self.opt.items] + self.opt.state_tensors() + pend
        return (state, [t.clone() for t in state], torch.cuda.get_rng_state(self.dev), host)

    def _restore(self, saved):
        torch.cuda.synchronize(self.dev)
        with torch.no_grad():
            for t, sv in zip(saved[0], saved[1]):
                t.copy_(sv)
        torch.cuda.set_rng_state(saved[2], self.dev)
        self.opt.restore_pending(saved[3])
        self.opt.bump()

    def invalidate_graphs(self):
        """Drop every captured graph (a learning-rate change: the rates live in the captured kernel arguments); they are
        captured again on first use, into the same memory pool."""
        dropped = any(ds.graph for ds in self.sets.values())
        if dropped:
            torch.cuda.synchronize(self.dev)      # a replay may still be running: its executable graph goes only after it
        for ds in self.sets.values():
            ds.graph = None
        if dropped:
            import gc
            gc.collect()
            torch.cuda.synchronize(self.dev)
        self._pool = None             # the allocator releases a pool with its last graph: the next capture opens a new one

    def capture(self, warmup=2, restore=False):
        """Device-side target sampling, eager warm-up, then the whole step for the staged sizes as ONE HIP graph (other
        sizes are captured when they are first staged).  False (eager form kept, host-side sampling restored, ``graph_error``
        says why) when capture fails.  ``restore``: the warm-up steps are real training steps on the staged batch; put
        parameters, momentum and the RNG state back afterwards.  ``warmup=0`` re-captures (after a learning-rate change:
        the rates live in the captured kernel arguments)."""
        if warmup == 0:
            self.invalidate_graphs()
        ok = self._capture_set(self._cur, warmup, restore)
        if ok:
            self._graphs_on = True
        elif not self._graphs_on:
            self._device_sampling(False)         # the eager form keeps the np.random contract it documents
        return ok

    def _capture_set(self, ds, warmup, restore):
        saved = self._snapshot() if (restore and warmup) else None
        try:
            self._device_sampling(True)
            body = self._body_branches if self.branches else self._body
            s = ops.role_stream(self.dev, "warmup")
            s.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(s):
                for i in range(warmup):
                    body()
                    if i == 0 and not ds.fitted:
                        for ctx in (ds.ctx, ds.ctx_src, ds.ctx_tgt):
                            if ctx is not None:
                                ctx.fit()
                        ds.fitted = True
            torch.cuda.current_stream(self.dev).wait_stream(s)
            torch.cuda.synchronize(self.dev)
        finally:
            if saved is not None:
                self._restore(saved)
        live = [d for d in self.sets.values() if d.graph]
        if len(live) >= self.max_graphs:                          # least recently used goes (after whatever is still running)
            torch.cuda.synchronize(self.dev)
            min(live, key=lambda d: d.tick).graph = None
        try:
            g = torch.cuda.CUDAGraph()
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            parallel.wait_for_collectives(self.dev)
            with torch.cuda.graph(g, pool=self._pool, **parallel.capture_kwargs()):
                body()
            ds.graph = g
            return True
        except Exception as e:
            ds.graph = False
            self.graph_error = repr(e)
            ops.reset_branches()
            torch.cuda.synchronize(self.dev)
            return False

    def __call__(self):
        ds = self._cur
        try:
            if self._graphs_on:
                if ds.graph is None:
                    # first sight of this pair of sizes (or the graphs were invalidated): one eager step sizes its arenas and
                    # fills the host-built caches, its effect on the training state is undone, then the step is recorded
                    self._capture_set(ds, 0 if ds.fitted else 1, True)
                if ds.graph:
                    replay_graph(ds.graph, self.dev)
                else:
                    self._device_sampling(True)
                    (self._body_branches if self.branches else self._body)()
                self.opt.bump()
            else:
                self._body()
                if not ds.fitted:           # eager use: size the arena of atomically accumulated outputs after the first step
                    ds.ctx.fit()
                    ds.fitted = True
        except BaseException:
            ops.reset_branches()            # a body that raised between a branch and its join: see SGGEmbStep.__call__
            raise
        return self.losses["total"]


def _mean1(t):
    """``t.mean()`` of the loop (trainval_net_instance_styleD_bilinear.py:276-279) without a reduction kernel when t holds one
    element (the detector returns its four losses as 1-element tensors)."""
    return t.reshape(()) if t.numel() == 1 else t.mean()


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


def build_sgg_net(layers=101, n_rel=62, n_cls=16, seed=0, device="cuda:0", emb_dim=300, use_obj_visual=True, spatial_type=2):
    """Random-init SGG_emb model of the reference architecture (no checkpoint is reachable).  ``emb_dim``,
    ``use_obj_visual``, ``spatial_type``: the reference's flags (parser_func.py:155-163,182)."""
    import argparse
    from .model.faster_rcnn.resnet_SGG_emb import resnet
    torch.manual_seed(seed)
    args = argparse.Namespace(num_relations=n_rel, num_classes=n_cls, emb_dim=int(emb_dim), use_obj_visual=bool(use_obj_visual),
                              spatial_type=int(spatial_type), vrd_task="pre_det")
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
