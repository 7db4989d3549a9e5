"""Per-frame evaluation helpers: what the reference's test scripts do between the model forward and the pickle.

``detect_frame``  test_net_instance_styleD_bilinear.py:140-221 -- eval forward, then the per-class detection
                  post-processing as ONE device pass (``ops.detection_postprocess``; the reference makes
                  n_classes - 1 host NMS calls per frame) and a single D2H copy of the surviving boxes.
``detection_output``  lib/utils.py:584-628 -- top-100 relation triplets of a frame from the ``vrd_data`` dict the
                  eval branch of ``forward_relation`` returns; scaling by the box confidences, the ranking over the
                  (pair, predicate) grid and the top-k cut run on the device (``ops.relation_topk``).
``relation_frame``  backbone forward + ``forward_relation_eval`` + ``detection_output`` for one frame.
``DetectStep``    ``detect_frame`` for several frames at a time as one replayable HIP graph (a branch per frame).
``RelationStep``  ``relation_frame`` + ``detection_output`` the same way.
"""
import numpy as np
import torch

from . import ops
from .model.utils.config import cfg


class _FrameShape:
    """What a frame-graph step owns for one (frames, H, W): the staged frames, a launch context per frame branch, the graph."""

    def __init__(self, key, device):
        n, h, w = key
        self.key = key
        self.im = torch.zeros((n, 4, h, w), device=device).contiguous(memory_format=torch.channels_last)
        self.ctx = [ops.LaunchContext(device) for _ in range(n)]
        self.graph = None             # None: not captured yet; False: capture failed (eager launches for this size)
        self.tick = 0


class _FrameGraphStep:
    """Shared machinery of ``DetectStep`` / ``RelationStep``: ``frames`` independent per-frame bodies (``_frame(fs, f)``, written by
    the subclass against static device buffers) as one HIP graph per frame size with a branch per frame; at most ``max_graphs``
    sizes are kept (least recently used goes first), their graphs share one memory pool (no two ever run at once)."""

    def __init__(self, net, frames, device, use_graph, max_graphs):
        if net.training:
            raise RuntimeError("%s: the network must be in eval mode (test_net_...:131 fasterRCNN.eval())" % type(self).__name__)
        self.net, self.dev, self.frames = net, torch.device(device), int(frames)
        self.use_graph, self.max_graphs = bool(use_graph), max(int(max_graphs), 1)
        self._streams = [ops.role_stream(self.dev, ("frame", f)) for f in range(self.frames)]
        self.shapes, self._pool, self._tick, self._staged = {}, None, 0, None
        self.graph_error = None
        self._slot = 0

    @torch.no_grad()
    def _body(self, fs):
        main = torch.cuda.current_stream(self.dev)
        for f, st in enumerate(self._streams):     # fork from the launching stream itself (a fork inside a fork breaks capture)
            with ops.branch(st, main):
                self._frame(fs, f)
        ops.join(main, *self._streams)

    def _shape(self, H, W):
        key = (self.frames, int(H), int(W))
        fs = self.shapes.get(key)
        if fs is None:
            if len(self.shapes) >= self.max_graphs:           # a new size: the least recently used one goes, once it is idle
                torch.cuda.synchronize(self.dev)
                del self.shapes[min(self.shapes, key=lambda k: self.shapes[k].tick)]
            fs = self.shapes[key] = _FrameShape(key, self.dev)
        self._staged = key
        return fs

    def _place(self, fs, im_data):
        n, c = im_data.shape[:2]
        if c == 4:
            fs.im[:n].copy_(im_data, non_blocking=True)
        else:
            fs.im[:n, :3].copy_(im_data, non_blocking=True)      # NCHW3 -> NHWC4 (channel 3 stays zero)

    def _place_u8(self, frames_u8, meta):
        """The device front-end of ``roibatchLoader(training=False, device_prep=True)`` items: uint8 frames as decoded cross PCIe,
        BGR swap / mean subtraction / resize run in ``i2v_image_prep``.  ``meta`` rows: [flipped, canvas_h, canvas_w, scale,
        target]; the frames of a call share their canvas.  -> (frame set, the (n,3) im_info rows)."""
        from .train import _Uploader, _place_u8
        meta = np.asarray(meta.cpu() if torch.is_tensor(meta) else meta, np.float64).reshape(-1, 5)
        n = len(frames_u8)
        if n > self.frames or len(meta) != n:
            raise ValueError("%s.stage_u8: %d frames / %d meta rows, the step was built for %d" % (type(self).__name__, n, len(meta), self.frames))
        sizes = {(int(m[1]), int(m[2])) for m in meta}
        if len(sizes) != 1:
            raise ValueError("stage_u8: the frames of one call must share their resized size, got %s" % sorted(sizes))
        (H, W), = sizes
        fs = self._shape(H, W)
        if getattr(self, "_uploader", None) is None:
            self._uploader = _Uploader(self.dev)
        frames_u8 = [f.reshape(f.shape[-3:]) for f in frames_u8]              # (1,H,W,3) items of a batch_size-1 loader
        _place_u8(self._uploader, frames_u8, meta, fs.im[:n])
        return fs, np.array([[m[1], m[2], m[3]] for m in meta], np.float32)

    def invalidate_graphs(self):
        """Drop every captured graph (they are captured again on first use).  Needed after the network's weights change: a
        graph holds the Winograd-domain filters of the weights it was captured with."""
        if any(fs.graph for fs in self.shapes.values()):
            torch.cuda.synchronize(self.dev)
        for fs in self.shapes.values():
            fs.graph = None
        self._pool = None             # the allocator releases a pool with its last graph

    def _capture(self, fs):
        for _ in range(2):                         # eager passes: anchors / transformed filters cached, arenas sized
            for f in range(self.frames):
                with torch.no_grad():
                    self._frame(fs, f)
                fs.ctx[f].fit()
        torch.cuda.synchronize(self.dev)
        if not self.use_graph:
            fs.graph = False
            return
        try:
            g = torch.cuda.CUDAGraph()
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(g, pool=self._pool):
                self._body(fs)
            fs.graph = g
        except Exception as e:                     # report, keep the eager form for this size
            fs.graph, self.graph_error = False, repr(e)
            ops.reset_branches()
            torch.cuda.synchronize(self.dev)

    def _run_staged(self):
        fs = self.shapes[self._staged]
        if fs.graph is None:
            self._capture(fs)
        self._tick += 1
        fs.tick = self._tick
        if fs.graph:
            from .train import replay_graph
            replay_graph(fs.graph, self.dev)
        else:
            self._body(fs)

    def run(self, batches, u8=False):
        """Generator over batches (the argument tuples of ``stage``, or of ``stage_u8`` with ``u8``): yields each batch's result
        list while the next batch runs.  (Results sit in two alternating pinned buffers: collect a token before launching
        twice more.)"""
        pending = None
        for b in batches:
            (self.stage_u8 if u8 else self.stage)(*b)
            token = self.launch()
            if pending is not None:
                yield self.collect(pending)
            pending = token
        if pending is not None:
            yield self.collect(pending)

    def __call__(self, *batch):
        self.stage(*batch)
        return self.collect(self.launch())


class DetectStep(_FrameGraphStep):
    """The per-frame body of test_net_instance_styleD_bilinear.py:133-221 (eval forward, de-normalise / decode / clip, per class
    threshold + sort + NMS 0.3, top-``max_per_image``) for ``frames`` frames at a time as ONE replayable HIP graph with a branch
    per frame.  The reference evaluates frame by frame (batch_size 1, :95); one frame's kernels leave much of the chip idle
    (a layer3 GEMM of one frame is 480 workgroups for 1024 slots), so independent frames side by side are the cheap speed-up --
    every frame is still processed alone: same launches, same arithmetic, same results as ``detect_frame`` (up to the fp32
    atomics of the split-K GEMMs, as between two calls of ``detect_frame``).

    ``step(im_data, im_info)`` -> list over the frames of the reference's ``all_boxes[j][i]`` lists; ``step.run(batches)``
    keeps one batch in flight while the host unpacks the previous one.  Frames of one call share a size (the loader pads a
    minibatch to one size).  The frame's ``im_info`` row is read on the device (``i2v_det_postprocess_info``), so frames of
    any scale replay the same graph."""

    def __init__(self, net, frames=2, thresh=0.0, max_per_image=100, class_agnostic=False, device="cuda:0", use_graph=True,
                 max_graphs=8):
        super().__init__(net, frames, device, use_graph, max_graphs)
        self.thresh, self.max_per_image, self.class_agnostic = float(thresh), int(max_per_image), bool(class_agnostic)
        self.R, self.C = int(cfg.TEST.RPN_POST_NMS_TOP_N), int(net.n_classes)
        self.info = torch.zeros((self.frames, 3), device=self.dev)
        self.dets = torch.zeros((self.frames, self.C, self.R, 5), device=self.dev)
        self.counts = torch.zeros((self.frames, self.C), device=self.dev, dtype=torch.int32)
        self._host = [(torch.zeros(self.dets.shape).pin_memory(), torch.zeros(self.counts.shape, dtype=torch.int32).pin_memory(),
                       torch.cuda.Event()) for _ in range(2)]

    def _frame(self, fs, f):
        """One frame, as the reference's loop body runs it."""
        with fs.ctx[f]:
            rois, cls_prob, bbox_pred = self.net.forward_detect(fs.im[f:f + 1], self.info[f:f + 1])
            stds = means = None
            agnostic = self.class_agnostic
            if cfg.TEST.BBOX_REG and cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED:
                stds, means = cfg.TRAIN.BBOX_NORMALIZE_STDS, cfg.TRAIN.BBOX_NORMALIZE_MEANS
            if not cfg.TEST.BBOX_REG:              # :167-169 "simply repeat the boxes"
                bbox_pred, agnostic, stds, means = torch.zeros((self.R, 4), device=self.dev), True, None, None
            ops.detection_postprocess(rois[0], cls_prob[0], bbox_pred.reshape(self.R, -1), 0.0, 0.0, 0.0, agnostic, stds, means,
                                      self.thresh, cfg.TEST.NMS, self.max_per_image, im_info=self.info[f],
                                      out=(self.dets[f], self.counts[f]))

    def stage(self, im_data, im_info):
        """``im_data``: (n,3,H,W) float frames (device or pinned host) or the (n,4,H,W) channels_last blob of the device
        front-end, n <= frames (a short last batch: the other branches re-run what they hold); ``im_info``: (n,3)."""
        n, _, H, W = im_data.shape
        if n > self.frames:
            raise ValueError("DetectStep.stage: %d frames, the step was built for %d" % (n, self.frames))
        fs = self._shape(H, W)
        self._place(fs, im_data)
        self._set_info(im_info, n)
        return fs

    def stage_u8(self, frames_u8, meta):
        """The same from ``roibatchLoader(training=False, device_prep=True)`` items: a list of uint8 (H,W,3) frames and their
        meta rows (``_place_u8``)."""
        fs, info = self._place_u8(frames_u8, meta)
        self._set_info(info, len(frames_u8))
        return fs

    def _set_info(self, im_info, n):
        self.info[:n].copy_(torch.as_tensor(im_info, dtype=torch.float32).reshape(-1, 3), non_blocking=True)
        if n < self.frames:                                      # idle branches: any valid im_info row (scale > 0) will do
            self.info[n:].copy_(self.info[:1].expand(self.frames - n, 3))
        self._n = int(n)

    def launch(self):
        """Run the staged batch (asynchronous) and queue the copy of its results into pinned host memory; returns a token
        for ``collect``."""
        self._run_staged()
        dets_h, counts_h, ev = self._host[self._slot]
        dets_h.copy_(self.dets, non_blocking=True)
        counts_h.copy_(self.counts, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.dev))
        self._slot ^= 1
        return (dets_h, counts_h, ev, self._n)

    def collect(self, token):
        dets_h, counts_h, ev, n = token
        ev.synchronize()
        dets, counts = dets_h.numpy(), counts_h.numpy()
        return [[np.array(dets[f, j, :counts[f, j]]) for j in range(self.C)] for f in range(n)]


class RelationStep(_FrameGraphStep):
    """The per-frame body of test_net_SGG_emb.py (backbone, eval branch of forward_relation on the frame's annotated boxes --
    every ordered pair --, softmax over predicates, ``detection_output`` top-``k`` triplets) for ``frames`` frames at a time as ONE
    replayable HIP graph with a branch per frame -- ``DetectStep``'s schedule for the relation test loop.

    A frame's boxes and pairs are zero-padded to a capacity (``cap_boxes`` boxes incl. one spare, all their ordered pairs):
    a pad pair points at the spare box, whose confidence is 0, so its cells score 0 and sort behind every real cell (real
    cells are products of a softmax and confidences in (0, 1]); the relation head is row-wise in eval mode, so the real rows
    are what ``relation_frame`` computes.  A frame with more boxes grows the capacity (graphs are captured again).

    ``step(im_data, im_info, im_paths)`` -> list over the frames of ``detection_output``'s five values (five Nones for a frame
    with fewer than two boxes); ``step.run(batches)`` keeps one batch in flight while the host unpacks the previous one."""

    def __init__(self, net, frames=2, k=100, device="cuda:0", cap_boxes=9, use_graph=True, max_graphs=8):
        super().__init__(net, frames, device, use_graph, max_graphs)
        self.k, self.n_rel = int(k), int(net.vrd.n_rel)
        self._alloc(int(cap_boxes))

    def _alloc(self, cap_boxes):
        from .train import _Slot
        F_, cb = self.frames, max(int(cap_boxes), 3)
        cp = (cb - 1) * (cb - 2)
        self.cap_boxes, self.cap_pairs = cb, cp
        self.kk = min(self.k, cp * self.n_rel)
        self.invalidate_graphs()
        self.inputs = _Slot({"rois": ((F_, cb + cp, 5), torch.float32), "bounds": ((F_, cp, 2, 4), torch.int32),
                             "ix": ((F_, 2, cp), torch.int64), "conf": ((F_, cb), torch.float32)}, self.dev, host=True)
        self.out = _Slot({"pair": ((F_, self.kk), torch.int32), "pred": ((F_, self.kk), torch.int32),
                          "conf": ((F_, self.kk), torch.float32)}, self.dev)
        self._host = [(torch.zeros(self.out.nbytes, dtype=torch.uint8).pin_memory(), torch.cuda.Event()) for _ in range(2)]

    def _frame(self, fs, f):
        from .model.faster_rcnn.faster_rcnn_SGG_emb import rasterize_masks
        v = self.inputs.views
        with fs.ctx[f]:
            fmap = self.net.RCNN_base(fs.im[f:f + 1])
            rois, ix = v["rois"][f], v["ix"][f]
            score, _ = self.net.vrd.forward_device(fmap, rois[:self.cap_boxes], None, rasterize_masks(v["bounds"][f], self.dev),
                                                   ix[0], ix[1], rois=rois)
            pair, pred, conf = ops.relation_topk(score, v["conf"][f], ix[0], ix[1], self.kk)
            o = self.out.views
            o["pair"][f].copy_(pair)
            o["pred"][f].copy_(pred)
            o["conf"][f].copy_(conf)

    def stage(self, im_data, im_info, im_paths):
        """``im_data`` (n,3|4,H,W), n <= frames; ``im_info`` (n,3) on the host; ``im_paths``: the frames' keys into
        ``net.vrd.target_gt_rels``."""
        n, _, H, W = im_data.shape
        if n > self.frames or len(im_paths) != n:
            raise ValueError("RelationStep.stage: %d frames / %d paths, the step was built for %d" % (n, len(im_paths), self.frames))
        self._stage_pairs(im_info, im_paths)
        fs = self._shape(H, W)
        self._place(fs, im_data)
        return fs

    def stage_u8(self, frames_u8, meta, im_paths):
        """The same from ``roibatchLoader(training=False, device_prep=True)`` items (``_place_u8``)."""
        if len(im_paths) != len(frames_u8):
            raise ValueError("RelationStep.stage_u8: %d frames / %d paths" % (len(frames_u8), len(im_paths)))
        m = np.asarray(meta.cpu() if torch.is_tensor(meta) else meta, np.float64).reshape(-1, 5)
        self._stage_pairs(np.array([[r[1], r[2], r[3]] for r in m], np.float32), im_paths)     # im_info as the host form holds it (fp32); may grow the capacity
        fs, _ = self._place_u8(frames_u8, meta)
        return fs

    def _stage_pairs(self, im_info, im_paths):
        from .model.faster_rcnn.faster_rcnn_SGG_emb import build_eval_pair_tables
        info = np.asarray(im_info.cpu() if torch.is_tensor(im_info) else im_info, np.float64).reshape(-1, 3)
        annos = [self.net.vrd.target_gt_rels[p] for p in im_paths]
        need = max([len(a["boxes"]) for a in annos] + [1]) + 1
        if need > self.cap_boxes:
            self._alloc(need)
        cb, cp = self.cap_boxes, self.cap_pairs
        rois = np.zeros((self.frames, cb + cp, 5), np.float32)
        bounds = np.zeros((self.frames, cp, 2, 4), np.int32)
        ix = np.full((self.frames, 2, cp), cb - 1, np.int64)           # pad pairs point at the spare box (confidence 0)
        conf = np.zeros((self.frames, cb), np.float32)
        meta = []
        for f, a in enumerate(annos):
            nb = len(a["boxes"])
            if nb < 2:
                meta.append(None)
                continue
            ih, iw, sc = info[f]
            boxes = np.array(a["boxes"], np.float64).reshape(-1, 4) * sc
            union, bnd, ixs, ixo = build_eval_pair_tables(boxes, ih, iw)
            rois[f, :nb, 1:] = boxes
            rois[f, cb:cb + len(ixs), 1:] = union
            bounds[f, :len(ixs)] = bnd
            ix[f, 0, :len(ixs)], ix[f, 1, :len(ixs)] = ixs, ixo
            conf[f, :nb] = 1                                            # :608 every annotated box enters with confidence 1
            meta.append((np.array(a["boxes"], np.float64).reshape(-1, 4), np.asarray(a["box_classes"]), ixs, ixo))
        self.inputs.write_host({"rois": rois, "bounds": bounds, "ix": ix, "conf": conf})
        self._meta = meta

    def launch(self):
        self._run_staged()
        hb, ev = self._host[self._slot]
        hb.copy_(self.out.buf, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.dev))
        self._slot ^= 1
        return (hb, ev, self._meta, self.out.spec, self.kk)          # the buffers travel with the token: a capacity growth replaces them

    def collect(self, token):
        hb, ev, meta, spec, kk = token
        ev.synchronize()
        host = {name: hb[o:o + n].view(dt).view(shape).numpy() for name, o, n, dt, shape in spec}
        res = []
        for f, m in enumerate(meta):
            if m is None:
                res.append((None,) * 5)
                continue
            boxes, classes, ixs, ixo = m
            pair, pred, tconf = host["pair"][f], host["pred"][f], host["conf"][f]
            n = min(int((pair < len(ixs)).sum()), len(ixs) * self.n_rel)          # real cells are a prefix (pad cells score 0)
            pair, pred, tconf = pair[:n].astype(np.int64), pred[:n], np.array(tconf[:n])
            rlp, sub, obj = np.zeros((self.k, 3), np.float64), np.zeros((self.k, 4), np.float64), np.zeros((self.k, 4), np.float64)
            sub[:n], obj[:n] = boxes[ixs[pair]], boxes[ixo[pair]]
            rlp[:n] = np.stack([classes[ixs[pair]], pred, classes[ixo[pair]]], 1)
            res.append((rlp, tconf, sub, obj, pair))
        return res


@torch.no_grad()
def detect_frame(net, im_data, im_info, gt_boxes, num_boxes, thresh=0.0, max_per_image=100, class_agnostic=False):
    """Returns ``all_boxes_i``: list over classes of (n_j, 5) float32 arrays [x1,y1,x2,y2,score] in original-image
    coordinates (entry 0, background, is empty) -- the reference's ``all_boxes[j][i]`` for this frame."""
    assert im_data.shape[0] == 1, "the reference evaluates one frame at a time (test_net_...:95 batch_size 1)"
    if hasattr(net, "forward_detect") and not net.training:
        out = net.forward_detect(im_data, im_info)        # the three outputs read below; the discriminators' are not computed
    else:
        out = net(im_data, im_info, gt_boxes, num_boxes)
    rois, cls_prob, bbox_pred = out[0], out[1], out[2]
    info = im_info.reshape(-1).tolist()
    stds = means = None
    if cfg.TEST.BBOX_REG and cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED:
        stds, means = cfg.TRAIN.BBOX_NORMALIZE_STDS, cfg.TRAIN.BBOX_NORMALIZE_MEANS
    R, C = cls_prob.shape[-2], cls_prob.shape[-1]
    if not cfg.TEST.BBOX_REG:                      # :167-169 "simply repeat the boxes": zero deltas, no de-normalisation
        bbox_pred = torch.zeros((R, 4), device=rois.device)
        class_agnostic, stds, means = True, None, None
    dets, counts = ops.detection_postprocess(rois, cls_prob, bbox_pred, info[0], info[1], info[2], class_agnostic, stds, means,
                                             thresh, cfg.TEST.NMS, max_per_image)
    counts = counts.cpu().numpy()                  # the one synchronisation of the frame
    dets = dets.cpu().numpy()
    return [np.ascontiguousarray(dets[j, :counts[j]]) for j in range(C)]


def detection_output(vrd_data, k=100):
    """lib/utils.py:584-628.  Returns (rlp_labels_im (100,3), tuple_confs_im (k',), sub_bboxes_im (100,4),
    obj_bboxes_im (100,4), rel_idex (k',)) or five Nones when the frame has fewer than two boxes."""
    if len(vrd_data["bboxes"]) <= 1:
        return None, None, None, None, None
    ixs, ixo = np.asarray(vrd_data["ixs"]), np.asarray(vrd_data["ixo"])
    boxes, classes = np.asarray(vrd_data["bboxes"], np.float64), np.asarray(vrd_data["classes"])
    rel = vrd_data["rel_score"]
    dev = rel.device
    conf = torch.as_tensor(np.asarray(vrd_data["scores"], np.float32), device=dev)
    pair, pred, tconf = ops.relation_topk(rel, conf, torch.as_tensor(ixs, device=dev), torch.as_tensor(ixo, device=dev), k)
    pair, pred, tconf = pair.cpu().numpy(), pred.cpu().numpy(), tconf.cpu().numpy()      # <= 100 rows
    n = pair.shape[0]
    rlp = np.zeros((k, 3), np.float64)
    sub = np.zeros((k, 4), np.float64)
    obj = np.zeros((k, 4), np.float64)
    sub[:n], obj[:n] = boxes[ixs[pair]], boxes[ixo[pair]]
    rlp[:n] = np.stack([classes[ixs[pair]], pred, classes[ixo[pair]]], 1)
    return rlp, tconf, sub, obj, pair.astype(np.int64)


@torch.no_grad()
def relation_frame(net, im_data, im_info, im_path, k=100):
    """test_net_SGG_emb.py per-frame work: backbone, eval relation branch, top-k triplets."""
    fmap = net.RCNN_base(im_data)
    vrd_data = net.forward_relation_eval(fmap, im_info, im_path)
    return vrd_data, detection_output(vrd_data, k)
