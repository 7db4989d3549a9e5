"""SGG_emb relation head ``vrd`` and its ``resnet`` wrapper (faster_rcnn/resnet_SGG_emb.py).

The reference pushes the subject/object boxes and the union boxes through ``fc6`` (50176 -> 4096,
822 MB of fp32 weights) in two separate passes per frame.  Here every row of every frame of the batch
-- boxes and union boxes alike -- goes through fc6/fc7 in ONE GEMM, so the 822 MB weight is streamed
once per step (forward) instead of 2 x frames times."""
import math
import pickle

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from i2vsgg_amd import ops
from ..roi_layers import ROIPool
from ..utils.config import cfg
from .faster_rcnn_SGG_emb import _fasterRCNN
from .layers import C4Base, load_reference_state, make_layer
from .utils import FC, Conv2d, Linear

RESNET_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


def _load_pickle(path):
    with open(path, "rb") as f:
        return pickle.load(f, encoding="bytes")


class vrd(nn.Module):
    """resnet_SGG_emb.py:65-221.  ``args`` needs num_relations, num_classes, emb_dim, and may carry ``use_obj_visual``
    (default True: the [subject | object] visual embedding branch ``fc_so``, :96-98 / :166-170), ``spatial_type`` (default 2:
    the dual-mask conv branch ``conv_lo`` + ``fc_lov = FC(64, 256)``; 1: ``fc_lov = FC(8, 256)`` on the 8-d relative-location
    feature of ``_getRelativeLoc``, :100-102 / :172-174; anything else: no spatial branch) and the three pickle paths;
    synthetic runs assign ``source_gt_rels`` directly.  ``fc_fusion`` is FC(256 x branches, 256) as in the reference (:94-123).
    The reference's scripts only ever run the defaults (its ``type=bool`` flags parse every CLI value to True);
    the captured training step (``train.SGGEmbStep``) packs the default form's inputs and says so for the others."""

    def __init__(self, args, all_obj_vecs=None, all_prd_vecs=None, bn=False):
        super().__init__()
        assert not bn
        self.args = args
        self.n_rel, self.n_obj, self.emb_dim = args.num_relations, args.num_classes, args.emb_dim
        self.obj_vecs, self.prd_vecs = all_obj_vecs, all_prd_vecs
        self._so_prior = None
        self.source_gt_rels, self.target_gt_rels = {}, {}
        if getattr(args, "source_so_prior_path", None):
            self._so_prior = np.array(_load_pickle(args.source_so_prior_path))
        if getattr(args, "source_gt_rels_path", None):
            self.source_gt_rels = _load_pickle(args.source_gt_rels_path)
        if getattr(args, "target_gt_rels_path", None):
            self.target_gt_rels = _load_pickle(args.target_gt_rels_path)
        self.use_obj_visual = bool(getattr(args, "use_obj_visual", True))
        self.spatial_type = int(getattr(args, "spatial_type", 2))

        self.roi_pool = ROIPool((cfg.POOLING_SIZE, cfg.POOLING_SIZE), 1.0 / 16.0)
        self.fc6 = FC(1024 * 7 * 7, 4096)
        self.fc7 = FC(4096, 4096)
        self.so_vis_embeddings = FC(4096, self.emb_dim, relu=False)
        self.fc8 = FC(4096, 256)
        self.criterion = nn.BCEWithLogitsLoss()
        n_fusion = 256
        if self.use_obj_visual:
            self.fc_so = FC(300 * 2, 256)
            n_fusion += 256
        if self.spatial_type == 1:
            self.fc_lov = FC(8, 256)
            n_fusion += 256
        elif self.spatial_type == 2:
            self.conv_lo = nn.Sequential(Conv2d(2, 96, 5, same_padding=True, stride=2),
                                         Conv2d(96, 128, 5, same_padding=True, stride=2),
                                         Conv2d(128, 64, 8, same_padding=False))
            self.fc_lov = FC(64, 256)
            n_fusion += 256
        self.fc_fusion = FC(n_fusion, 256)
        self.fc_rel = FC(256, self.emb_dim, relu=False)
        self.prd_sem_embeddings = nn.Sequential(Linear(300, 1024), nn.LeakyReLU(0.1), Linear(1024, self.emb_dim))
        self.dropout = True          # F.dropout(training=self.training) of the reference (:149-163)
        self._prd_dev = None
        self.tp = None               # (rank, world) once enable_fc6_tp() has cut fc6 by output columns

    def enable_fc6_tp(self, rank, world):
        """Cut fc6 (50176 -> 4096, 822 MB) by output columns across the data-parallel ranks (i2vsgg_amd.parallel:
        the layer's gradient never crosses xGMI and its update stays fused).  ``fc6.fc.weight`` / ``.bias`` become this
        rank's (4096/world, 50176) / (4096/world,) shards; ``gather_fc6()`` reassembles the full tensors."""
        from i2vsgg_amd import parallel
        w, b = self.fc6.fc.weight.data, self.fc6.fc.bias.data
        assert self.tp is None and w.shape[0] % world == 0
        n = w.shape[0] // world
        self.fc6.fc.weight = parallel.mark_local(nn.Parameter(w[rank * n:(rank + 1) * n].clone()))
        self.fc6.fc.bias = parallel.mark_local(nn.Parameter(b[rank * n:(rank + 1) * n].clone()))
        self.tp = (rank, world)

    def gather_fc6(self):
        """Full (4096, 50176) weight and (4096,) bias from the column shards (checkpointing)."""
        from i2vsgg_amd import parallel
        if self.tp is None:
            return self.fc6.fc.weight.data, self.fc6.fc.bias.data
        return parallel.gather_rows(self.fc6.fc.weight.data), parallel.gather_rows(self.fc6.fc.bias.data.view(-1, 1)).view(-1)

    def _dev(self, x, dtype=torch.float32):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        return x.to(device=self.fc6.fc.weight.device, dtype=dtype)

    def _drop(self, x):
        return F.dropout(x, training=self.training) if self.dropout else x

    def forward(self, fmap, boxes, rel_boxes, SpatialFea, classes, ix1, ix2):
        """Reference signature.  ``fmap`` (B,1024,H,W) (numpy as in the reference, or a device tensor);
        boxes (nb,5), rel_boxes (nr,5) with the frame index in column 0; ix1/ix2 index rows of ``boxes``.
        Returns (prd_cls_scores (nr,n_rel), relation feature (nr,emb_dim) as a numpy array)."""
        scores, x = self.forward_device(self._dev(fmap), self._dev(boxes), self._dev(rel_boxes),
                                        self._dev(SpatialFea), self._dev(ix1, torch.long), self._dev(ix2, torch.long))
        return scores, x.detach().cpu().numpy()

    def forward_device(self, fmap, boxes, rel_boxes, spatial, ix1, ix2, rois=None, ix12=None):
        """``rois`` / ``ix12`` (optional): cat(boxes, rel_boxes) and cat(ix1, ix2) when the caller already holds them in one
        buffer (a training step packs them so on the host)."""
        nb = boxes.size(0)
        if rois is None:
            rois = torch.cat((boxes, rel_boxes), 0)
        pooled = self.roi_pool(fmap, rois)                       # (nb+nr, 1024, 7, 7), NCHW flatten order
        x6 = pooled.view(pooled.size(0), -1)
        if self.tp is None:
            h = self.fc6(x6)                                     # one pass over the 822 MB weight
        else:                                                    # column-parallel fc6: everybody's rows, my columns
            from i2vsgg_amd import parallel
            h = parallel.ColShardToOwnRows.apply(self.fc6(parallel.gather_rows(x6.detach())))
        h = self.fc7(self._drop(h))
        h = self._drop(h)
        h_box, h_rel = torch.split(h, [nb, h.size(0) - nb])      # one cat in the backward (two slices = 2 fills + 2 copies + add)
        parts = [self.fc8(h_rel)]
        if self.use_obj_visual:
            obj = self.so_vis_embeddings(h_box)
            parts.append(self.fc_so(ops.pair_gather(obj, ix1, ix2)))       # [subject | object] per pair: one kernel each way
        if self.spatial_type == 1:
            parts.append(self.fc_lov(spatial.reshape(spatial.size(0), -1)))
        elif self.spatial_type == 2:
            lo = self.conv_lo(spatial)
            parts.append(self.fc_lov(lo.reshape(lo.size(0), -1)))
        x = self.fc_rel(self.fc_fusion(torch.cat(parts, 1) if len(parts) > 1 else parts[0]))
        if self._prd_dev is None or self._prd_dev.device != x.device:
            self._prd_dev = torch.from_numpy(np.asarray(self.prd_vecs, np.float32)).to(x.device)
        sem = ops.l2norm_rows(self.prd_sem_embeddings(self._prd_dev))       # F.normalize(p=2, dim=1), one kernel each way
        # logits = V . S^T through the same implicit-GEMM kernel as every other layer (a torch.mm here drags a
        # hipBLASLt launch with its device-side argument upload into the captured step)
        scores = ops.linear(ops.l2norm_rows(x), sem)
        if not self.training:
            scores = F.softmax(scores, dim=1)
        return scores, x

    # ---- host helpers with the reference names (float64 like the reference's numpy) -------------
    def _getUnionBBox(self, aBB, bBB, ih, iw, margin=10):
        return [max(0, min(aBB[0], bBB[0]) - margin), max(0, min(aBB[1], bBB[1]) - margin),
                min(iw, max(aBB[2], bBB[2]) + margin), min(ih, max(aBB[3], bBB[3]) + margin)]

    def _getRelativeLoc(self, aBB, bBB):
        """:258-264 (the spatial_type == 1 feature): offsets and log size ratios of the subject and object boxes, float32."""
        sx1, sy1, sx2, sy2 = np.asarray(aBB).astype(np.float32)
        ox1, oy1, ox2, oy2 = np.asarray(bBB).astype(np.float32)
        sw, sh, ow, oh = sx2 - sx1, sy2 - sy1, ox2 - ox1, oy2 - oy1
        xy = np.array([(sx1 - ox1) / ow, (sy1 - oy1) / oh, (ox1 - sx1) / sw, (oy1 - sy1) / sh])
        wh = np.log(np.array([sw / ow, sh / oh, ow / sw, oh / sh]))
        return np.hstack((xy, wh))

    def _getDualMask(self, ih, iw, bb):
        rh, rw = 32.0 / ih, 32.0 / iw
        x1, x2 = max(0, int(math.floor(bb[0] * rw))), min(32, int(math.ceil(bb[2] * rw)))
        y1, y2 = max(0, int(math.floor(bb[1] * rh))), min(32, int(math.ceil(bb[3] * rh)))
        mask = np.zeros((32, 32))
        mask[y1:y2, x1:x2] = 1
        return mask


class resnet(_fasterRCNN):
    def __init__(self, classes, args, num_layers=101, pretrained=False, class_agnostic=False,
                 obj_vecs=None, prd_vecs=None):
        self.dout_base_model = 1024
        self.pretrained = pretrained
        self.class_agnostic = class_agnostic
        self.layers = num_layers
        self.args = args
        self.obj_vecs, self.prd_vecs = obj_vecs, prd_vecs      # GloVe rows are an input array here
        _fasterRCNN.__init__(self, classes, args)

    def _init_modules(self):
        blocks = RESNET_BLOCKS[self.layers]
        self.RCNN_base = C4Base(blocks[:3])
        self.vrd = vrd(self.args, self.obj_vecs, self.prd_vecs)
        layer4, _ = make_layer(1024, 512, blocks[3], 2)
        self.RCNN_top = nn.Sequential(layer4)
        self.RCNN_cls_score = Linear(2048, self.n_classes)
        self.RCNN_bbox_pred = Linear(2048, 4 if self.class_agnostic else 4 * self.n_classes)
        for p in self.RCNN_base[0].parameters():
            p.requires_grad = False
        if self.pretrained:
            path = cfg.RESNET_PATH if self.layers == 101 else cfg.RESNET_PATH50
            sd = torch.load(path, map_location="cpu")
            names = {"conv1": "RCNN_base.0", "bn1": "RCNN_base.1", "layer1": "RCNN_base.4", "layer2": "RCNN_base.5",
                     "layer3": "RCNN_base.6", "layer4": "RCNN_top.0"}
            load_reference_state(self, {names[k.split(".")[0]] + k[len(k.split(".")[0]):]: v for k, v in sd.items()
                                        if k.split(".")[0] in names}, strict=False)

    def train(self, mode=True):
        nn.Module.train(self, mode)
        return self

    def _head_to_tail(self, pool5):
        return self.RCNN_top(pool5).mean(3).mean(2)
