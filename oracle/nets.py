"""torch-CPU fp32 restatement of the dense parts of the path (TEST INFRASTRUCTURE --
see oracle/__init__.py).  Plain ``torch.nn.functional`` ops on CPU tensors, driven by
a flat ``{reference state_dict key: tensor}`` mapping so the same seeded weights can
be loaded into the reference modules (tools/gen_golden.py) and into the HIP path.

Citations are into /root/reference/lib/model/faster_rcnn.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import cops

BN_EPS = 1e-5  # nn.BatchNorm2d default, resnet_instance_styleD_bilinear.py:186-191


def _bn(x, p, k):
    return F.batch_norm(x, p[k + ".running_mean"], p[k + ".running_var"],
                        p[k + ".weight"], p[k + ".bias"], False, 0.0, BN_EPS)


def bottleneck(x, p, k, stride):
    """resnet_instance_styleD_bilinear.py:181-217 (stride on the first 1x1, caffe style)."""
    out = F.relu(_bn(F.conv2d(x, p[k + ".conv1.weight"], stride=stride), p, k + ".bn1"))
    out = F.relu(_bn(F.conv2d(out, p[k + ".conv2.weight"], padding=1), p, k + ".bn2"))
    out = _bn(F.conv2d(out, p[k + ".conv3.weight"]), p, k + ".bn3")
    if (k + ".downsample.0.weight") in p:
        x = _bn(F.conv2d(x, p[k + ".downsample.0.weight"], stride=stride), p, k + ".downsample.1")
    return F.relu(out + x)


def layer(x, p, k, nblocks, stride):
    for i in range(nblocks):
        x = bottleneck(x, p, "%s.%d" % (k, i), stride if i == 0 else 1)
    return x


def stem(x, p):
    """conv1 7x7/2 p3 + BN + ReLU + maxpool k3 s2 p0 ceil (:224-228)."""
    x = F.relu(_bn(F.conv2d(x, p["RCNN_base.0.weight"], stride=2, padding=3), p, "RCNN_base.1"))
    return F.max_pool2d(x, 3, 2, 0, ceil_mode=True)


def extract_feature(im, p, blocks=(3, 4, 23)):
    """resnet.extract_feature (:412-420): returns (base_feat C4, base_feat1 = layer2 tap)."""
    x = stem(im, p)
    x = layer(x, p, "RCNN_base.4", blocks[0], 1)
    feat1 = layer(x, p, "RCNN_base.5", blocks[1], 2)
    feat = layer(feat1, p, "RCNN_base.6", blocks[2], 2)
    return feat, feat1


def head_to_tail(pool5, p, nblocks=3):
    """_head_to_tail (:441-443): layer4 then spatial mean."""
    return layer(pool5, p, "RCNN_top.0", nblocks, 2).mean(3).mean(2)


def rpn_head(feat, p):
    """rpn/rpn.py:63-72: 3x3 conv+ReLU, cls 1x1 -> pairwise (bg,fg) softmax, bbox 1x1."""
    x = F.relu(F.conv2d(feat, p["RCNN_rpn.RPN_Conv.weight"], p["RCNN_rpn.RPN_Conv.bias"], padding=1))
    cls = F.conv2d(x, p["RCNN_rpn.RPN_cls_score.weight"], p["RCNN_rpn.RPN_cls_score.bias"])
    B, C2, H, W = cls.shape
    prob = F.softmax(cls.view(B, 2, C2 // 2 * H, W), 1).view(B, C2, H, W)
    box = F.conv2d(x, p["RCNN_rpn.RPN_bbox_pred.weight"], p["RCNN_rpn.RPN_bbox_pred.bias"])
    return cls, prob, box


class _GRL(torch.autograd.Function):
    """utils/net_utils.py:52-61."""

    @staticmethod
    def forward(ctx, x, lam):
        ctx.lam = lam
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * -ctx.lam, None


def netd_pixel(x, p, lam=1.0, context=False):
    """netD_pixel.forward (:66-83); no biases."""
    x = _GRL.apply(x, lam)
    x = F.relu(F.conv2d(x, p["netD_pixel.conv1.weight"]))
    x = F.relu(F.conv2d(x, p["netD_pixel.conv2.weight"]))
    feat = F.avg_pool2d(x, (x.size(2), x.size(3))) if context else None
    d = torch.sigmoid(F.conv2d(x, p["netD_pixel.conv3.weight"]))
    return (d, feat) if context else d


def netd_style(x, p, lam=1.0, context=False, dim=512, rank=5):
    """netD_style.forward (:122-146): factorised bilinear pooling discriminator."""
    x = _GRL.apply(x, lam)
    b, c, h, w = x.shape
    x = x.reshape(b, c, -1).permute(0, 2, 1)
    x1 = F.linear(x, p["netD_style.fc_1.weight"], p["netD_style.fc_1.bias"])
    x2 = F.linear(x, p["netD_style.fc_2.weight"], p["netD_style.fc_2.bias"])
    z = (x1 * x2).reshape(b, h * w, dim, rank).sum(-1).sum(1)
    z = torch.sqrt(F.relu(z)) - torch.sqrt(F.relu(-z))
    z = F.normalize(z, p=2, dim=1)
    d = torch.sigmoid(F.linear(z, p["netD_style.fc1.weight"], p["netD_style.fc1.bias"]))
    return (d, z) if context else d


def _fc(x, p, k, relu=True):
    y = F.linear(x, p[k + ".fc.weight"], p[k + ".fc.bias"])
    return F.relu(y) if relu else y


def _convrelu(x, p, k, stride, pad):
    return F.relu(F.conv2d(x, p[k + ".conv.weight"], p[k + ".conv.bias"], stride=stride, padding=pad))


def relative_loc(a, b):
    """vrd._getRelativeLoc (resnet_SGG_emb.py:258-264): the 8-d spatial feature of spatial_type == 1, float32 arithmetic."""
    sx1, sy1, sx2, sy2 = np.asarray(a).astype(np.float32)
    ox1, oy1, ox2, oy2 = np.asarray(b).astype(np.float32)
    sw, sh, ow, oh = sx2 - sx1, sy2 - sy1, ox2 - ox1, oy2 - oy1
    xy = np.array([(sx1 - ox1) / ow, (sy1 - oy1) / oh, (ox1 - sx1) / sw, (oy1 - sy1) / sh])
    wh = np.log(np.array([sw / ow, sh / oh, ow / sw, oh / sh]))
    return np.hstack((xy, wh))


def vrd_head(fmap, boxes, rel_boxes, spatial, ix_s, ix_o, prd_vecs, p, training=True, use_obj_visual=True, spatial_type=2):
    """vrd.forward (resnet_SGG_emb.py:128-221), dropout disabled (eval-mode dropout
    for reproducibility).  fmap (1,1024,H,W) NCHW numpy/torch;
    boxes (nb,5), rel_boxes (nr,5), spatial (nr,2,32,32) -- (nr,8) for spatial_type 1 (:172-174); returns (scores, rel_feat).
    ``use_obj_visual`` / ``spatial_type``: the branches of :166-180.
    ``training`` only selects whether the final softmax is applied (:216-219)."""
    fmap = torch.as_tensor(fmap, dtype=torch.float32)
    fm = fmap.numpy()

    def pool(rois):
        out, _ = cops.roi_pool_fwd(fm, np.asarray(rois, np.float32), 7, 7, 1.0 / 16.0)
        return torch.from_numpy(out).reshape(out.shape[0], -1)

    ix_s = torch.as_tensor(np.asarray(ix_s), dtype=torch.long)
    ix_o = torch.as_tensor(np.asarray(ix_o), dtype=torch.long)
    x_so = _fc(_fc(pool(boxes), p, "vrd.fc6"), p, "vrd.fc7")
    obj = _fc(x_so, p, "vrd.so_vis_embeddings", relu=False)
    x_s, x_o = obj.index_select(0, ix_s), obj.index_select(0, ix_o)
    x = _fc(_fc(_fc(pool(rel_boxes), p, "vrd.fc6"), p, "vrd.fc7"), p, "vrd.fc8")
    if use_obj_visual:
        x = torch.cat((x, _fc(torch.cat((x_s, x_o), 1), p, "vrd.fc_so")), 1)
    lo = torch.as_tensor(np.asarray(spatial), dtype=torch.float32)
    if spatial_type == 1:
        x = torch.cat((x, _fc(lo.reshape(lo.size(0), -1), p, "vrd.fc_lov")), 1)
    elif spatial_type == 2:
        lo = _convrelu(lo, p, "vrd.conv_lo.0", 2, 2)
        lo = _convrelu(lo, p, "vrd.conv_lo.1", 2, 2)
        lo = _convrelu(lo, p, "vrd.conv_lo.2", 1, 0)
        x = torch.cat((x, _fc(lo.reshape(lo.size(0), -1), p, "vrd.fc_lov")), 1)
    x = _fc(_fc(x, p, "vrd.fc_fusion"), p, "vrd.fc_rel", relu=False)
    sem = torch.as_tensor(prd_vecs, dtype=torch.float32)
    sem = F.linear(sem, p["vrd.prd_sem_embeddings.0.weight"], p["vrd.prd_sem_embeddings.0.bias"])
    sem = F.leaky_relu(sem, 0.1)
    sem = F.linear(sem, p["vrd.prd_sem_embeddings.2.weight"], p["vrd.prd_sem_embeddings.2.bias"])
    scores = F.normalize(x, p=2, dim=1) @ F.normalize(sem, p=2, dim=1).t()
    if not training:
        scores = F.softmax(scores, 1)
    return scores, x


# --------------------------------------------------------------------- pair builder
def union_box(a, b, ih, iw, margin=10):
    """vrd._getUnionBBox (resnet_SGG_emb.py:240-244)."""
    return [max(0, min(a[0], b[0]) - margin), max(0, min(a[1], b[1]) - margin),
            min(iw, max(a[2], b[2]) + margin), min(ih, max(a[3], b[3]) + margin)]


def dual_mask(ih, iw, bb):
    """vrd._getDualMask (resnet_SGG_emb.py:246-256): 32x32 binary mask of a box."""
    rh, rw = 32.0 / ih, 32.0 / iw
    x1, x2 = max(0, int(math.floor(bb[0] * rw))), min(32, int(math.ceil(bb[2] * rw)))
    y1, y2 = max(0, int(math.floor(bb[1] * rh))), min(32, int(math.ceil(bb[3] * rh)))
    m = np.zeros((32, 32))
    m[y1:y2, x1:x2] = 1
    return m


def build_pairs(anno_boxes, rels, im_scale, ih, iw, n_rel):
    """forward_predicate (faster_rcnn_SGG_emb.py:170-245): unique (s,o) pairs in
    first-seen order, multi-hot labels, union boxes (+10 px), dual masks."""
    gt = np.array(anno_boxes) * im_scale
    pairs, prds = [], []
    for s, o, r in rels:
        if [s, o] not in pairs:
            pairs.append([s, o])
            prds.append([r])
        else:
            prds[pairs.index([s, o])].append(r)
    ixs = np.array([q[0] for q in pairs])
    ixo = np.array([q[1] for q in pairs])
    n = len(pairs)
    rel_boxes = np.zeros((n, 5))
    labels = np.zeros((n, n_rel))
    spatial = np.zeros((n, 2, 32, 32))
    for i in range(n):
        sb, ob = gt[ixs[i]], gt[ixo[i]]
        rel_boxes[i, 1:5] = union_box(sb, ob, ih, iw)
        spatial[i, 0], spatial[i, 1] = dual_mask(ih, iw, sb), dual_mask(ih, iw, ob)
        labels[i, prds[i]] = 1
    boxes = np.zeros((gt.shape[0], 5), np.float32)
    boxes[:, 1:5] = gt
    return boxes, rel_boxes, spatial, labels, ixs, ixo


def detection_output(vrd_data, k=100):
    """lib/utils.py:584-628 with ``rel_score`` given as a numpy array.  Tie rule of the argsort: descending value,
    ascending flat index (numpy's default quicksort leaves ties unspecified)."""
    if len(vrd_data["bboxes"]) <= 1:
        return None, None, None, None, None
    ixs, ixo = np.asarray(vrd_data["ixs"]), np.asarray(vrd_data["ixo"])
    boxes, classes, confs = vrd_data["bboxes"], vrd_data["classes"], vrd_data["scores"]
    rel_prob = np.array(vrd_data["rel_score"], np.float32)
    rlp = np.zeros((k, 3), np.float64)
    sub = np.zeros((k, 4), np.float64)
    obj = np.zeros((k, 4), np.float64)
    for i in range(rel_prob.shape[0]):
        rel_prob[i] = rel_prob[i] * float(confs[ixs[i]]) * float(confs[ixo[i]])     # float32 row x python floats (:613)
    order = np.argsort(-rel_prob.ravel(), kind="stable")
    rel_res = np.dstack(np.unravel_index(order, rel_prob.shape))[0][:k]
    tconf, ridx = [], []
    for ii in range(rel_res.shape[0]):
        t, rel = rel_res[ii, 0], rel_res[ii, 1]
        sub[ii], obj[ii] = boxes[ixs[t]], boxes[ixo[t]]
        rlp[ii] = [classes[ixs[t]], rel, classes[ixo[t]]]
        tconf.append(rel_prob[t, rel])
        ridx.append(t)
    return rlp, np.array(tconf), sub, obj, np.array(ridx)


def eval_pair_tables(boxes_scaled, ih, iw, union_fn, mask_fn):
    """Eval branch of forward_relation (faster_rcnn_SGG_emb.py:597-652), the reference's double loop as written:
    all ordered pairs i != j, union boxes via ``_getUnionBBox``, dual masks via ``_getDualMask``."""
    n = len(boxes_scaled)
    ixs, ixo = [], []
    for i in range(n):
        for j in range(n):
            if i != j:
                ixs.append(i)
                ixo.append(j)
    rel_boxes = np.zeros((len(ixs), 5))
    masks = np.zeros((len(ixs), 2, 32, 32))
    for t in range(len(ixs)):
        s, o = boxes_scaled[ixs[t]], boxes_scaled[ixo[t]]
        rel_boxes[t, 1:5] = np.array(union_fn(s, o, ih, iw))
        masks[t] = [mask_fn(ih, iw, s), mask_fn(ih, iw, o)]
    return np.array(ixs), np.array(ixo), rel_boxes, masks
