"""``get_minibatch(roidb, num_classes)`` (lib/roi_data_layer/minibatch.py:19-55): one image ->
blobs ``data`` (1,H,W,3) fp32 BGR mean-subtracted, shorter side scaled to cfg.TRAIN.SCALES with NO
max-size clamp (blob.py:44-50), ``gt_boxes`` (n,5) scaled, ``im_info`` (1,3) = [H, W, scale],
``img_id``, ``path``.  The resize is cv2's float INTER_LINEAR (half-pixel centres, fx = fy = the scale).

``get_minibatch_device`` is the same contract with the image work on the GPU: the decoded
uint8 image is uploaded (a quarter of the float blob's bytes) and ``ops.image_prep`` does BGR swap, flip, mean
subtraction, resize and the 4-channel NHWC placement; ``data`` is then a (1,4,H,W) channels_last device tensor that
the backbone's stem consumes as is."""
import numpy as np
import numpy.random as npr
import torch

from ..model.utils.config import cfg


def _axis(n_src, n_dst, inv_f):
    pos = ((np.arange(n_dst, dtype=np.float64) + 0.5) * inv_f - 0.5).astype(np.float32)
    s = np.floor(pos).astype(np.int64)
    a = (pos - s.astype(np.float32)).astype(np.float32)
    s, a = np.where(s < 0, 0, s), np.where(s < 0, np.float32(0), a)
    hi = s >= n_src - 1
    return np.where(hi, n_src - 1, s), np.minimum(np.where(hi, n_src - 1, s) + 1, n_src - 1), np.where(hi, np.float32(0), a)


def resize_linear(im, f):
    """cv2.resize(im, None, None, fx=f, fy=f, interpolation=cv2.INTER_LINEAR) on a float32 image: output size
    round-half-even(src * f), source position (d + 0.5) / f - 0.5, edge taps clamped, horizontal pass then
    vertical, fp32 throughout (host twin of csrc/image.hip)."""
    im = np.asarray(im, np.float32)
    H, W = im.shape[:2]
    Ho, Wo = int(np.rint(H * f)), int(np.rint(W * f))
    sx, sx1, ax = _axis(W, Wo, 1.0 / f)
    sy, sy1, ay = _axis(H, Ho, 1.0 / f)
    ax, ay = ax.astype(np.float32)[None, :, None], ay.astype(np.float32)[:, None, None]
    one = np.float32(1.0)
    rows = ((im[:, sx] * (one - ax)).astype(np.float32) + (im[:, sx1] * ax).astype(np.float32)).astype(np.float32)
    return ((rows[sy] * (one - ay)).astype(np.float32) + (rows[sy1] * ay).astype(np.float32)).astype(np.float32)


def _read_image(entry):
    """-> (H,W,3) uint8/float RGB array.  Synthetic entries generate seeded pixels."""
    if "pixels_seed" in entry:
        rng = np.random.default_rng(entry["pixels_seed"])
        return rng.integers(0, 256, (entry["height"], entry["width"], 3), dtype=np.uint8)
    import PIL.Image
    return np.asarray(PIL.Image.open(entry["image"]).convert("RGB"))


def prep_im_for_blob(im, pixel_means, target_size, max_size=None):
    """blob.py:35-52: mean-subtract, scale the shorter side to target_size (max_size ignored there too)."""
    im = im.astype(np.float32, copy=True)
    im -= np.asarray(pixel_means, np.float32).reshape(1, 1, -1)
    scale = float(target_size) / float(min(im.shape[:2]))
    return resize_linear(im, scale), scale


def _gt_blob(e, scale):
    if cfg.TRAIN.USE_ALL_GT:
        inds = np.where(e["gt_classes"] != 0)[0]
    else:
        inds = np.where((e["gt_classes"] != 0) & np.all(e["gt_overlaps"].toarray() > -1.0, axis=1))[0]
    gt = np.empty((len(inds), 5), dtype=np.float32)
    gt[:, :4] = e["boxes"][inds, :] * scale
    gt[:, 4] = e["gt_classes"][inds]
    return gt


def get_minibatch_device(roidb, num_classes, device="cuda:0"):
    """get_minibatch with the image work on the device; ``data`` is a (1,4,H,W) channels_last CUDA tensor."""
    from .. import ops
    assert len(roidb) == 1, "Single batch only"
    scale_ind = npr.randint(0, high=len(cfg.TRAIN.SCALES), size=1)[0]
    e = roidb[0]
    im = np.ascontiguousarray(_read_image(e))
    if im.ndim == 2:
        im = np.repeat(im[:, :, None], 3, 2)
    u8 = torch.from_numpy(im.astype(np.uint8, copy=False)).to(device, non_blocking=True)
    blob, (ho, wo, scale) = ops.image_prep(u8, cfg.PIXEL_MEANS, cfg.TRAIN.SCALES[scale_ind], flipped=bool(e["flipped"]), rgb=True)
    return {"data": blob, "gt_boxes": _gt_blob(e, scale), "im_info": np.array([[ho, wo, scale]], np.float32),
            "img_id": e["img_id"], "path": e["image"]}


def get_minibatch(roidb, num_classes):
    assert len(roidb) == 1, "Single batch only"
    scale_ind = npr.randint(0, high=len(cfg.TRAIN.SCALES), size=1)[0]
    e = roidb[0]
    im = _read_image(e)[:, :, ::-1]                       # RGB -> BGR
    if e["flipped"]:
        im = im[:, ::-1, :]
    im, scale = prep_im_for_blob(im, cfg.PIXEL_MEANS, cfg.TRAIN.SCALES[scale_ind], cfg.TRAIN.MAX_SIZE)
    return {"data": im[None], "gt_boxes": _gt_blob(e, scale), "im_info": np.array([[im.shape[0], im.shape[1], scale]], np.float32),
            "img_id": e["img_id"], "path": e["image"]}
