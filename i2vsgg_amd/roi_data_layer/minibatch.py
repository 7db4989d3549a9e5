"""``get_minibatch(roidb, num_classes)`` (lib/roi_data_layer/minibatch.py:19-55): one image ->
blobs ``data`` (1,H,W,3) fp32 BGR mean-subtracted, shorter side scaled to cfg.TRAIN.SCALES with NO
max-size clamp (blob.py:44-50), ``gt_boxes`` (n,5) scaled, ``im_info`` (1,3) = [H, W, scale],
``img_id``, ``path``.  The resize is bilinear with half-pixel centres (what cv2.INTER_LINEAR does);
moving it to the device is listed as next."""
import numpy as np
import numpy.random as npr
import torch
import torch.nn.functional as F

from ..model.utils.config import cfg


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
    im -= pixel_means.astype(np.float32)
    scale = float(target_size) / float(min(im.shape[:2]))
    h, w = int(round(im.shape[0] * scale)), int(round(im.shape[1] * scale))
    t = torch.from_numpy(im).permute(2, 0, 1).unsqueeze(0)
    t = F.interpolate(t, size=(h, w), mode="bilinear", align_corners=False)
    return t[0].permute(1, 2, 0).contiguous().numpy(), scale


def get_minibatch(roidb, num_classes):
    assert len(roidb) == 1, "Single batch only"
    scale_ind = npr.randint(0, high=len(cfg.TRAIN.SCALES), size=1)[0]
    e = roidb[0]
    im = _read_image(e)[:, :, ::-1]                       # RGB -> BGR
    if e["flipped"]:
        im = im[:, ::-1, :]
    im, scale = prep_im_for_blob(im, cfg.PIXEL_MEANS, cfg.TRAIN.SCALES[scale_ind], cfg.TRAIN.MAX_SIZE)
    if cfg.TRAIN.USE_ALL_GT:
        inds = np.where(e["gt_classes"] != 0)[0]
    else:
        inds = np.where((e["gt_classes"] != 0) & np.all(e["gt_overlaps"].toarray() > -1.0, axis=1))[0]
    gt = np.empty((len(inds), 5), dtype=np.float32)
    gt[:, :4] = e["boxes"][inds, :] * scale
    gt[:, 4] = e["gt_classes"][inds]
    return {"data": im[None], "gt_boxes": gt, "im_info": np.array([[im.shape[0], im.shape[1], scale]], np.float32),
            "img_id": e["img_id"], "path": e["image"]}
