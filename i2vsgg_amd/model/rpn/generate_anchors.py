"""Base anchor table (rpn/generate_anchors.py:45-56): ratio-major, then scale; float64, 0-based."""
import numpy as np


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=2 ** np.arange(3, 6)):
    ratios = np.asarray(ratios, dtype=np.float64).reshape(-1, 1)
    scales = np.asarray(scales, dtype=np.float64).reshape(1, -1)
    centre = (base_size - 1) / 2.0
    w0 = np.round(np.sqrt(base_size * base_size / ratios))       # half-to-even like the reference
    h0 = np.round(w0 * ratios)
    half_w = ((w0 * scales).reshape(-1, 1) - 1) / 2.0
    half_h = ((h0 * scales).reshape(-1, 1) - 1) / 2.0
    return np.hstack([centre - half_w, centre - half_h, centre + half_w, centre + half_h])


def shifted_anchors(feat_h, feat_w, feat_stride, base):
    """All anchors of a feature map in (y, x, a) order -> (H*W*A, 4) float32 (proposal_layer.py:81-95)."""
    ys, xs = np.meshgrid(np.arange(feat_h) * feat_stride, np.arange(feat_w) * feat_stride, indexing="ij")
    shift = np.stack([xs, ys, xs, ys], -1).reshape(-1, 1, 4).astype(np.float32)
    return (shift + base.astype(np.float32)[None]).reshape(-1, 4)
