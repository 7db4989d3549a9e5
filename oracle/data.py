"""CPU restatement of the image front-end (TEST INFRASTRUCTURE ONLY: tests/, smoke and the cpu_baseline leg).

``prep_im_for_blob`` follows model/utils/blob.py:35-52; its resize is cv2.resize(..., INTER_LINEAR) on a float32
image.  cv2 is not installed in this image and is a third-party dependency of the reference (requirements.txt lists
``opencv-python`` without a version), so its float bilinear path is restated from its documented algorithm:
**parity with cv2 itself is unpinned**; the GPU kernel is checked bit-for-bit against this restatement.
"""
import numpy as np


def _axis(n_src, n_dst, inv_f):
    d = np.arange(n_dst, dtype=np.float64)
    pos = ((d + 0.5) * inv_f - 0.5).astype(np.float32)         # evaluated in double, stored as float
    s = np.floor(pos).astype(np.int64)
    a = (pos - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    s[lo], a[lo] = 0, 0.0
    hi = s >= n_src - 1
    s[hi], a[hi] = n_src - 1, 0.0
    return s, np.minimum(s + 1, n_src - 1), a


def resize_linear(im, f):
    """cv2.resize(im, None, None, fx=f, fy=f, interpolation=cv2.INTER_LINEAR) for a float32 H x W x C image."""
    im = np.asarray(im, np.float32)
    H, W = im.shape[:2]
    Ho, Wo = int(np.rint(H * f)), int(np.rint(W * f))          # saturate_cast<int>: round half to even
    sx, sx1, ax = _axis(W, Wo, 1.0 / f)
    sy, sy1, ay = _axis(H, Ho, 1.0 / f)
    ax = ax[None, :, None]
    one = np.float32(1.0)
    rows = (im[:, sx] * (one - ax)).astype(np.float32) + (im[:, sx1] * ax).astype(np.float32)     # horizontal pass
    rows = rows.astype(np.float32)
    ay = ay[:, None, None]
    out = (rows[sy] * (one - ay)).astype(np.float32) + (rows[sy1] * ay).astype(np.float32)        # vertical pass
    return out.astype(np.float32)


def prep_im_for_blob(im_bgr_u8, pixel_means, target_size):
    """blob.py:35-52 (max_size is ignored there too): returns (float32 image, im_scale)."""
    im = im_bgr_u8.astype(np.float32)
    im -= np.asarray(pixel_means, np.float32).reshape(1, 1, 3)
    im_scale = float(target_size) / float(min(im.shape[:2]))
    return resize_linear(im, im_scale), im_scale


def minibatch_image(im_rgb_u8, pixel_means, target_size, flipped=False):
    """minibatch.py:66-81: RGB file order -> BGR, optional flip, prep_im_for_blob."""
    im = im_rgb_u8[:, :, ::-1]
    if flipped:
        im = im[:, ::-1, :]
    return prep_im_for_blob(np.ascontiguousarray(im), pixel_means, target_size)
