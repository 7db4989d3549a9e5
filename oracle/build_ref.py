"""oracle/_ref: the reference's OWN C for RoIAlign forward, compiled where it lies (TEST INFRASTRUCTURE).

``/root/reference/lib/model/roi_align/src/roi_align.c`` as a file needs ``<TH/TH.h>`` (its two wrappers at :16-78
unpack ``THFloatTensor*``), which this image does not have, and no stand-in header is written.  But the function the
wrappers call, ``ROIAlignForwardCpu`` (:80-136), is self-contained: plain ``const float*`` / ``int`` arguments,
``fmaxf`` / ``fminf`` / ``floor`` from ``<math.h>`` and nothing else.  This recipe finds that one definition in the
file (by its signature and brace matching, so it does not depend on line numbers), pipes its text UNMODIFIED to gcc on
stdin with ``-include math.h`` on the command line, and writes ``oracle/_ref/libref_roi_align.so``.  No reference text
is written to disk anywhere: the repo gets only the shared object (git-ignored; it travels to the GPU box as a built
artefact like the product's own .so) and the golden vectors made with it (``tests/golden/roi_align_fwd.npz``, tier
"extracted").

``ROIAlignBackwardCpu`` (:138-190) is NOT built: its bounds test is inverted (:175) and nothing calls it
(functions/roi_align.py:38 asserts is_cuda).  The backward is pinned through the forward: it is the transpose of a
linear map whose every coefficient the pinned forward exposes (tests/test_oracle_golden.py).

Build container only -- a GPU box has no /root/reference and uses the prebuilt file.
"""
import ctypes
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_C = "/root/reference/lib/model/roi_align/src/roi_align.c"
OUT_DIR = os.path.join(_HERE, "_ref")
SO = os.path.join(OUT_DIR, "libref_roi_align.so")
CFLAGS = ["-O2", "-fPIC", "-shared", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math"]
_lib = None


def _definition(text, name="ROIAlignForwardCpu"):
    """The text of ``void <name>(...) {...}`` -- the definition, not the prototype -- and its 1-based line span."""
    at = 0
    while True:
        at = text.index("void " + name + "(", at)
        close = text.index(")", at)
        rest = text[close + 1:].lstrip()
        if rest.startswith("{"):
            break
        at = close
    i = text.index("{", close)
    depth, j = 0, i
    while True:
        depth += {"{": 1, "}": -1}.get(text[j], 0)
        j += 1
        if depth == 0:
            break
    return text[at:j], (text.count("\n", 0, at) + 1, text.count("\n", 0, j) + 1)


def available():
    return os.path.exists(SO)


def build(force=False):
    """Returns the .so path, or None when neither the reference nor a prebuilt file is here."""
    if not os.path.exists(REF_C):
        return SO if os.path.exists(SO) else None
    src = open(REF_C).read()
    body, span = _definition(src)
    want = hashlib.sha256((body + " ".join(CFLAGS)).encode()).hexdigest()
    stamp = SO + ".sha256"
    have = open(stamp).read().split()[0] if os.path.exists(stamp) else None
    if force or not os.path.exists(SO) or have != want:
        os.makedirs(OUT_DIR, exist_ok=True)
        subprocess.run(["gcc", *CFLAGS, "-include", "math.h", "-x", "c", "-", "-o", SO, "-lm"],
                       input=body.encode(), check=True)
        with open(stamp, "w") as f:
            f.write("%s roi_align.c:%d-%d\n" % (want, span[0], span[1]))
    return SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise FileNotFoundError("oracle/_ref is not built (python -m oracle.build_ref, build container only)")
        _lib = ctypes.CDLL(SO)
        _lib.ROIAlignForwardCpu.restype = None
    return _lib


def roi_align_fwd(feat, rois, ah, aw, scale):
    """The reference's ``ROIAlignForwardCpu`` itself: feat NCHW fp32, rois (R,5) -> (R,C,ah,aw)."""
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    _, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.empty((R, C, ah, aw), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    lib().ROIAlignForwardCpu(feat.ctypes.data_as(fp), ctypes.c_float(scale), ctypes.c_int(R), ctypes.c_int(H),
                             ctypes.c_int(W), ctypes.c_int(C), ctypes.c_int(ah), ctypes.c_int(aw),
                             rois.ctypes.data_as(fp), out.ctypes.data_as(fp))
    return out


if __name__ == "__main__":
    print(build(force=True))
    print(open(SO + ".sha256").read().strip())
