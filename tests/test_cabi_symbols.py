"""The C-ABI library loads without a GPU and exports every entry point include/i2vsgg_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "i2vsgg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(i2v_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = _declared()
    assert len(names) >= 20
    for must in ("i2v_roi_align_fwd", "i2v_roi_align_bwd", "i2v_nms_sorted", "i2v_rpn_proposal", "i2v_conv_fwd"):
        assert must in names


def test_library_builds_and_exports_every_declared_symbol():
    from i2vsgg_amd import build
    so = build.build()
    lib = ctypes.CDLL(so)
    for name in _declared():
        assert hasattr(lib, name), "missing export: " + name
    assert lib.i2v_version() >= 100


def test_python_binding_covers_the_header():
    from i2vsgg_amd import _lib
    assert set(_declared()) == set(_lib.SIGNATURES)


def test_argument_errors_are_status_codes_not_crashes():
    """No compute without a GPU: only the argument validation in front of the launch is exercised."""
    from i2vsgg_amd import _lib
    rc = _lib.lib.i2v_roi_align_fwd(None, 0, 1, 4, 8, 8, None, 1, 7, 7, 0.0625, 1, None, 0, None)
    assert rc == -1
    assert b"null" in _lib.lib.i2v_last_error()
    rc = _lib.lib.i2v_conv_fwd(ctypes.c_void_p(16), ctypes.c_void_p(16), None, None, None, ctypes.c_void_p(16),
                               1, 8, 8, 3, 8, 1, 1, 1, 0, 0, None, 0, None)
    assert rc == -1 and b"multiple of 4" in _lib.lib.i2v_last_error()
    # split-K scratch is the caller's (no library-owned device memory): the per-shape query, and the dry plan that tells
    # a caller whether the output must start at zero, depend only on shapes and on the size the caller offers
    need = _lib.lib.i2v_conv_split_workspace_bytes(2, 38, 63, 1024, 256, 1, 1, 1, 0)        # layer3 1x1, two frames
    assert 4096 < need <= (48 << 20) + 4096
    assert _lib.lib.i2v_conv_fwd_splits(2, 38, 63, 1024, 256, 1, 1, 1, 0, 0) == 1          # no workspace: atomics
    assert _lib.lib.i2v_conv_fwd_splits(2, 38, 63, 1024, 256, 1, 1, 1, 0, need) == 0       # finished in the kernel
    assert _lib.lib.i2v_conv_split_workspace_bytes(2, 150, 250, 64, 64, 1, 1, 1, 0) == 0   # large grid: no split
    assert _lib.lib.i2v_set_tuning(99, 1) == -1 and _lib.lib.i2v_get_tuning(1) == 2
    assert _lib.lib.i2v_nms_workspace_bytes(2, 12000) == 2 * 12000 * 188 * 8
