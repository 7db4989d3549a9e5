"""Global configuration tree -- the reference's config surface (lib/model/utils/config.py).

Same keys, defaults and types as config.py:19-309 so ``cfgs/res101.yml`` / ``cfgs/res50.yml`` and
``--set``-style overrides merge unchanged; same rules as ``_merge_a_into_b`` (:344-374: unknown
key -> KeyError, type mismatch -> ValueError) and ``cfg_from_list`` (:386-406).  Differences:
``yaml.safe_load`` (the reference's bare ``yaml.load`` raises on PyYAML >= 6, :381) and no
dependency on ``easydict``.
"""
import os.path as osp
from ast import literal_eval

import numpy as np


class AttrDict(dict):
    """dict with attribute access; nested dicts are wrapped on assignment."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict(v) if type(v) is dict else v)

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


_ROOT = osp.abspath(osp.join(osp.dirname(__file__), "..", "..", ".."))

cfg = AttrDict({
    "TRAIN": {
        "LEARNING_RATE": 0.001, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.0005, "GAMMA": 0.1, "STEPSIZE": [30000],
        "DISPLAY": 10, "DOUBLE_BIAS": True, "TRUNCATED": False, "BIAS_DECAY": False, "USE_GT": False,
        "ASPECT_GROUPING": False, "SNAPSHOT_KEPT": 3, "SUMMARY_INTERVAL": 180, "SCALES": (600,), "MAX_SIZE": 1000,
        "TRIM_HEIGHT": 600, "TRIM_WIDTH": 600, "IMS_PER_BATCH": 1, "BATCH_SIZE": 128, "FG_FRACTION": 0.25,
        "FG_THRESH": 0.5, "BG_THRESH_HI": 0.5, "BG_THRESH_LO": 0.1, "USE_FLIPPED": True, "BBOX_REG": True,
        "BBOX_THRESH": 0.5, "SNAPSHOT_ITERS": 5000, "SNAPSHOT_PREFIX": "res101_faster_rcnn",
        "BBOX_NORMALIZE_TARGETS": True, "BBOX_INSIDE_WEIGHTS": (1.0, 1.0, 1.0, 1.0),
        "BBOX_NORMALIZE_TARGETS_PRECOMPUTED": True, "BBOX_NORMALIZE_MEANS": (0.0, 0.0, 0.0, 0.0),
        "BBOX_NORMALIZE_STDS": (0.1, 0.1, 0.2, 0.2), "PROPOSAL_METHOD": "gt", "HAS_RPN": True,
        "RPN_POSITIVE_OVERLAP": 0.7, "RPN_NEGATIVE_OVERLAP": 0.3, "RPN_CLOBBER_POSITIVES": False,
        "RPN_FG_FRACTION": 0.5, "RPN_BATCHSIZE": 256, "RPN_NMS_THRESH": 0.7, "RPN_PRE_NMS_TOP_N": 12000,
        "RPN_POST_NMS_TOP_N": 2000, "RPN_POST_NMS_TOP_N_TARGET": 128, "RPN_MIN_SIZE": 8,
        "RPN_BBOX_INSIDE_WEIGHTS": (1.0, 1.0, 1.0, 1.0), "RPN_POSITIVE_WEIGHT": -1.0, "USE_ALL_GT": True,
        "BN_TRAIN": False,
    },
    "TEST": {
        "SCALES": (600,), "MAX_SIZE": 1000, "NMS": 0.3, "SVM": False, "BBOX_REG": True, "HAS_RPN": False,
        "PROPOSAL_METHOD": "gt", "RPN_NMS_THRESH": 0.7, "RPN_PRE_NMS_TOP_N": 6000, "RPN_POST_NMS_TOP_N": 300,
        "RPN_MIN_SIZE": 16, "MODE": "nms", "RPN_TOP_N": 5000,
    },
    "RESNET": {"MAX_POOL": False, "FIXED_BLOCKS": 1},
    "MOBILENET": {"REGU_DEPTH": False, "FIXED_LAYERS": 5, "WEIGHT_DECAY": 0.00004, "DEPTH_MULTIPLIER": 1.0},
    "VGG_PATH": "./data/pretrained_model/vgg16_caffe.pth",
    "RESNET_PATH": "./data/pretrained_model/resnet101_caffe.pth",
    "RESNET_PATH50": "./data/pretrained_model/resnet50_caffe.pth",
    "DEDUP_BOXES": 1.0 / 16.0,
    "PIXEL_MEANS": np.array([[[102.9801, 115.9465, 122.7717]]]),
    "RNG_SEED": 3, "EPS": 1e-14, "ROOT_DIR": _ROOT, "DATA_DIR": osp.join(_ROOT, "data"), "MATLAB": "matlab",
    "EXP_DIR": "default", "USE_GPU_NMS": True, "GPU_ID": 0, "POOLING_MODE": "align", "POOLING_SIZE": 7,
    "MAX_NUM_GT_BOXES": 20, "ANCHOR_SCALES": [8, 16, 32], "ANCHOR_RATIOS": [0.5, 1, 2], "FEAT_STRIDE": [16],
    "CUDA": False, "CROP_RESIZE_WITH_MAX_POOL": True,
})
__C = cfg


def _merge(src, dst, path=""):
    for k, v in src.items():
        if k not in dst:
            raise KeyError("{} is not a valid config key".format(path + k))
        old = dst[k]
        if type(old) is not type(v) and not (isinstance(old, AttrDict) and isinstance(v, dict)):
            if isinstance(old, np.ndarray):
                v = np.array(v, dtype=old.dtype)
            else:
                raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(type(old), type(v), path + k))
        if isinstance(old, AttrDict):
            _merge(v, old, path + k + ".")
        else:
            dst[k] = v


def _merge_a_into_b(a, b):
    """Name kept for callers of the reference API (config.py:344)."""
    if isinstance(a, dict):
        _merge(a, b)


def cfg_from_file(filename):
    import yaml
    with open(filename, "r") as f:
        _merge(yaml.safe_load(f) or {}, cfg)


def cfg_from_list(cfg_list):
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        node = cfg
        *parents, leaf = k.split(".")
        for sub in parents:
            assert sub in node
            node = node[sub]
        assert leaf in node
        try:
            value = literal_eval(v) if isinstance(v, str) else v
        except Exception:
            value = v
        assert type(value) == type(node[leaf]), "type {} does not match original type {}".format(
            type(value), type(node[leaf]))
        node[leaf] = value


def default_cfg_file(net="res101"):
    return osp.join(osp.dirname(__file__), "..", "..", "cfgs", net + ".yml")
