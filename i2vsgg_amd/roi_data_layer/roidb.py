"""roidb enrichment and aspect-ratio ranking with the reference's API (lib/roi_data_layer/roidb.py).

``combined_roidb(imdb_names, training=True) -> (imdb, roidb, ratio_list, ratio_index)``.
Dataset classes (VOC-style XML readers, lib/datasets) are out of scope; an imdb is anything with
``name, classes, num_images, image_index, roidb, image_path_at(i), image_id_at(i)`` and optionally
``append_flipped_images()`` / ``set_proposal_method()``.  ``register_imdb`` plugs one in under a name;
``synthetic_<n>`` names create a VidVRD-shaped synthetic imdb (no dataset is reachable offline), ``synthetic_<n>_v`` one
whose frames differ in size, aspect ratio and number of boxes."""
import numpy as np
import scipy.sparse

from ..model.utils.config import cfg

_REGISTRY = {}


def register_imdb(name, factory):
    _REGISTRY[name] = factory


# native (height, width) of the frames of a ``synthetic_<n>_v`` imdb: five video resolutions, four aspect-ratio groups, one of
# them portrait -- after the 600-px rescale and the per-batch padding of roibatchLoader (:162-190) the minibatches come in
# several sizes, as they do on VidVRD / VidOR
VARIED_SIZES = ((480, 800), (360, 640), (480, 640), (720, 1280), (640, 480))


class SyntheticImdb:
    """n frames with seeded ground-truth boxes; pixels are generated on demand by the minibatch layer (entry key
    ``pixels_seed``) -- nothing is read from disk.  ``varied``: frame sizes cycle through ``VARIED_SIZES`` and the number of
    boxes per frame varies in [3, 12]; otherwise every frame is (height, width) with ``boxes_per_image`` boxes."""

    def __init__(self, name, n, num_classes=16, height=480, width=800, boxes_per_image=8, seed=0, varied=False):
        from .. import synthetic as syn
        self.name = name
        self.classes = tuple(["__background__"] + ["class%d" % i for i in range(1, num_classes)])
        self.num_classes = num_classes
        self.image_index = list(range(n))
        self._seed, self._varied = seed, varied
        self._sizes = []
        self.roidb = []
        rng = np.random.default_rng(seed)
        for i in range(n):
            h, w = VARIED_SIZES[i % len(VARIED_SIZES)] if varied else (height, width)
            nb = int(rng.integers(3, 13)) if varied else boxes_per_image
            self._sizes.append((h, w))
            bx = np.floor(syn.boxes(seed * 7919 + i, nb, h, w, 24, min(h, w) // 2))
            cls = rng.integers(1, num_classes, nb).astype(np.int32)
            ov = np.zeros((nb, num_classes), np.float32)
            ov[np.arange(nb), cls] = 1.0
            self.roidb.append({"boxes": bx.astype(np.uint16), "gt_classes": cls, "gt_ishard": np.zeros(nb, np.int32),
                               "gt_overlaps": scipy.sparse.csr_matrix(ov), "flipped": False,
                               "seg_areas": ((bx[:, 2] - bx[:, 0] + 1) * (bx[:, 3] - bx[:, 1] + 1)).astype(np.float32),
                               "pixels_seed": seed * 104729 + i})

    def gt_rels(self, n_rel=62):
        """Relation annotations in the layout of the reference's ``source_gt_rels`` pickle (faster_rcnn_SGG_emb.py:169-172):
        {image file name: {"boxes": unscaled pixel boxes, "box_classes": ..., "rels": [[s, o, predicate], ...]}} keyed by
        the last path component, which is what the training loop looks up (trainval_net_SGG_emb.py:217).  ``varied``:
        4-32 boxes and 2-32 annotated pairs per frame, else 32 + 32."""
        from .. import synthetic as syn
        rng = np.random.default_rng(self._seed + 31337)
        out = {}
        for i in range(len(self._sizes)):
            h, w = self._sizes[i]
            nb = int(rng.integers(4, 33)) if self._varied else 32
            npair = int(rng.integers(2, min(32, nb * (nb - 1)) + 1)) if self._varied else 32
            out[self.image_path_at(i).split("/")[-1]] = syn.relation_annotation(self._seed * 1000 + i, nb, npair, n_rel,
                                                                               self.num_classes, h, w)
        return out

    @property
    def num_images(self):
        return len(self.image_index)

    def image_path_at(self, i):
        # by image INDEX, as imdb.image_path_at does: the flipped copy of a frame (image_index is doubled) has its frame's path
        return "synthetic://%s/%06d" % (self.name, self.image_index[i])

    def image_id_at(self, i):
        return i

    def image_size_at(self, i):
        h, w = self._sizes[i % len(self._sizes)]
        return w, h

    def set_proposal_method(self, method):
        assert method == "gt"

    def append_flipped_images(self):
        n = self.num_images
        for i in range(n):
            e = self.roidb[i]
            boxes = e["boxes"].copy()
            x1, x2 = boxes[:, 0].copy(), boxes[:, 2].copy()
            width = self._sizes[i][1]
            boxes[:, 0] = width - x2 - 1            # imdb.append_flipped_images (datasets/imdb.py)
            boxes[:, 2] = width - x1 - 1
            f = dict(e, boxes=boxes, flipped=True)
            self.roidb.append(f)
        self.image_index = self.image_index * 2


def get_imdb(name):
    if name in _REGISTRY:
        return _REGISTRY[name]()
    if name.startswith("synthetic_"):              # synthetic_<n> (one size, 8 boxes) or synthetic_<n>_v[_<seed>] (varied sizes / counts)
        parts = name.split("_")
        return SyntheticImdb(name, int(parts[1]), varied=len(parts) > 2 and parts[2] == "v",
                             seed=int(parts[3]) if len(parts) > 3 else 0)
    raise KeyError("Unknown dataset: %s (register it with roi_data_layer.roidb.register_imdb)" % name)


def prepare_roidb(imdb):
    """roidb.py:17-49: add img_id, image, width, height, max_classes, max_overlaps."""
    roidb = imdb.roidb
    for i in range(len(imdb.image_index)):
        e = roidb[i]
        e["img_id"] = imdb.image_id_at(i)
        e["image"] = imdb.image_path_at(i)
        if hasattr(imdb, "image_size_at"):
            e["width"], e["height"] = imdb.image_size_at(i)
        else:
            import PIL.Image
            e["width"], e["height"] = PIL.Image.open(e["image"]).size
        ov = e["gt_overlaps"].toarray()
        e["max_overlaps"] = ov.max(axis=1)
        e["max_classes"] = ov.argmax(axis=1)
        assert all(e["max_classes"][e["max_overlaps"] == 0] == 0)
        assert all(e["max_classes"][e["max_overlaps"] > 0] != 0)


def rank_roidb_ratio(roidb, ratio_large=2.0, ratio_small=0.5):
    """roidb.py:52-84: clamp aspect ratios to [0.5, 2] (marking need_crop) and sort ascending."""
    ratios = []
    for e in roidb:
        r = e["width"] / float(e["height"])
        e["need_crop"] = int(r > ratio_large or r < ratio_small)
        ratios.append(min(max(r, ratio_small), ratio_large))
    ratios = np.array(ratios)
    index = np.argsort(ratios)
    return ratios[index], index


def filter_roidb(roidb):
    return [e for e in roidb if len(e["boxes"]) > 0]


def combined_roidb(imdb_names, training=True):
    def load(name):
        imdb = get_imdb(name)
        if hasattr(imdb, "set_proposal_method"):
            imdb.set_proposal_method(cfg.TRAIN.PROPOSAL_METHOD)
        if cfg.TRAIN.USE_FLIPPED and hasattr(imdb, "append_flipped_images"):
            imdb.append_flipped_images()
        prepare_roidb(imdb)
        return imdb

    imdbs = [load(s) for s in imdb_names.split("+")]
    roidb = imdbs[0].roidb
    for other in imdbs[1:]:
        roidb.extend(other.roidb)
    imdb = imdbs[0] if len(imdbs) == 1 else imdbs[1]
    ratio_list, ratio_index = rank_roidb_ratio(roidb)
    return imdb, roidb, ratio_list, ratio_index
