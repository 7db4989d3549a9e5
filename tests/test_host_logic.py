"""CPU tests of the host-side logic: config surface, roi_data_layer API, frame sharding, and the
world_size-2 gloo rehearsal of the gradient exchange."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cfg_surface_matches_reference_yaml_and_rules():
    from i2vsgg_amd.model.utils import config as c
    c.cfg_from_file(c.default_cfg_file("res101"))
    assert c.cfg.EXP_DIR == "res101" and c.cfg.TRAIN.BG_THRESH_LO == 0.0 and c.cfg.TRAIN.RPN_BATCHSIZE == 256
    c.cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30"])
    assert c.cfg.MAX_NUM_GT_BOXES == 30
    with pytest.raises(KeyError):
        c._merge_a_into_b({"NOT_A_KEY": 1}, c.cfg)
    with pytest.raises(ValueError):
        c._merge_a_into_b({"TRAIN": {"BATCH_SIZE": "many"}}, c.cfg)
    with pytest.raises(AssertionError):
        c.cfg_from_list(["TRAIN.BATCH_SIZE", "1.5"])
    assert c.cfg.TRAIN.RPN_PRE_NMS_TOP_N == 12000 and c.cfg.TEST.RPN_POST_NMS_TOP_N == 300


def test_generate_anchors_matches_reference_golden(gold):
    from i2vsgg_amd.model.rpn.generate_anchors import generate_anchors, shifted_anchors
    g = gold("anchors")
    base = generate_anchors(scales=np.array([8, 16, 32]), ratios=np.array([0.5, 1, 2]))
    assert np.array_equal(base, g["base"])
    assert np.array_equal(shifted_anchors(38, 63, 16, base), g["grid_38x63"])


def test_roi_data_layer_api():
    from i2vsgg_amd.model.utils import config as c
    from i2vsgg_amd.model.utils.net_utils import sampler
    from i2vsgg_amd.roi_data_layer.roidb import combined_roidb
    from i2vsgg_amd.roi_data_layer.roibatchLoader import roibatchLoader
    c.cfg.TRAIN.USE_FLIPPED = True
    imdb, roidb, ratio_list, ratio_index = combined_roidb("synthetic_6")
    assert len(roidb) == 12 and imdb.num_classes == 16
    for key in ("boxes", "gt_classes", "gt_overlaps", "flipped", "img_id", "image", "width", "height", "max_classes",
                "max_overlaps", "need_crop"):
        assert key in roidb[0]
    assert roidb[0]["boxes"].dtype == np.uint16 and roidb[6]["flipped"]
    assert np.all(np.diff(ratio_list) >= 0)
    ds = roibatchLoader(roidb, ratio_list, ratio_index, 2, imdb.num_classes, training=True)
    np.random.seed(3)
    data, im_info, gt, n = ds[0]
    assert data.dim() == 3 and data.size(0) == 3 and im_info.shape == (3,)
    assert min(data.shape[1:]) >= 600 and gt.shape == (c.cfg.MAX_NUM_GT_BOXES, 5) and n == 8
    assert float(im_info[0]) == data.size(1) and float(im_info[1]) == data.size(2)
    assert torch.all(gt[n:] == 0) and torch.all(gt[:n, 4] >= 1)
    test = roibatchLoader(roidb, ratio_list, ratio_index, 1, imdb.num_classes, training=False)
    d, info, g, nb, path = test[1]
    assert nb == 0 and g.tolist() == [1, 1, 1, 1, 1] and path.startswith("synthetic://")
    roidb[int(ratio_index[0])]["need_crop"] = 1
    assert isinstance(ds[0], torch.Tensor) and ds[0].numel() == 3        # bare im_info, as the reference returns
    idx = list(iter(sampler(10, 4)))
    assert sorted(int(i) for i in idx) == list(range(10)) and [int(i) for i in idx[-2:]] == [8, 9]


def test_varied_synthetic_imdb_feeds_batches_of_several_sizes():
    """``synthetic_<n>_v``: frames in five resolutions / four aspect-ratio groups with 3-12 boxes each; the loader pads every
    minibatch to its own aspect ratio (roibatchLoader.py:162-190), so a training loop sees several (H, W); relation
    annotations are keyed by the last path component, as trainval_net_SGG_emb.py:217 looks them up."""
    from i2vsgg_amd.model.utils import config as c
    from i2vsgg_amd.model.utils.net_utils import sampler
    from i2vsgg_amd.roi_data_layer.roidb import combined_roidb
    from i2vsgg_amd.roi_data_layer.roibatchLoader import roibatchLoader
    saved = (c.cfg.TRAIN.USE_FLIPPED, c.cfg.TRAIN.SCALES)
    c.cfg.TRAIN.USE_FLIPPED, c.cfg.TRAIN.SCALES = False, (96,)
    try:
        imdb, roidb, ratio_list, ratio_index = combined_roidb("synthetic_20_v")
        assert len({(e["height"], e["width"]) for e in roidb}) == 5 and len({len(e["boxes"]) for e in roidb}) >= 4
        assert np.all(np.diff(ratio_list) >= 0) and ratio_list[0] < 1 < ratio_list[-1]
        ds = roibatchLoader(roidb, ratio_list, ratio_index, 2, imdb.num_classes, training=True, path_return=True)
        dl = torch.utils.data.DataLoader(ds, batch_size=2, sampler=sampler(len(roidb), 2, seed=3))
        np.random.seed(1)
        batches = list(dl)
        assert len(batches) == 10 and all(len(b) == 5 for b in batches)
        sizes = {tuple(b[0].shape[2:]) for b in batches}
        assert len(sizes) >= 3 and all(min(s) == 96 for s in sizes)
        for b in batches:                       # im_info carries the PADDED size (roibatchLoader.py:168,180)
            assert b[1][:, 0].tolist() == [b[0].shape[2]] * 2 and b[1][:, 1].tolist() == [b[0].shape[3]] * 2
        rels = imdb.gt_rels(62)
        assert set(rels) == {p.split("/")[-1] for b in batches for p in b[4]}
        assert len({len(a["boxes"]) for a in rels.values()}) >= 5
        for a in rels.values():
            assert all(0 <= s < len(a["boxes"]) and 0 <= o < len(a["boxes"]) and s != o and 0 <= r < 62 for s, o, r in a["rels"])
        other = combined_roidb("synthetic_20_v_7")[1]
        assert not np.array_equal(other[0]["boxes"], roidb[0]["boxes"])
    finally:
        c.cfg.TRAIN.USE_FLIPPED, c.cfg.TRAIN.SCALES = saved


def test_device_prep_loader_items_carry_the_host_items_metadata():
    """roibatchLoader(device_prep=True): the item is the uint8 frame as decoded + [flipped, canvas_h, canvas_w, scale, target];
    gt_boxes, num_boxes, paths, the scale and the padded canvas equal what the host form (get_minibatch + the loader's padding,
    roibatchLoader.py:162-190) returns for the same index and np.random state; the square-trim case is handed back
    (canvas 0); collate_device_prep keeps the frames a list."""
    from i2vsgg_amd.model.utils import config as c
    from i2vsgg_amd.roi_data_layer.roidb import combined_roidb
    from i2vsgg_amd.roi_data_layer.roibatchLoader import collate_device_prep, roibatchLoader
    saved = (c.cfg.TRAIN.USE_FLIPPED, c.cfg.TRAIN.SCALES)
    c.cfg.TRAIN.USE_FLIPPED, c.cfg.TRAIN.SCALES = True, (96,)
    try:
        imdb, roidb, rl, ri = combined_roidb("synthetic_10_v")
        host = roibatchLoader(roidb, rl, ri, 2, imdb.num_classes, training=True, path_return=True)
        dev = roibatchLoader(roidb, rl, ri, 2, imdb.num_classes, training=True, path_return=True, device_prep=True)
        trims = flips = 0
        for i in range(len(roidb)):
            np.random.seed(i); x = host[i]
            np.random.seed(i); y = dev[i]
            assert y[0].dtype == torch.uint8 and y[0].dim() == 3 and y[0].shape[2] == 3
            e = roidb[int(ri[i])]
            assert tuple(y[0].shape[:2]) == (e["height"], e["width"]) and bool(y[1][0]) == bool(e["flipped"])
            flips += int(y[1][0])
            assert torch.equal(x[2], y[2]) and x[3] == y[3] and x[4] == y[4]
            assert abs(float(x[1][2]) - float(y[1][3])) < 1e-7 and int(y[1][4]) == 96
            if int(y[1][1]) == 0:
                trims += 1
                continue
            assert tuple(x[0].shape[1:]) == (int(y[1][1]), int(y[1][2])) == (int(x[1][0]), int(x[1][1]))
        assert flips == 10 and trims <= 4
        b = collate_device_prep([dev[0], dev[1]])
        assert isinstance(b[0], list) and len(b[0]) == 2 and b[1].shape == (2, 5) and b[2].shape == (2, c.cfg.MAX_NUM_GT_BOXES, 5)
        assert len(b[4]) == 2
        # flipped copies carry their frame's path: the relation annotations of a frame serve both (imdb.image_path_at semantics)
        assert {p.split("/")[-1] for p in (e["image"] for e in roidb)} == set(imdb.gt_rels(62))
    finally:
        c.cfg.TRAIN.USE_FLIPPED, c.cfg.TRAIN.SCALES = saved


def test_device_prep_loader_items_in_test_mode():
    """roibatchLoader(training=False, device_prep=True): the frame alone -- uint8 as decoded, meta = [0, resized height,
    resized width, scale, target], the host form's placeholder box, 0, the path -- with the im_info of the host form's item."""
    from i2vsgg_amd.model.utils import config as c
    from i2vsgg_amd.roi_data_layer.roidb import combined_roidb
    from i2vsgg_amd.roi_data_layer.roibatchLoader import roibatchLoader
    saved = (c.cfg.TRAIN.USE_FLIPPED, c.cfg.TRAIN.SCALES)
    c.cfg.TRAIN.USE_FLIPPED, c.cfg.TRAIN.SCALES = False, (96,)
    try:
        imdb, roidb, rl, ri = combined_roidb("synthetic_6_v", False)
        host = roibatchLoader(roidb, rl, ri, 1, imdb.num_classes, training=False, normalize=False)
        dev = roibatchLoader(roidb, rl, ri, 1, imdb.num_classes, training=False, normalize=False, device_prep=True)
        sizes = set()
        for i in range(len(roidb)):
            x, y = host[i], dev[i]
            assert y[0].dtype == torch.uint8 and tuple(y[0].shape) == (roidb[i]["height"], roidb[i]["width"], 3)
            assert tuple(x[0].shape[1:]) == (int(y[1][1]), int(y[1][2])) == (int(x[1][0]), int(x[1][1]))
            assert float(y[1][0]) == 0 and int(y[1][4]) == 96 and np.float32(float(y[1][3])) == np.float32(float(x[1][2]))
            assert torch.equal(x[2], y[2]) and x[3] == y[3] == 0 and x[4] == y[4]
            sizes.add(tuple(x[0].shape[1:]))
        assert len(sizes) >= 2
    finally:
        c.cfg.TRAIN.USE_FLIPPED, c.cfg.TRAIN.SCALES = saved


def test_rank_sharded_sampler_partitions_the_block_order():
    """One process per GPU: every rank draws the same block order and keeps every world-th block -- disjoint, equally many
    minibatches per rank, blocks still contiguous (one aspect-ratio group per minibatch), a new order every epoch."""
    from i2vsgg_amd.model.utils.net_utils import sampler
    samplers = [sampler(23, 2, rank=r, world=3, seed=5) for r in range(3)]
    first = [[int(i) for i in s] for s in samplers]
    assert all(len(x) == len(samplers[0]) == 6 for x in first)
    flat = sorted(i for x in first for i in x)
    assert len(set(flat)) == 18 and all(b == a + 1 for x in first for a, b in zip(x[0::2], x[1::2])) and all(a % 2 == 0 for x in first for a in x[0::2])
    second = [[int(i) for i in s] for s in samplers]
    assert second != first and len({i for x in second for i in x}) == 18
    ref = sampler(10, 4)                         # the reference's form: every index once, leftovers last
    idx = [int(i) for i in ref]
    assert sorted(idx) == list(range(10)) and idx[-2:] == [8, 9] and len(ref) == 10


def test_sgg_head_inputs_of_a_minibatch():
    """train.sgg_head_inputs: faster_rcnn_SGG_emb.py:170-245 for every frame of a minibatch -- rows of frame f carry f in
    column 0, pair indices point into the concatenated box table, a frame without an annotated relation contributes nothing
    (:177-183), the row weights make sum_r w[r] * mean_c BCE the mean over frames of the per-frame mean; the host
    rasteriser equals the device one."""
    from i2vsgg_amd import synthetic as syn, train
    from i2vsgg_amd.model.faster_rcnn.faster_rcnn_SGG_emb import build_pair_tables, rasterize_masks
    a0 = syn.relation_annotation(1, 7, 5, 62, 16, 360, 640)
    a1 = {"boxes": [[1, 2, 30, 40]], "box_classes": [3], "rels": []}
    a2 = syn.relation_annotation(2, 12, 9, 62, 16, 360, 640)
    info = np.array([[192, 342, 0.5333], [192, 342, 0.5333], [192, 342, 0.5333]], np.float32)
    f = train.sgg_head_inputs([a0, a1, a2], info, 62)
    assert f["boxes"].shape == (19, 5) and f["relb"].shape == (14, 5) and f["labels"].shape == (14, 62)
    assert f["boxes"][:7, 0].tolist() == [0] * 7 and f["boxes"][7:, 0].tolist() == [2] * 12
    assert f["relb"][:5, 0].tolist() == [0] * 5 and f["relb"][5:, 0].tolist() == [2] * 9
    assert f["ixs"][:5].max() < 7 and f["ixs"][5:].min() >= 7 and f["ixo"][5:].max() < 19
    np.testing.assert_allclose(f["wrow"], [1 / 10.0] * 5 + [1 / 18.0] * 9, rtol=1e-6)
    gt, union, bnd, lab, s, o = build_pair_tables(a2, float(info[2][2]), 192.0, 342.0, 62)
    assert np.array_equal(f["boxes"][7:, 1:], gt.astype(np.float32)) and np.array_equal(f["relb"][5:, 1:], union.astype(np.float32))
    assert np.array_equal(f["ixs"][5:], s + 7) and np.array_equal(f["labels"][5:], lab) and np.array_equal(f["bounds"][5:], bnd)
    m = train._rasterize_host(f["bounds"])
    assert m.shape == (14, 4, 32, 32) and np.array_equal(m[:, :2], rasterize_masks(f["bounds"], "cpu").numpy()) and not m[:, 2:].any()
    assert train.sgg_head_inputs([a1, None], info, 62) is None


def test_shard_frames_is_a_partition():
    from i2vsgg_amd.parallel import shard_frames
    for n, w in ((16, 8), (32, 8), (5, 2), (3, 4)):
        spans = [shard_frames(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under i2vsgg_amd/ or bench.py's GPU path may use it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "i2vsgg_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                if "import oracle" in text or "from oracle" in text or "oracle/" in text:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from i2vsgg_amd import parallel
rk, world, dev = parallel.init_from_env("gloo")
assert world == 2 and dev.type == "cpu"
torch.manual_seed(0)
lin = torch.nn.Sequential(torch.nn.Linear(300, 2000), torch.nn.ReLU(), torch.nn.Linear(2000, 7))   # one >1 MiB tensor, small ones
x = torch.randn(8, 300)
y = torch.randn(8, 7)
lo, hi = parallel.shard_frames(8, rk, world)
loss = torch.nn.functional.mse_loss(lin(x[lo:hi]), y[lo:hi])          # per-rank mean over its 4 frames
(loss / world).backward()
token = parallel.all_reduce_grads_start(list(lin.parameters()))       # the split form the pipelined step uses
overlapped = (x * 2).sum()                                            # independent work between start and finish
parallel.all_reduce_grads_finish(token)
ref = torch.nn.Sequential(torch.nn.Linear(300, 2000), torch.nn.ReLU(), torch.nn.Linear(2000, 7))
ref.load_state_dict(lin.state_dict())
torch.nn.functional.mse_loss(ref(x), y).backward()                     # single-process, whole batch
for a, b in zip(lin.parameters(), ref.parameters()):
    assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-7), (a.grad - b.grad).abs().max()
m = parallel.max_over_ranks(float(rk + 1), dev)
assert m == 2.0
# a step in which only rank 0 has a loss (trainval_sgg_emb.train_variant: the reference's `continue` is single-process):
# every rank learns it from one all-reduced count, rank 1 takes part in the exchange with zero gradients
for q in list(lin.parameters()) + list(ref.parameters()):
    q.grad = None
have = rk == 0
assert parallel.any_rank(have) is True
if have:
    (torch.nn.functional.mse_loss(lin(x[lo:hi]), y[lo:hi]) / world).backward()
parallel.all_reduce_grads(list(lin.parameters()))
(torch.nn.functional.mse_loss(ref(x[0:4]), y[0:4]) / world).backward()
for a, b in zip(lin.parameters(), ref.parameters()):
    assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-7), (a.grad - b.grad).abs().max()
assert parallel.any_rank(False) is False
parallel.barrier()
dist.destroy_process_group()
print("rank", rk, "ok")
"""


def test_gradient_exchange_world_size_2_gloo(tmp_path):
    """2 ranks, frames sharded, loss/world, one sum all-reduce == the single-process whole-batch gradient."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


BUCKET_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from i2vsgg_amd import parallel
rk, world, dev = parallel.init_from_env("gloo")
assert world == 2 and dev.type == "cpu"
torch.manual_seed(0)
L = lambda: torch.nn.Linear(16, 16)
def build():
    net = torch.nn.Module()             # the detector's parameter names, toy layers
    net.RCNN_base = torch.nn.Sequential(*[torch.nn.Identity() for _ in range(4)], L(), L(), torch.nn.Sequential(*[L() for _ in range(7)]))
    net.RCNN_top, net.RCNN_rpn, net.netD_style, net.netD_pixel = L(), L(), L(), L()
    return net
def losses(net, x, target):
    f1 = torch.relu(net.RCNN_base[5](torch.relu(net.RCNN_base[4](x))))
    d_style = net.netD_style(f1)
    f = f1
    for blk in net.RCNN_base[6]:
        f = torch.relu(blk(f)) + f
    d_pix = net.netD_pixel(f)
    if target:                          # the target pass trains the discriminators only (no detection losses)
        return ((1 - d_style) ** 2).mean() + ((1 - d_pix) ** 2).mean()
    return (d_style ** 2).mean() + (d_pix ** 2).mean() + net.RCNN_top(f).pow(2).mean() + net.RCNN_rpn(f).abs().mean()
net, ref = build(), build()
ref.load_state_dict(net.state_dict())
names = [n for n, _ in net.named_parameters()]
params = [p for _, p in net.named_parameters()]
ids, nb = parallel.detector_buckets(names)
assert nb == 5 and ids[names.index("RCNN_top.weight")] == 0 and ids[names.index("RCNN_rpn.bias")] == 0
assert ids[names.index("RCNN_base.6.6.weight")] == 1 and ids[names.index("RCNN_base.6.0.weight")] == 3      # the last block first
assert ids[names.index("RCNN_base.5.weight")] == 4 and ids[names.index("netD_style.weight")] == 4
buckets = [[i for i, b in enumerate(ids) if b == k] for k in range(nb)]
state = {"m": None}
for i, p in enumerate(params):
    p.register_hook(lambda g, k=ids[i]: state["m"].hit(k))
xs, xt = torch.randn(8, 16), torch.randn(8, 16)
lo, hi = parallel.shard_frames(8, rk, world)
grads, marks = [], []
for x, target in ((xs, False), (xt, True)):                # the step's two branches: source frames, target frames
    state["m"] = m = parallel.BucketMarks(False)
    grads.append(torch.autograd.grad(losses(net, x[lo:hi], target) / world, params, allow_unused=True))
    marks.append(m)
assert marks[0].order == [0, 1, 2, 3, 4], marks[0].order       # the buckets complete in backward order ...
assert marks[1].order[0] == 0 and 0 in marks[1].events         # ... also on the branch without detection losses (netD_pixel)
assert grads[1][names.index("RCNN_top.weight")] is None        # the target pass has no gradient for the detection heads
tokens = parallel.exchange_in_buckets(params, buckets, grads, marks)
parallel.finish_buckets(tokens)
(losses(ref, xs, False) + losses(ref, xt, True)).backward()     # single process, every frame
for n, a, b in zip(names, params, ref.parameters()):
    assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-7), (n, (a.grad - b.grad).abs().max())
parallel.barrier()
dist.destroy_process_group()
print("rank", rk, "ok")
"""


def test_bucketed_detector_exchange_world_size_2_gloo(tmp_path):
    """Round 6: the detector step's exchange in buckets (parallel.detector_buckets / BucketMarks / exchange_in_buckets), two
    ranks x two branches (source frames, target frames; the target branch has no gradient for the detection heads): the
    buckets complete in backward order on both branches, and summed per bucket over the branches and all-reduced bucket by
    bucket the gradients equal the single-process gradient of every frame."""
    script = tmp_path / "worker.py"
    script.write_text(BUCKET_WORKER % ROOT)
    port = 31500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


def test_host_front_end_equals_oracle_restatement():
    """roi_data_layer.minibatch.prep_im_for_blob (product, host) == oracle.data (checker) bit for bit."""
    from i2vsgg_amd.roi_data_layer import minibatch as mb
    from oracle import data as odata
    rng = np.random.default_rng(1)
    means = np.array([[[102.9801, 115.9465, 122.7717]]])
    for (h, w), target in (((37, 53), 60), ((375, 500), 600), ((60, 40), 30)):
        u8 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        a, sa = mb.prep_im_for_blob(u8[:, :, ::-1], means, target)
        b, sb = odata.minibatch_image(u8, means.reshape(-1), target)
        assert sa == sb and a.shape == b.shape and np.array_equal(a, b)


TP_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import torch.nn.functional as F
from i2vsgg_amd import parallel
rk, world, dev = parallel.init_from_env("gloo")
assert world == 2
R, K, C = 5, 24, 8                        # rows per rank, input features, output features (C %% world == 0)
g = torch.Generator().manual_seed(0)
W = torch.randn(C, K, generator=g); b = torch.randn(C, generator=g)
xs = [torch.randn(R, K, generator=g) for _ in range(world)]
ts = [torch.randn(R, C, generator=g) for _ in range(world)]
# ---- data parallel reference: full layer on every rank, gradient all-reduced
Wd, bd = W.clone().requires_grad_(), b.clone().requires_grad_()
h_dp = F.relu(F.linear(xs[rk], Wd, bd))
((h_dp * ts[rk]).sum() / world).backward()
dist.all_reduce(Wd.grad); dist.all_reduce(bd.grad)
# ---- column-parallel: my output columns for everybody's rows; nothing of W crosses the ranks
n = C // world
Ws = parallel.mark_local(W[rk * n:(rk + 1) * n].clone().requires_grad_())
bs = parallel.mark_local(b[rk * n:(rk + 1) * n].clone().requires_grad_())
x_all = parallel.gather_rows(xs[rk])
assert torch.equal(x_all, torch.cat(xs))
h = parallel.ColShardToOwnRows.apply(F.relu(F.linear(x_all, Ws, bs)))
assert torch.allclose(h, h_dp.detach(), rtol=1e-6, atol=1e-6)
((h * ts[rk]).sum() / world).backward()
assert torch.allclose(Ws.grad, Wd.grad[rk * n:(rk + 1) * n], rtol=1e-5, atol=1e-6), (Ws.grad - Wd.grad[rk * n:(rk + 1) * n]).abs().max()
assert torch.allclose(bs.grad, bd.grad[rk * n:(rk + 1) * n], rtol=1e-5, atol=1e-6)
# local shards are exempt from the exchange
other = torch.nn.Parameter(torch.ones(3)); other.grad = torch.full((3,), float(rk + 1))
before = Ws.grad.clone()
parallel.all_reduce_grads([Ws, bs, other])
assert torch.equal(Ws.grad, before) and torch.equal(other.grad, torch.full((3,), 3.0))
parallel.barrier()
dist.destroy_process_group()
print("rank", rk, "ok")
"""


def test_column_parallel_fc6_equals_data_parallel_world_size_2_gloo(tmp_path):
    """vrd.fc6 cut by output columns (parallel.gather_rows + ColShardToOwnRows): same activations and the same weight
    gradient as the data-parallel layer with an all-reduce -- without moving the weight gradient."""
    script = tmp_path / "tp_worker.py"
    script.write_text(TP_WORKER % ROOT)
    port = 31500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


def _run_bench(args, env=None, timeout=180):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, **(env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, env=e,
                         timeout=timeout)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out.returncode, [json.loads(l) for l in lines], out.stderr


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without torchrun starts N ranks itself (the role of nn.DataParallel behind --mGPUs,
    trainval_net_SGG_emb.py:175-176): rehearsed on the CPU with gloo -- rendezvous on 127.0.0.1, one collective over
    both ranks, ONE JSON line from rank 0 carrying the world size."""
    rc, lines, err = _run_bench(["--gpus", "2", "--dry-run", "--config", "sgg"])
    assert rc == 0, err
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["dry_run"] is True


def test_bench_fails_when_a_rank_does_not_come_up():
    rc, lines, err = _run_bench(["--gpus", "2", "--dry-run"], env={"I2V_BENCH_FAIL_RANK": "1"})
    assert rc != 0 and not lines
    assert "rank exited" in err


def test_block_backward_protocol_wiring():
    """The pre-masked gradient hand-over of ops._BottleneckFn is only wired where a block's output has exactly one
    consumer: inside a layer and from layer1 into layer2 -- not across the layer2 tap (netD_style reads it),
    not out of layer3 (RPN, ROI pooling), not into layer4 (its input is ROI-pooled, not a ReLU output)."""
    from i2vsgg_amd.model.faster_rcnn.layers import C4Base, make_layer
    base = C4Base((3, 4, 23))
    l1, l2, l3 = base[4], base[5], base[6]
    for layer in (l1, l2, l3):
        for i, blk in enumerate(layer):
            if i:
                assert blk.in_relu and layer[i - 1]._next[0] is blk
    assert not l1[0].in_relu                              # behind the frozen stem: no gradient leaves the block anyway
    assert l2[0].in_relu and l1[-1]._next[0] is l2[0]     # layer1 -> layer2: one consumer
    assert not l3[0].in_relu and l2[-1]._next == []       # the tap
    assert l3[-1]._next == []
    layer4, _ = make_layer(1024, 512, 3, 2)
    assert not layer4[0].in_relu and layer4[1].in_relu and layer4[-1]._next == []
    # no submodule was registered by the wiring: the state_dict keys stay the reference's
    assert not any("_next" in k for k in base.state_dict())


def test_stage_pipeline_bookkeeping_trains_every_minibatch_once():
    """train._StagePipeline, the host-side twin of SGGEmbStep's three head-input slots (round 6: the backbone beside the head cut
    by stage, a minibatch reaches the head two calls after its stage()): a loop that stages before every call trains every
    minibatch exactly once, in order, with ONE call that runs no head (its second); a resident minibatch is trained by every call;
    a minibatch staged into a pipeline of trained ones arrives after two bubbles; ``train.run_staged`` drives any of it."""
    import torch
    from i2vsgg_amd import train

    class Step:                                  # SGGEmbStep's calling protocol around the real bookkeeping object
        def __init__(self):
            self.pipe, self.trained, self.n_bubbles = train._StagePipeline(), [], 0
            self.pipe.staged()                   # the first minibatch, staged before capture()
            self.pipe.prime()

        bubble = property(lambda self: self.pipe.bubble)

        def stage(self):
            self.pipe.staged()

        def __call__(self):
            head = self.pipe.call()
            if head is None:
                self.n_bubbles += 1
                return torch.tensor(-1.0)
            self.trained.append(head)
            return torch.tensor(float(head))

    st = Step()                                  # (a) resident: every call trains minibatch 1, none is a bubble
    for _ in range(5):
        assert not st.bubble
        st()
    assert st.trained == [1] * 5 and st.n_bubbles == 0
    st.stage()                                   # (b) a new minibatch into a pipeline of trained ones: two bubbles, then it trains
    assert [st.bubble, float(st()), st.bubble, float(st()), st.bubble, float(st())] == [True, -1.0, True, -1.0, False, 2.0]
    st = Step()                                  # (c) the staging loop
    keep = torch.zeros(7)
    train.run_staged(st, [st.stage] * 6, keep)
    assert st.trained == [1, 2, 3, 4, 5, 6, 7] and keep.tolist() == [1, 2, 3, 4, 5, 6, 7] and st.n_bubbles == 1
    for _ in range(3):                           # ... and the last minibatch stays resident behind it
        assert not st.bubble
        st()
    assert st.trained[-3:] == [7, 7, 7]
