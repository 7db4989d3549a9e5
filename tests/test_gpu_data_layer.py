"""The training steps behind the reference's data path: ``combined_roidb -> roibatchLoader -> DataLoader(sampler)``
(trainval_net_SGG_emb.py:77-91,204-217; trainval_net_instance_styleD_bilinear.py:73-97,236-291) with minibatches that differ
in image size (the loader pads every batch to its own aspect ratio, roibatchLoader.py:162-190) and in boxes / pairs per
frame -- through the CAPTURED steps (one HIP graph per frame size, boxes / pairs padded to a capacity), against the eager
un-padded step on the same batches."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
SET = ["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30", "TRAIN.SCALES", "(192,)"]


@pytest.fixture()
def small_cfg():
    """cfgs/res101.yml at a 192-px shorter side; the global cfg singleton is put back afterwards."""
    from i2vsgg_amd.model.utils import config as c
    saved = copy.deepcopy(dict(c.cfg))

    def load(extra=()):
        c.cfg_from_file(c.default_cfg_file("res101"))
        c.cfg_from_list(SET + list(extra))
        return c.cfg
    yield load
    c._merge_a_into_b(c.AttrDict(saved), c.cfg)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _loader(name, bs, seed, flipped=False, paths=True):
    from i2vsgg_amd.model.utils import config as c
    from i2vsgg_amd.model.utils.net_utils import sampler
    from i2vsgg_amd.roi_data_layer.roibatchLoader import roibatchLoader
    from i2vsgg_amd.roi_data_layer.roidb import combined_roidb
    c.cfg.TRAIN.USE_FLIPPED = flipped
    imdb, roidb, ratio_list, ratio_index = combined_roidb(name)
    ds = roibatchLoader(roidb, ratio_list, ratio_index, bs, imdb.num_classes, training=True, path_return=paths)
    dl = torch.utils.data.DataLoader(ds, batch_size=bs, sampler=sampler(len(roidb), bs, seed=seed), pin_memory=True)
    return imdb, dl


def test_roi_pool_with_device_side_extent_equals_the_static_kernel():
    """i2v_roi_pool_fwd_geom (H, W read from device memory, maps packed at the start of a larger buffer) == i2v_roi_pool_fwd
    for every extent that fits; an extent that does not fit yields zeros instead of reading out of bounds."""
    from i2vsgg_amd import ops, synthetic as syn
    rng = np.random.default_rng(2)
    cap = 16 * 24
    for C in (1024, 96):
        buf = torch.zeros(2 * cap * C, device=DEV)
        for (h, w) in ((12, 22), (16, 12), (16, 24), (5, 7)):
            fm = torch.from_numpy(np.abs(rng.standard_normal((2, C, h, w), dtype=np.float32))).to(DEV) \
                .contiguous(memory_format=torch.channels_last)
            rois = np.zeros((9, 5), np.float32)
            rois[:, 0] = rng.integers(0, 2, 9)
            rois[:, 1:] = syn.boxes(h * w, 9, h * 16, w * 16, 8, 200)
            rois[0, 1:] = (0, 0, 0, 0)                                  # the pad row of a capacity-padded batch
            rois_d = torch.from_numpy(rois).to(DEV)
            buf.zero_()
            buf[:fm.numel()].copy_(fm.permute(0, 2, 3, 1).reshape(-1))
            geom = torch.tensor([h, w], dtype=torch.int32, device=DEV)
            maps = ops.PackedMaps(buf, 2, C, geom)
            for nchw in (True, False):
                want = ops.roi_pool(fm, rois_d, 7, 7, 1 / 16.0, out_nchw=nchw)
                got = ops.roi_pool_packed(maps, rois_d, 7, 7, 1 / 16.0, out_nchw=nchw)
                assert torch.equal(got, want), (C, h, w, nchw)
            assert torch.equal(maps.view(h, w), fm)
        big = torch.tensor([40, 40], dtype=torch.int32, device=DEV)
        assert float(ops.roi_pool_packed(ops.PackedMaps(buf, 2, C, big), rois_d, 7, 7, 1 / 16.0).abs().max()) == 0.0


@pytest.mark.parametrize("copy_stream", [False, True])
def test_sgg_captured_step_consumes_loader_batches_of_varying_size(small_cfg, monkeypatch, copy_stream):
    """``copy_stream``: the loop with the frames crossing PCIe on the copy stream (the default since round 4) / on the caller's
    stream (I2V_UPLOAD_STREAM=0).  Round 3 met a host segfault in hipGraphLaunch in exactly this test, in exactly this file order
    (tests/test_gpu_configs.py first), with the copy stream on, and switched it off by default.  Since then (a) no graph is dropped while a replay of it may still be running
    (capacity growth in stage() dropped every graph right behind an asynchronous replay -- ``invalidate_graphs`` synchronises
    first now) and (b) the copy stream is a stream of its own (ops.role_stream) instead of the next of torch's 32 pooled handles.
    DESIGN.md 5.5 has the analysis.

    SGG_emb: 8 loader minibatches of 2 frames in >= 3 sizes, 4-32 boxes and 2-32 pairs per frame, through (a) the eager
    un-padded step and (b) the captured, overlapped step (capacity-padded head inputs that have to grow once, one graph per
    frame size, extent of the maps read on the device): the same per-batch losses (1e-3; measured ~1e-6) and the same
    weights, with the lag of the pipeline (two calls: the backbone beside the head is cut by stage)."""
    small_cfg()
    from i2vsgg_amd import ops, train
    monkeypatch.setenv("I2V_UPLOAD_STREAM", "1" if copy_stream else "0")
    imdb, dl = _loader("synthetic_20_v", 2, seed=3)
    rels = imdb.gt_rels(62)
    batches = [d for d in dl][:8]
    sizes = {tuple(d[0].shape[2:]) for d in batches}
    assert len(sizes) >= 3, sizes
    rows = [sum(len(rels[p.split("/")[-1]]["boxes"]) for p in d[4]) for d in batches]
    assert len(set(rows)) >= 4 and max(rows) > 32           # boxes per batch vary and exceed the initial capacity below

    def make(use_graph):
        net = train.build_sgg_net(layers=50, seed=5, device=DEV)
        net.vrd.dropout = False
        net.vrd.source_gt_rels = rels
        step = train.SGGEmbStep(net, 2, device=DEV, n_boxes=16, n_pairs=8, use_graph=use_graph, stage_synthetic=False)
        return net, step

    # (a) eager, exact sizes
    net, step = make(False)
    want = []
    for d in batches:
        assert step.stage_batch(d)
        want.append(float(step()))
    w_want = net.vrd.fc7.fc.weight.detach().cpu().numpy().copy()
    step.opt.unfuse()
    assert len({round(x, 5) for x in want}) == len(want)     # the batches differ

    # (b) captured + overlapped, padded to a capacity
    net, step = make(True)
    assert step.stage_batch(batches[0])
    assert step.capture(warmup=1, restore=True) and step.overlap and step.lag == 2, step.graph_error
    cap0 = (step.cap_boxes, step.cap_pairs)
    keep = torch.zeros(len(batches), device=DEV)

    def stager(d):
        def fn():
            assert step.stage_batch(d)
        return fn
    # head of batch k beside the front half of the backbone on batch k+2 and the back half on batch k+1 (a graph per pair of
    # consecutive sizes); one call of the run trains nothing while the pipeline fills
    train.run_staged(step, [stager(d) for d in batches[1:]], keep)
    assert step.n_bubbles == 1
    torch.cuda.synchronize()
    got = keep.tolist()
    w_got = net.vrd.fc7.fc.weight.detach().cpu().numpy().copy()
    graphs = sum(1 for fs in step.shapes.values() if fs.graph)
    step.opt.unfuse()
    assert step.graph_error is None, step.graph_error
    assert (step._uploader.stream is not None) == copy_stream
    if copy_stream and os.environ.get("I2V_ALIAS_REPRO") != "1":       # tools/alias_repro.py undoes the role streams on purpose
        dev = torch.device(DEV)
        table = ops.stream_table()
        assert step._uploader.stream.cuda_stream == table[(dev.index, "copy", 0)]
        assert len(set(table.values())) == len(table)            # no role shares a handle with another
        cap = getattr(torch.cuda.graph, "default_capture_stream", None)
        assert cap is None or cap.cuda_stream not in table.values()
    # one graph per frame size met since the capacity last grew (growing drops every graph)
    assert 2 <= graphs <= len(sizes) and set(k[1:] for k in step.shapes) == sizes
    assert (step.cap_boxes, step.cap_pairs) != cap0          # the capacity grew (and every graph was captured again)
    assert step.cap_boxes >= max(rows)
    for a, b in zip(want, got):
        assert abs(a - b) <= 1e-3 * abs(a), (want, got)
    assert _rel(w_got, w_want) < 1e-4


def test_sgg_script_selects_the_relation_head_variants(small_cfg, tmp_path, capsys):
    """parser_func.py:155-163,182: --use_obj_visual / --spatial_type / --emb_dim select the head the four reference-run goldens
    pin (tests/test_gpu_models.py::test_vrd_head_variants_vs_reference_golden).  The non-default heads train through the
    model's own forward on eager launches; --emb_dim composes only without the visual embeddings, as in the reference."""
    import trainval_sgg_emb as ts
    common = ["--bs", "2", "--imdb_name", "synthetic_12_v", "--scale", "192", "--disp_interval", "2", "--save_dir", str(tmp_path),
              "--iters_per_epoch", "4", "--epochs", "1"]
    ts.main(common + ["--use_obj_visual", "0", "--spatial_type", "1"])
    out = capsys.readouterr().out
    assert "variant head" in out and "loss:" in out
    ck = torch.load(tmp_path / "res101" / "synthetic" / "SGG_emb_p_prior_adap_synthetic_pre_det_session_1_epoch_1_step_3_un.pth",
                    map_location="cpu")
    assert ck["model"]["vrd.fc_lov.fc.weight"].shape == (256, 8) and "vrd.fc_so.fc.weight" not in ck["model"]
    assert ck["model"]["vrd.fc_fusion.fc.weight"].shape == (256, 512)           # two branches: union-box feature + location
    assert all(torch.isfinite(v).all() for v in ck["model"].values() if v.is_floating_point())
    ts.main(common + ["--spatial_type", "0", "--no-save"])
    assert "variant head" in capsys.readouterr().out
    ts.main(common + ["--emb_dim", "128", "--use_obj_visual", "0", "--no-save"])
    assert "variant head" in capsys.readouterr().out
    with pytest.raises(SystemExit):                      # fc_so is FC(300*2, 256) in the reference: no other width composes with it
        ts.main(common + ["--emb_dim", "128", "--no-save"])


def test_sgg_padded_rows_have_no_effect(small_cfg):
    """Capacity padding is exact: the same batch through the captured step with a tight and with a generous capacity gives
    the same loss and the same update (pad rows carry loss weight 0: zero gradient; dropout off)."""
    small_cfg()
    from i2vsgg_amd import train
    imdb, dl = _loader("synthetic_10_v", 2, seed=1)
    rels = imdb.gt_rels(62)
    d = next(iter(dl))
    out = []
    for nb, npair in ((32, 32), (48, 64)):
        net = train.build_sgg_net(layers=50, seed=5, device=DEV)
        net.vrd.dropout = False
        net.vrd.source_gt_rels = rels
        step = train.SGGEmbStep(net, 2, device=DEV, n_boxes=nb, n_pairs=npair, overlap=False, stage_synthetic=False)
        assert step.stage_batch(d)
        assert step.capture(warmup=1, restore=True), step.graph_error
        losses = [float(step()) for _ in range(3)]
        out.append((losses, net.vrd.fc7.fc.weight.detach().cpu().numpy().copy(), net.vrd.fc_rel.fc.bias.detach().cpu().numpy().copy()))
        step.opt.unfuse()
    (l0, w0, b0), (l1, w1, b1) = out
    assert l0[0] != l0[2]
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 1e-5 * abs(a), (l0, l1)
    assert _rel(w1, w0) < 1e-5 and _rel(b1, b0) < 1e-5


def test_instance_styled_captured_step_consumes_loader_batches_of_varying_size(small_cfg):
    """instance_styleD: 6 (source, target) loader minibatch pairs in >= 3 size combinations through the captured step -- one
    HIP graph per (source size, target size), captured on first sight -- and through the same step on eager launches: the
    same per-batch losses (1e-3).  Both runs sample anchors / ROIs on the device from the same generator state (the captured
    form's sampler), with the rates at zero so that batch k's losses do not depend on the batches before it; a second
    captured run with the default rate trains."""
    cfg = small_cfg(["TRAIN.BATCH_SIZE", "16", "TRAIN.RPN_POST_NMS_TOP_N_TARGET", "16"])
    from i2vsgg_amd import train
    _, dl_s = _loader("synthetic_12_v", 2, seed=2, flipped=True, paths=False)
    _, dl_t = _loader("synthetic_10_v_7", 2, seed=5, flipped=True, paths=False)
    pairs = list(zip([d for d in dl_s][:6], [d for d in dl_t][:6]))
    keys = {(tuple(s[0].shape[2:]), tuple(t[0].shape[2:])) for s, t in pairs}
    assert len(keys) >= 3, keys
    assert any(a != b for a, b in keys)                      # source and target sizes differ within a step

    def run(graph, lr):
        torch.manual_seed(0)
        np.random.seed(cfg.RNG_SEED)
        net = train.build_instance_styled_net(101, device=DEV)
        step = train.InstanceStyleDStep(net, 2, lr=lr, device=DEV, stage_synthetic=False)
        w0 = net.RCNN_base[6][22].conv3.weight.detach().clone()
        out = []
        assert step.stage_batch(*pairs[0])
        if graph:
            assert step.capture(warmup=1, restore=True), step.graph_error
        torch.manual_seed(7)                                 # both runs draw their samples from the same generator state
        for k, (s, t) in enumerate(pairs):
            if k:
                assert step.stage_batch(s, t)
            if graph:
                step()
            else:
                step._device_sampling(True)
                step._body_branches()
            out.append({n: float(v) for n, v in step.losses.items()})
        moved = not torch.equal(w0, net.RCNN_base[6][22].conv3.weight.detach())
        n_graphs = sum(1 for d in step.sets.values() if d.graph)
        return out, moved, n_graphs, step.graph_error

    want, moved, _, _ = run(False, 0.0)
    assert not moved
    got, moved, n_graphs, err = run(True, 0.0)
    assert err is None, err
    assert n_graphs == len(keys) and not moved
    for a, b in zip(want, got):
        for n in a:
            assert np.isfinite(b[n]) and abs(a[n] - b[n]) <= 1e-3 * max(abs(a[n]), 1e-6), (n, want, got)
    assert len({round(d["total"], 4) for d in got}) == len(got)          # the batches differ
    trained, moved, n_graphs, err = run(True, 5e-4)
    assert err is None and moved and n_graphs == len(keys)
    assert all(np.isfinite(v) for d in trained for v in d.values()), trained


def test_device_front_end_loader_equals_the_host_loader(small_cfg):
    """``roibatchLoader(device_prep=True)`` + ``stage_batch_u8`` (uint8 frames across PCIe, BGR swap / flip / mean subtraction /
    resize / canvas placement by ``i2v_image_prep``) stages the same bits as the host form (``get_minibatch`` + the loader's
    padding) for every minibatch size, flipped frames included -- frames, im_info, head inputs -- so the captured relation
    step returns identical losses either way."""
    small_cfg()
    from i2vsgg_amd import train
    from i2vsgg_amd.model.utils import config as c
    from i2vsgg_amd.model.utils.net_utils import sampler
    from i2vsgg_amd.roi_data_layer.roibatchLoader import collate_device_prep, roibatchLoader
    from i2vsgg_amd.roi_data_layer.roidb import combined_roidb
    c.cfg.TRAIN.USE_FLIPPED = True
    imdb, roidb, ratio_list, ratio_index = combined_roidb("synthetic_10_v")
    mk = lambda dp: torch.utils.data.DataLoader(
        roibatchLoader(roidb, ratio_list, ratio_index, 2, imdb.num_classes, training=True, path_return=True, device_prep=dp),
        batch_size=2, sampler=sampler(len(roidb), 2, seed=9), pin_memory=True, collate_fn=collate_device_prep if dp else None)
    np.random.seed(4)
    host = list(mk(False))
    np.random.seed(4)
    dev = list(mk(True))
    assert len(host) == len(dev) == 10
    rels = imdb.gt_rels(62)
    rels.update({k: v for k, v in rels.items()})
    net = train.build_sgg_net(layers=50, seed=5, device=DEV)
    net.vrd.dropout = False
    net.vrd.source_gt_rels = rels
    step = train.SGGEmbStep(net, 2, vrd_lr=0.0, device=DEV, stage_synthetic=False)
    assert step.stage_batch(host[0]) and step.capture(warmup=1, restore=True), step.graph_error
    checked = flipped = 0
    sizes = set()

    def settle():
        """The loss of the minibatch staged last: ``lag`` calls bring it to the head (rates are zero: calls do not train)."""
        for _ in range(1 + step.lag):
            last = float(step())
        return last
    for h, d in zip(host, dev):
        assert torch.equal(h[2], d[2]) and torch.equal(h[3], d[3]) and list(h[4]) == list(d[4])
        if int(d[1][0][1]) == 0:                       # square trim: the device form hands it back
            assert not step.stage_batch_u8(d)
            continue
        assert step.stage_batch(h)
        want_im, want_info = step.im.clone(), step.info.copy()
        la2 = settle()
        want_slot = step.inp.buf.clone()               # after the call: a size met for the first time gets its extent there
        assert step.stage_batch_u8(d)
        assert torch.equal(step.im, want_im), tuple(h[0].shape)
        assert torch.equal(step.im[:, :3].cpu(), h[0]) and float(step.im[:, 3].abs().max()) == 0.0
        diff = [name for name, o, nb, dt, sh in step.inp.spec
                if not torch.equal(step.inp.buf[o:o + nb], want_slot[o:o + nb])]
        assert not diff and np.array_equal(step.info, want_info), (diff, step.info, want_info)
        lb2 = settle()
        assert abs(la2 - lb2) <= 1e-6 * abs(la2), (la2, lb2)     # rates are zero: the same bits in, the same loss out (up to
        #                                                           the fp32 atomics of the head's split-K GEMMs, ~1e-7)
        checked += 1
        flipped += int(d[1][:, 0].sum())
        sizes.add(tuple(h[0].shape[2:]))
    step.opt.unfuse()
    assert checked >= 6 and flipped >= 2 and len(sizes) >= 3, (checked, flipped, sizes)


def test_training_scripts_run_through_the_data_layer(small_cfg, tmp_path):
    """trainval_sgg_emb.py / trainval_instance_styled.py (the reference loops on the HIP path) fed by
    combined_roidb -> roibatchLoader -> DataLoader(sampler) on synthetic imdbs whose frames differ in size: two epochs of
    the captured steps, reference-layout checkpoints (``epoch`` = the next epoch, reference file names and state_dict keys),
    resume into a third epoch with the learning-rate decay of that epoch applied (round-2 advice: a resumed run dropped
    it)."""
    import trainval_instance_styled as tv
    import trainval_sgg_emb as ts
    # ---- SGG_emb
    common = ["--bs", "2", "--imdb_name", "synthetic_12_v", "--scale", "192", "--disp_interval", "3", "--save_dir", str(tmp_path), "--lr_decay_step", "1", "--lr_decay_gamma", "0.5", "--vrd_lr", "1e-4"]
    ts.main(["--epochs", "2"] + common)
    name = tmp_path / "res101" / "synthetic" / "SGG_emb_p_prior_adap_synthetic_pre_det_session_1_epoch_2_step_5_un.pth"
    ck = torch.load(name, map_location="cpu")
    assert ck["epoch"] == 3 and ck["pooling_mode"] == "align"
    for k in ("RCNN_base.0.weight", "vrd.fc6.fc.weight", "vrd.prd_sem_embeddings.2.bias", "vrd.conv_lo.2.conv.weight"):
        assert k in ck["model"], k
    assert all(torch.isfinite(v).all() for v in ck["model"].values() if v.is_floating_point())
    lrs = {round(g["lr"], 12) for g in ck["optimizer"]["param_groups"]}
    assert lrs == {round(1e-4 * 0.5, 12), round(2e-4 * 0.5, 12)}          # one decay (epoch 2); biases at twice the rate
    ts.main(["--epochs", "3", "--resume_train", "--load_name", str(name)] + common)
    ck3 = torch.load(str(name).replace("epoch_2", "epoch_3"), map_location="cpu")
    assert ck3["epoch"] == 4
    lrs = {round(g["lr"], 12) for g in ck3["optimizer"]["param_groups"]}
    assert lrs == {round(1e-4 * 0.25, 12), round(2e-4 * 0.25, 12)}        # the resumed epoch 3 decayed once more
    assert not torch.equal(ck3["model"]["vrd.fc7.fc.weight"], ck["model"]["vrd.fc7.fc.weight"])
    assert torch.equal(ck3["model"]["RCNN_base.6.5.conv3.weight"], ck["model"]["RCNN_base.6.5.conv3.weight"])     # frozen here
    # ---- instance_styleD
    common = ["--bs", "2", "--imdb_name", "synthetic_6_v", "--imdb_name_target", "synthetic_5_v_7", "--scale", "192",
              "--iters_per_epoch", "3", "--disp_interval", "3", "--save_dir", str(tmp_path), "--lr_decay_step", "1", "--set",
              "TRAIN.BATCH_SIZE", "16", "TRAIN.RPN_POST_NMS_TOP_N_TARGET", "16"]
    tv.main(["--epochs", "2"] + common)
    name = tmp_path / "res101" / "synthetic" / ("instance_pixel_styleD_bilinear_cr_False_source_synthetic_target_synthetic_t_"
                                                "session_1_lr_0.0005_epoch_2_bs_2_mscoco.pth")
    ck = torch.load(name, map_location="cpu")
    assert ck["epoch"] == 3 and ck["pooling_mode"] == "align" and ck["class_agnostic"] is False
    for k in ("RCNN_base.0.weight", "RCNN_base.6.22.conv3.weight", "RCNN_top.0.2.bn3.running_var", "netD_pixel.conv1.weight",
              "netD_style.fc_1.weight", "RCNN_rpn.RPN_Conv.weight", "RCNN_cls_score.bias"):
        assert k in ck["model"], k
    assert all(torch.isfinite(v).all() for v in ck["model"].values() if v.is_floating_point())
    lrs = {round(g["lr"], 10) for g in ck["optimizer"]["param_groups"]}
    assert lrs == {round(5e-4 * 0.1, 10), round(2 * 5e-4 * 0.1, 10)}          # one decay; biases at twice the rate
    tv.main(["--epochs", "3", "--r", "--load_name", str(name)] + common)
    ck3 = torch.load(str(name).replace("epoch_2", "epoch_3"), map_location="cpu")
    assert ck3["epoch"] == 4
    lrs = {round(g["lr"], 10) for g in ck3["optimizer"]["param_groups"]}
    assert lrs == {round(5e-4 * 0.01, 10), round(2 * 5e-4 * 0.01, 10)}        # the decay of the resumed epoch is applied
    assert not torch.equal(ck3["model"]["RCNN_base.6.22.conv3.weight"], ck["model"]["RCNN_base.6.22.conv3.weight"])
    # ---- --o adam (trainval_net_*.py:143-147): torch.optim.Adam's state layout in the checkpoint, resume from it
    tv.main(["--epochs", "1", "--o", "adam", "--s", "2"] + common)
    ck_a = torch.load(str(name).replace("session_1", "session_2").replace("epoch_2", "epoch_1"), map_location="cpu")
    st = ck_a["optimizer"]["state"][0]
    assert set(st) == {"step", "exp_avg", "exp_avg_sq"} and float(st["step"]) == 3.0 and ck_a["optimizer"]["param_groups"][0]["betas"] == (0.9, 0.999)
    assert all(torch.isfinite(v).all() for v in ck_a["model"].values() if v.is_floating_point())
    tv.main(["--epochs", "2", "--o", "adam", "--s", "2", "--r", "--checksession", "2", "--checkepoch", "1"] + common)
    ts.main(["--epochs", "1", "--o", "adam", "--bs", "2", "--imdb_name", "synthetic_12_v", "--scale", "192", "--disp_interval", "3",
             "--save_dir", str(tmp_path / "adam"), "--vrd_lr", "1e-4"])
    # ---- both loops once more with the device front-end (uint8 frames, image work on the GPU)
    ts.main(["--epochs", "1", "--bs", "2", "--imdb_name", "synthetic_12_v", "--scale", "192", "--disp_interval", "3", "--save_dir",
             str(tmp_path / "u8"), "--device_prep", "--no-save"])
    tv.main(["--epochs", "1", "--device_prep", "--no-save"] + common)


def test_stage_chaining_detector_checkpoint_initialises_the_relation_stage(small_cfg, tmp_path):
    """The method's two stages are chained through checkpoints (scripts/SGG_emb_resnet.sh: ``--r --load_name adapt/instance_pixel_
    styleD_bilinear_...pth``):
      * trainval_net_SGG_emb.py:155-173 -- ``--r`` loads every key WITHOUT 'vrd' from the detector file, prints the keys the
        file lacks, takes its pooling mode; the relation head keeps its initialisation, no optimizer state is read;
      * trainval_net_instance_styleD_bilinear.py:153-183 -- with 'faster_rcnn' in --load_name the detector is initialised
        through an allow-list (no netD_pixel / RPN_cls_score / RPN_bbox_pred / RCNN_cls_score / RCNN_bbox_pred), from a file that
        may lack the discriminators altogether.
    Here: two steps of trainval_instance_styled.py, its checkpoint into trainval_sgg_emb.py --r; then a plain-detector file (no
    netD_*) into trainval_instance_styled.py --r."""
    import trainval_instance_styled as tv
    import trainval_sgg_emb as ts
    from i2vsgg_amd import train
    from i2vsgg_amd.model.utils.config import cfg
    det_args = ["--net", "res50", "--bs", "2", "--imdb_name", "synthetic_6_v", "--imdb_name_target", "synthetic_5_v_7", "--scale", "192",
                "--iters_per_epoch", "2", "--disp_interval", "2", "--save_dir", str(tmp_path), "--set", "TRAIN.BATCH_SIZE", "16",
                "TRAIN.RPN_POST_NMS_TOP_N_TARGET", "16"]
    tv.main(["--epochs", "1"] + det_args)
    det = tmp_path / "res50" / "synthetic" / ("instance_pixel_styleD_bilinear_cr_False_source_synthetic_target_synthetic_t_"
                                              "session_1_lr_0.0005_epoch_1_bs_2_mscoco.pth")
    ck = torch.load(det, map_location="cpu")
    fresh = train.build_instance_styled_net(50, n_cls=16, device=DEV).state_dict()
    assert not torch.equal(ck["model"]["RCNN_base.6.5.conv3.weight"], fresh["RCNN_base.6.5.conv3.weight"].cpu())     # it did train

    # ---- stage 2 from stage 1: the function, then the script
    net = train.build_sgg_net(50, device=DEV)
    twin = train.build_sgg_net(50, device=DEV).state_dict()
    cfg.POOLING_MODE = "pool"
    said = []
    loaded, missing = ts.init_from_detector(str(det), net, log=said.append)
    assert cfg.POOLING_MODE == "align" and not missing and not said
    sd = net.state_dict()
    assert loaded and all("vrd" not in k for k in loaded)
    assert set(loaded) == {k for k in sd if "vrd" not in k}
    for k, v in sd.items():
        if "vrd" in k:
            assert torch.equal(v, twin[k]), k                            # the relation head keeps its initialisation
        else:
            assert torch.equal(v.cpu(), ck["model"][k]), k                # RCNN_base / RCNN_rpn / RCNN_top / cls / bbox: the detector's
    assert not any(k.startswith("netD_") for k in sd)
    # a file that lacks keys: they are printed and keep their initialisation (:160-163)
    part = {"model": {k: v for k, v in ck["model"].items() if not k.startswith("RCNN_top.0.")}, "pooling_mode": "align"}
    torch.save(part, tmp_path / "partial.pth")
    net2 = train.build_sgg_net(50, device=DEV)
    said = []
    _, missing = ts.init_from_detector(str(tmp_path / "partial.pth"), net2, log=said.append)
    assert missing and said == missing and all(k.startswith("RCNN_top.0.") for k in missing)
    assert torch.equal(net2.state_dict()["RCNN_top.0.0.conv1.weight"], twin["RCNN_top.0.0.conv1.weight"])
    sgg_args = ["--net", "res50", "--bs", "2", "--imdb_name", "synthetic_12_v", "--scale", "192", "--iters_per_epoch", "2",
                "--disp_interval", "2", "--save_dir", str(tmp_path), "--vrd_lr", "1e-4"]
    ts.main(["--epochs", "1", "--r", "--load_name", str(det)] + sgg_args)
    out = torch.load(tmp_path / "res50" / "synthetic" / "SGG_emb_p_prior_adap_synthetic_pre_det_session_1_epoch_1_step_1_un.pth",
                     map_location="cpu")
    assert out["epoch"] == 2                                              # started at --start_epoch 1, not at the file's epoch
    for k in ("RCNN_base.0.weight", "RCNN_base.6.5.conv3.weight", "RCNN_rpn.RPN_Conv.weight", "RCNN_top.0.2.conv3.weight",
              "RCNN_cls_score.weight"):
        assert torch.equal(out["model"][k], ck["model"][k]), k           # bit-equal to the detector's (frozen in this stage)
    assert not torch.equal(out["model"]["vrd.fc7.fc.weight"], twin["vrd.fc7.fc.weight"].cpu())      # the head trained
    lrs = {round(g["lr"], 12) for g in out["optimizer"]["param_groups"]}
    assert lrs == {1e-4, round(1e-4 * (cfg.TRAIN.DOUBLE_BIAS + 1), 12)}   # a fresh optimizer: nothing of the detector's rates (5e-4)
    with pytest.raises(SystemExit):
        ts.main(["--epochs", "1", "--r", "--resume_train", "--load_name", str(det)] + sgg_args)

    # ---- stage 1 from a plain detector file: no discriminators in it, 'faster_rcnn' in its name
    plain = {"model": {k: v for k, v in ck["model"].items() if not k.startswith("netD_")}, "epoch": 7, "session": 9,
             "pooling_mode": "pool"}
    name = tmp_path / "faster_rcnn_1_7_9999.pth"
    torch.save(plain, name)
    dnet = train.build_instance_styled_net(50, n_cls=16, device=DEV)
    init = {k: v.clone() for k, v in dnet.state_dict().items()}
    cfg.POOLING_MODE = "align"
    loaded = tv.init_from_detector(str(name), dnet)
    assert cfg.POOLING_MODE == "align"                                    # :181 looks in the model dict: never taken
    sd = dnet.state_dict()
    for k, v in sd.items():
        dropped = any(t in k for t in tv.WO_PARAMETER)
        if dropped or k.startswith("netD_"):
            assert k not in loaded and torch.equal(v, init[k]), k        # keeps its initialisation
        else:
            assert k in loaded and torch.equal(v.cpu(), ck["model"][k]), k
    assert any("RPN_cls_score" in k for k in sd) and any(k.startswith("netD_style") for k in sd)
    tv.main(["--epochs", "1", "--s", "3", "--r", "--load_name", str(name), "--no-save"] + det_args)     # runs from epoch 1


def test_test_scripts_equal_their_frame_by_frame_form(small_cfg, tmp_path):
    """test_instance_styled.py / test_sgg_emb.py (the reference's test loops: combined_roidb(name, False) ->
    roibatchLoader(training=False) -> DataLoader(batch_size=1)) on a synthetic imdb whose frames differ in size: the graph form
    (frames grouped by size, 3 per replay, short groups at the end) writes what the frame-by-frame form writes, a checkpoint of
    the training script loads, ``all_boxes`` has the reference's [class][image] layout."""
    import pickle
    import test_instance_styled as td
    import test_sgg_emb as tr
    import trainval_instance_styled as tv
    from i2vsgg_amd._lib import TUNE, lib
    common = ["--imdbval_name", "synthetic_10_v", "--scale", "192", "--set", "TEST.RPN_POST_NMS_TOP_N", "64"]
    tv.main(["--epochs", "1", "--bs", "2", "--imdb_name", "synthetic_6_v", "--imdb_name_target", "synthetic_5_v_7", "--scale", "192",
             "--iters_per_epoch", "2", "--save_dir", str(tmp_path), "--set", "TRAIN.BATCH_SIZE", "16",
             "TRAIN.RPN_POST_NMS_TOP_N_TARGET", "16"])
    ck = tmp_path / "res101" / "synthetic" / ("instance_pixel_styleD_bilinear_cr_False_source_synthetic_target_synthetic_t_"
                                             "session_1_lr_0.0005_epoch_1_bs_2_mscoco.pth")
    old = lib.i2v_get_tuning(TUNE["I2V_SPLIT_BELOW"])
    try:
        lib.i2v_set_tuning(TUNE["I2V_SPLIT_BELOW"], 0)        # no split-K atomics: two forwards of a frame are bit-equal
        one = td.main(["--frames", "1", "--load_name", str(ck), "--output_dir", str(tmp_path / "o1")] + common)
        many = td.main(["--frames", "3", "--load_name", str(ck), "--output_dir", str(tmp_path / "o3")] + common)
        r_one = tr.main(["--frames", "1", "--output_dir", str(tmp_path / "o1"), "--imdbval_name", "synthetic_10_v", "--scale", "192"])
        r_many = tr.main(["--frames", "3", "--output_dir", str(tmp_path / "o3"), "--imdbval_name", "synthetic_10_v", "--scale", "192"])
        # the device front-end in test mode: uint8 frames from the loader, BGR swap / mean subtraction / resize on the GPU
        many_u8 = td.main(["--frames", "3", "--device_prep", "--load_name", str(ck), "--output_dir", str(tmp_path / "o8")] + common)
        r_u8 = tr.main(["--frames", "3", "--device_prep", "--output_dir", str(tmp_path / "o8"), "--imdbval_name", "synthetic_10_v",
                        "--scale", "192"])
    finally:
        lib.i2v_set_tuning(TUNE["I2V_SPLIT_BELOW"], old)
    with open(tmp_path / "o3" / "res101" / "synthetic" / "detections.pkl", "rb") as f:
        saved = pickle.load(f)
    assert len(saved) == 16 and len(saved[1]) == 10 and saved[0][0] == []
    n_det = 0
    for j in range(1, 16):
        for i in range(10):
            assert saved[j][i].shape[1] == 5 and np.array_equal(saved[j][i], one[j][i]) and np.array_equal(many[j][i], one[j][i])
            assert np.array_equal(many_u8[j][i], one[j][i]), (j, i)
            n_det += len(saved[j][i])
    assert n_det > 0
    assert set(r_one) == set(r_many) and len(r_one) == 10
    for path, want in r_one.items():
        for a, b in zip(r_many[path], want):
            assert np.array_equal(np.asarray(a), np.asarray(b)), path
        for a, b in zip(r_u8[path], want):
            assert np.array_equal(np.asarray(a), np.asarray(b)), path
    with open(tmp_path / "o3" / "res101" / "synthetic" / "relations.pkl", "rb") as f:
        assert set(pickle.load(f)) == set(r_one)
