"""End-to-end GPU parity of the HIP RCFModel: against the golden vectors captured from the
reference (tests/golden/rcf_small*.npz) and against the oracle restatement run on the same seeded
weights/inputs.  Tolerance: 1e-4 relative on losses / masks / gradients (BASELINE.json north_star);
argmax exact on every pixel whose top-2 logit margin exceeds that same tolerance (1e-4 x max |logit|).
Gradients of this network are ill-conditioned in fp32 (ReLU kinks, BN over small batches): the reference's own
fp32 results move by up to 1e-3 when it is run single-threaded or with channels_last convolutions, so gradient
checks use the float64 truth with 4x the worst of those reference fp32 runs as the limit (fixtures `ref32_err_*`)."""
import copy
import os
import types

import numpy as np
import pytest
import torch

import rcf_amd
from rcf_amd import config, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def _args():
    return types.SimpleNamespace(checkpoints_dir="/tmp/rcf_test", object_channel=None, eval_save=False,
                                 eval_export=False)


def _batch(B, H, W, device):
    nb = synth.make_batch(B, H, W, config_id=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return {"imgs": [t(a) for a in nb["imgs"]], "gt_fw_flows": [t(a) for a in nb["gt_fw_flows"]],
            "gt_bw_flows": [t(a) for a in nb["gt_bw_flows"]], "seq_ids": nb["seq_ids"], "seq_names": nb["seq_names"],
            "paths": nb["paths"]}


DROP_P, DROP_SEED2, DROP_SEED3 = 0.1, 21, 22              # tests/golden/make_golden_dropout.py


def _build(H, W, affine, device, cls, benched=None):
    """benched = B: the configuration bench.py runs -- SyncBN + Dropout2d 0.1 in both FCN heads
    (configs/rcf/rcf_stage1.yaml:83-85,118,139) -- with the Dropout2d draw of the dropout.* fixtures injected: into
    FCNHead.keep_mask here, as oracle.FixedDropout2d into the oracle (the reference took the same module when the fixtures
    were made)."""
    if benched is None:
        kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, affine=affine, norm="BN")
    else:
        kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=DROP_P, affine=affine, norm="SyncBN")
    kw.update(log_interval=10 ** 9, train_iter=1)
    m = cls(_args(), **copy.deepcopy(kw))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()}
    m.load_state_dict(sd)
    m = m.to(device)
    if benched is not None:
        s2 = torch.from_numpy(synth.dropout_scale(2 * benched, m.decode_head2.channels, DROP_P, DROP_SEED2))
        s3 = torch.from_numpy(synth.dropout_scale(benched, m.decode_head3.channels, DROP_P, DROP_SEED3))
        if cls is rcf_amd.RCFModel:
            m.decode_head2.keep_mask, m.decode_head3.keep_mask = s2.to(device), s3.to(device)
        else:
            import rcf_torch as orc
            m.decode_head2.dropout, m.decode_head3.dropout = orc.FixedDropout2d(s2), orc.FixedDropout2d(s3)
    return m


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("tag,H,W,affine", [("rcf_small", 96, 160, False), ("rcf_small_affine", 64, 96, True)])
def test_train_step_vs_reference_golden(tag, H, W, affine, golden_dir, report):
    fx = np.load(os.path.join(golden_dir, tag + ".npz"))
    B = int(fx["B"])
    model = _build(H, W, affine, DEV, rcf_amd.RCFModel)
    tr = rcf_amd.Trainer(model, lr=1e-4, weight_decay=1e-4, device=DEV)
    batch = _batch(B, H, W, DEV)
    # forward-only products first (fresh weights): masks / logits / residuals
    model.train()
    with torch.no_grad():
        from rcf_amd.layers import Tape, pair_concat
        t = Tape(enabled=False)
        imgs = torch.stack(batch["imgs"], dim=1)
        model._select_precision()                           # fp32 (model.precision is None, no autocast): _images_nhwc reads it
        img = model._images_nhwc(imgs)
        saved = copy.deepcopy(model.state_dict())           # BN running stats move in train mode
        feats = model.backbone2.fwd(img, t)
        logits = model.decode_head2.fwd(feats, t)
        res = model.decode_head3.fwd([pair_concat(feats[-1], t, B, 2)], t)
        model.load_state_dict(saved)
    l_nchw = rcf_amd.ops.nhwc_to_nchw(logits.t).cpu().numpy()
    r_nchw = rcf_amd.ops.nhwc_to_nchw(res.t).cpu().numpy()
    e_logits = rel(l_nchw, fx["logits"])
    e_res = max(rel(r_nchw[:, :8], fx["res_fw"]), rel(r_nchw[:, 8:], fx["res_bw"]))
    am = l_nchw.argmax(1).astype(np.uint8)
    sure = fx["margin"].astype(np.float32) > TOL * float(fx["logit_absmax"])   # top-2 gap above the logits' own tolerance
    mism = int((am != fx["argmax"])[sure].sum())
    feat_e = max(rel(float(f.t.abs().mean()), fx["feat_absmean"][i]) for i, f in enumerate(feats))
    # one full training step
    losses = tr.step(batch)
    e_loss = {k: rel(float(losses[k]), float(fx[k])) for k in ("loss", "loss_warp_seg", "loss_entropy")}
    named = dict(model.named_parameters())
    # gradients: this net's per-element gradients are ill-conditioned in fp32 (the reference's own CPU fp32
    # run is `ref32_err_*` away from the float64 ground truth), so the yardstick is the float64 truth with a
    # threshold of 4x the reference's own fp32 error (floor 1e-4)
    e_grad, lim_grad = {}, {}
    for i, name in enumerate(fx["sampled"]):
        g = named[str(name)].grad.detach().cpu().contiguous().numpy().ravel()[:256]
        e_grad[str(name)] = rel(g, fx[f"truth_grad_{i}"])
        lim_grad[str(name)] = max(TOL, 4 * float(fx["ref32_err_grad"][i]))
    gn = {}
    for n, p in named.items():
        if p.grad is not None:
            gn[n.split(".")[0]] = gn.get(n.split(".")[0], 0.0) + float(p.grad.double().pow(2).sum())
    e_gn = {str(k): rel(np.sqrt(gn[str(k)]), v) for k, v in zip(fx["gradnorm_keys"], fx["truth_gradnorm"])}
    lim_gn = {str(k): max(TOL, 4 * float(v)) for k, v in zip(fx["gradnorm_keys"], fx["ref32_err_gradnorm"])}
    e_logits64 = rel(l_nchw, fx["truth_logits"])
    with torch.no_grad():
        model.train_iter = 1
        l2 = model(batch)
    e_after = rel(float(l2["loss"]), float(fx["loss_after_step"]))
    report(f"{tag}: logits vs ref {e_logits:.2e} vs f64 {e_logits64:.2e} (ref32 {float(fx['ref32_err_logits']):.2e}) "
           f"res {e_res:.2e} feat {feat_e:.2e} argmax mismatches(sure px) {mism} unsure px {int((~sure).sum())} "
           f"loss {e_loss} gradnorm vs f64 {e_gn} (limits {lim_gn}) grad vs f64 {e_grad} (limits {lim_grad}) "
           f"loss_after_step {e_after:.2e}")
    assert e_logits < TOL and e_res < TOL and feat_e < TOL
    assert e_logits64 < max(TOL, 4 * float(fx["ref32_err_logits"]))
    assert mism == 0
    assert max(e_loss.values()) < TOL
    assert all(e_gn[k] < lim_gn[k] for k in e_gn), (e_gn, lim_gn)
    assert all(e_grad[k] < lim_grad[k] for k in e_grad), (e_grad, lim_grad)
    # the first Adam step is sign-like (g / (|g| + 1e-8)): fp32 noise on near-zero gradients flips +-lr updates,
    # so the post-step loss is only loosely comparable (the Adam kernel itself is pinned in test_kernels_gpu)
    assert e_after < 1e-2


@pytest.mark.parametrize("H,W,B,benched", [(64, 96, 2, False), (480, 854, 1, False), (96, 160, 2, True), (480, 854, 1, True)])
def test_train_step_all_grads_at_fixed_relu_pattern(H, W, B, benched, monkeypatch, report):
    """(benched: SyncBN + Dropout2d 0.1 with an injected draw -- the configuration bench.py times, _build.)
    EVERY parameter gradient against float64 with the ReLU lottery taken out -- at the small geometry and at the FULL
    480x854 frame size (one pair: the float64 oracle fits a host's memory), so that a full-size shape meets the tight
    criterion too.  A ReLU unit whose pre-activation lies
    within fp32 noise of zero falls either way in two equally valid fp32 evaluations, and one such unit moves a channel's
    weight gradient by percent -- a natural-pattern comparison needs a floor (1e-2) that would also hide a real precision
    loss.  Here the float64 truth (and the CPU fp32 yardstick) are evaluated AT THE ACTIVATION PATTERN OF THE HIP RUN:
    the HIP forward records `output > 0` of every batch norm + ReLU (layers.RELU_TRACE), and the oracle's F.relu is
    replaced by a multiplication with those masks.  With the pattern fixed the network is smooth in its parameters, so
    what remains is arithmetic error: limit 6x the CPU fp32 run's own error at the same pattern, floor 1e-4."""
    import rcf_torch as orc
    from rcf_amd import layers
    if H * W > 100000:
        avail = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 1e6
        if avail < 40:
            pytest.skip(f"the float64 oracle at {H}x{W} needs ~25 GB of host memory ({avail:.0f} GB available)")
    bb = B if benched else None
    hip = _build(H, W, False, DEV, rcf_amd.RCFModel, benched=bb)
    tr = rcf_amd.Trainer(hip, device=DEV)
    tr.fp.zero_grad()
    hip.train()
    trace = []
    monkeypatch.setattr(layers, "RELU_TRACE", trace)
    lh = hip(_batch(B, H, W, DEV))
    monkeypatch.setattr(layers, "RELU_TRACE", None)
    lh["loss"].backward()
    masks = [m.permute(0, 3, 1, 2).contiguous().cpu() for m in trace]              # NHWC -> NCHW, forward order

    def forced_run(double):
        m = _build(H, W, False, "cpu", orc.RCFModel, benched=bb)
        b = _batch(B, H, W, "cpu")
        if double:
            m = m.double()
            b = {k: ([t.double() for t in v] if k in ("imgs", "gt_fw_flows", "gt_bw_flows") else v) for k, v in b.items()}
        queue = list(masks)

        def forced_relu(x, inplace=False):
            mk = queue.pop(0)
            assert tuple(mk.shape) == tuple(x.shape), (tuple(mk.shape), tuple(x.shape))
            return x * mk.to(x.dtype)
        monkeypatch.setattr(orc.F, "relu", forced_relu)
        try:
            m.train()
            losses = m(b)
            losses["loss"].backward()
        finally:
            monkeypatch.undo()
        assert not queue, f"{len(queue)} recorded ReLU maps were not consumed: the two forward orders differ"
        return {k: float(v) for k, v in losses.items()}, dict(m.named_parameters())
    l64, g64 = forced_run(True)
    l32, g32 = forced_run(False)
    e_loss = rel(float(lh["loss"]), l64["loss"])
    worst_ratio, worst_name, worst_abs, worst_ref, bad = 0.0, "", 0.0, 0.0, []
    for n, p in hip.named_parameters():
        truth = g64[n].grad
        scale = float(truth.norm())
        if scale < 1e-12:
            continue
        e_hip = float((p.grad.cpu().double() - truth).norm()) / scale
        e_ref = float((g32[n].grad.double() - truth).norm()) / scale
        if e_hip > max(6.0 * e_ref, 1e-4):
            bad.append((n, e_hip, e_ref))
        if e_hip / max(e_ref, 1e-7) > worst_ratio:
            worst_ratio, worst_name, worst_abs, worst_ref = e_hip / max(e_ref, 1e-7), n, e_hip, e_ref
    report(f"all-grads at the HIP run's ReLU pattern, {H}x{W} B={B}{' SyncBN + injected Dropout2d draw' if benched else ''} ({len(masks)} ReLU layers, {sum(int(m.numel()) for m in masks)} units) vs "
           f"float64: loss {e_loss:.2e}; worst HIP / CPU-fp32 error ratio {worst_ratio:.2f} at {worst_name} (HIP {worst_abs:.2e}, "
           f"CPU fp32 {worst_ref:.2e}); parameters over max(6x ref, 1e-4): {len(bad)} {bad[:4]}")
    assert e_loss < 1e-5 and not bad, bad


def test_train_step_all_grads_natural_pattern(report):
    """the cheap companion of the test above: EVERY parameter gradient of the HIP fp32 step against the oracle's float64 at the
    NATURAL activation pattern (64x96, B = 2), with the oracle's own fp32 evaluation as the yardstick.  Here ReLU units within
    fp32 noise of zero fall differently in different evaluations and move whole tensors by ~5e-3 (measured: the oracle's fp32 run
    is that far from its float64 run), so this is a coverage test -- every tensor present, none beyond 5e-2, the typical HIP error
    no larger than twice the typical error of the CPU fp32 run -- and the tight criterion lives in the fixed-pattern test."""
    import rcf_torch as orc
    H, W, B = 64, 96, 2
    hip = _build(H, W, False, DEV, rcf_amd.RCFModel)
    tr = rcf_amd.Trainer(hip, device=DEV)
    tr.fp.zero_grad()
    hip.train()
    hip(_batch(B, H, W, DEV))["loss"].backward()

    def oracle(double):
        m = _build(H, W, False, "cpu", orc.RCFModel)
        b = _batch(B, H, W, "cpu")
        if double:
            m = m.double()
            b = {k: ([t.double() for t in v] if k in ("imgs", "gt_fw_flows", "gt_bw_flows") else v) for k, v in b.items()}
        m.train()
        m(b)["loss"].backward()
        return dict(m.named_parameters())
    truth, ref32 = oracle(True), oracle(False)
    e_hip, e_ref = {}, {}
    for n, p_ in hip.named_parameters():
        t = truth[n].grad
        assert p_.grad is not None and t is not None, n
        scale = float(t.norm())
        if scale < 1e-12:
            continue
        e_hip[n] = float((p_.grad.cpu().double() - t).norm()) / scale
        e_ref[n] = float((ref32[n].grad.double() - t).norm()) / scale
    vh, vr = np.sort(np.array(list(e_hip.values()))), np.sort(np.array(list(e_ref.values())))
    worst = max(e_hip.items(), key=lambda kv: kv[1])
    report(f"all {len(e_hip)} parameter gradients vs float64 at the natural ReLU pattern (64x96 B=2): HIP median {np.median(vh):.1e} / 95 % "
           f"{vh[int(0.95 * len(vh))]:.1e} / worst {worst[1]:.1e} at {worst[0]}; the oracle's own fp32: median {np.median(vr):.1e} / 95 % "
           f"{vr[int(0.95 * len(vr))]:.1e} / worst {vr[-1]:.1e}")
    assert vh[-1] < 5e-2 and np.median(vh) <= 2 * np.median(vr) + 1e-4 and vh[int(0.95 * len(vh))] <= 2 * vr[int(0.95 * len(vr))] + 1e-3


def test_eval_forward_matches_oracle(report):
    import rcf_torch as orc
    H, W, B = 64, 96, 2
    hip = _build(H, W, False, DEV, rcf_amd.RCFModel).eval()
    ora = _build(H, W, False, "cpu", orc.RCFModel).eval()
    bh, bo = _batch(B, H, W, DEV), _batch(B, H, W, "cpu")
    with torch.no_grad():
        ph, po = hip(bh), ora(bo)
    e = rel(ph.cpu().numpy(), po.numpy())
    report(f"eval masks vs oracle: {e:.2e}")
    assert tuple(ph.shape) == tuple(po.shape) and e < TOL


def test_eval_save_and_export(tmp_path, report):
    """evaluation side effects of models/rcf_model.py:275-320: the 2x visualisation list, the eval JPEG grid and
    the `pred_seg_*.png` export (file names and pixel values)"""
    import torch.nn.functional as F
    from PIL import Image
    H, W, B = 64, 96, 2
    hip = _build(H, W, False, DEV, rcf_amd.RCFModel).eval()
    hip.save_dir_eval, hip.save_dir_eval_export = str(tmp_path / "saved_eval"), str(tmp_path / "saved_eval_export")
    hip.args.eval_save, hip.args.eval_export, hip.args.object_channel = True, True, 1
    hip.train_iter = 42
    batch = _batch(B, H, W, DEV)
    batch["paths"] = [[f"/data/JPEGImages/480p/seq{i}/{i:05d}.jpg" for i in range(B)]] * 2
    batch["seq_names"], batch["seq_ids"] = [f"seq{i}" for i in range(B)], torch.tensor([3, 4])
    with torch.no_grad():
        masks, vis = hip(batch, return_pred_vis_list=True)
    h, w = masks.shape[-2:]
    assert len(vis) == 4 and tuple(vis[0].shape) == (B, 3, 2 * h, 2 * w)
    up = F.interpolate(masks[:, 1:2].cpu(), size=(2 * h, 2 * w), mode="bilinear", align_corners=False)
    e_vis = float((vis[1][:, 0].cpu() - up[:, 0]).abs().max())
    jpg = tmp_path / "saved_eval" / "eval_seq0_3_00000_0000042.jpg"
    pngs = [tmp_path / "saved_eval_export" / f"pred_seg_seq{i}_{i:05d}_0000042.png" for i in range(B)]
    assert jpg.exists() and all(p.exists() for p in pngs), list(tmp_path.rglob("*"))
    grid = np.asarray(Image.open(jpg))
    assert grid.shape == (5 * 2 * h + 4, 2 * (2 * w + 2) + 2, 3)          # make_grid of B=2 images, padding 2
    worst = 0
    for i, p in enumerate(pngs):
        img = np.asarray(Image.open(p)).astype(np.int32)
        want = (up[i, 0] * 255 + 0.5).clamp(0, 255).to(torch.uint8).numpy().astype(np.int32)
        assert img.shape == (2 * h, 2 * w, 3)
        worst = max(worst, int(np.abs(img[..., 0] - want).max()))
    report(f"eval export: vis list vs torch bilinear {e_vis:.2e}, PNG max |u8 diff| {worst}")
    assert e_vis < 2e-5 and worst <= 1
    # all-channel export goes to numbered sub-directories
    hip.args.export_all_seg = True
    with torch.no_grad():
        hip(batch)
    assert all((tmp_path / "saved_eval_export" / str(c) / "pred_seg_seq0_00000_0000042.png").exists() for c in range(4))


def test_fullsize_480x854_vs_reference_golden(golden_dir, report):
    """BASELINE config-1 geometry (480x854, mask 120x214), one pair, against the reference's own output."""
    fx = np.load(os.path.join(golden_dir, "rcf_480x854_b1.npz"))
    H, W, B = int(fx["H"]), int(fx["W"]), int(fx["B"])
    model = _build(H, W, False, DEV, rcf_amd.RCFModel)
    tr = rcf_amd.Trainer(model, device=DEV)
    batch = _batch(B, H, W, DEV)
    model.train()
    with torch.no_grad():
        from rcf_amd.layers import Tape
        t = Tape(enabled=False)
        saved = copy.deepcopy(model.state_dict())
        model._select_precision()                           # fp32 (model.precision is None, no autocast): _images_nhwc reads it
        img = model._images_nhwc(torch.stack(batch["imgs"], dim=1))
        logits = model.decode_head2.fwd(model.backbone2.fwd(img, t), t)
        model.load_state_dict(saved)
    l_nchw = rcf_amd.ops.nhwc_to_nchw(logits.t)
    masks = torch.softmax(l_nchw.view(B, 2, 4, *l_nchw.shape[-2:]), dim=2).cpu().numpy()
    e_mask = float(np.abs(masks[0] - fx["truth_masks0"]).max())          # absolute, vs the float64 truth
    lim_mask = max(TOL, 4 * float(fx["ref32_err_masks"]))
    e_logit = rel(l_nchw[:2].cpu().numpy(), fx["truth_logits0"])
    lim_logit = max(TOL, 4 * float(fx["ref32_err_logits"]))
    e_mean = rel(masks.mean(axis=(3, 4)), fx["mask_mean"])
    am = l_nchw.argmax(1).cpu().numpy().astype(np.uint8)
    sure = fx["margin"].astype(np.float32) > TOL * float(fx["logit_absmax"])   # top-2 gap above the logits' own tolerance
    mism = int((am != fx["argmax"])[sure].sum())
    losses = tr.step(batch)
    e_loss = {k: rel(float(losses[k]), float(fx[k])) for k in ("loss", "loss_warp_seg", "loss_entropy")}
    gn = {}
    for n, p in model.named_parameters():
        gn[n.split(".")[0]] = gn.get(n.split(".")[0], 0.0) + float(p.grad.double().pow(2).sum())
    e_gn = {str(k): rel(np.sqrt(gn[str(k)]), v) for k, v in zip(fx["gradnorm_keys"], fx["truth_gradnorm"])}
    lim_gn = {str(k): max(TOL, 4 * float(v)) for k, v in zip(fx["gradnorm_keys"], fx["ref32_err_gradnorm"])}
    report(f"480x854 b1: logits vs f64 {e_logit:.2e} (limit {lim_logit:.2e}) masks |d| vs f64 {e_mask:.2e} (limit "
           f"{lim_mask:.2e}) mask means {e_mean:.2e} argmax mismatches (sure px) {mism} of {int(sure.sum())} "
           f"(unsure {int((~sure).sum())}) loss {e_loss} gradnorm vs f64 {e_gn} (limits {lim_gn})")
    assert e_logit < lim_logit and e_mask < lim_mask and e_mean < TOL and mism == 0
    assert max(e_loss.values()) < TOL
    assert all(e_gn[k] < lim_gn[k] for k in e_gn), (e_gn, lim_gn)


@pytest.mark.parametrize("tag", ["small", "480x854"])
def test_benched_config_dropout_syncbn_vs_reference_golden(tag, golden_dir, report):
    """Parity ON THE CONFIGURATION bench.py RUNS: norm SyncBN and Dropout2d 0.1 in decode_head2 /
    decode_head3 (models/decode_head.py:84-87, models/fcn_head.py:142-147).  The draw is injected: the reference model took it as
    its `dropout` module when tests/golden/make_golden_dropout.py ran, this path takes it as FCNHead.keep_mask -- the scale then
    travels through the last norm's apply pass, its backward bound, the thin classifier convs and `pair_concat`, none of which
    the dropout-free fixtures exercise.  Criteria as everywhere: losses / logits 1e-4 of the reference, arg-max equal on every
    pixel whose top-2 margin exceeds 1e-4 of the logits' range, module gradient norms and sampled gradients against float64
    within 4x the reference's own fp32 spread."""
    import json
    fx = json.load(open(os.path.join(golden_dir, "dropout.json")))[tag]
    arr = np.load(os.path.join(golden_dir, "dropout.npz"))
    H, W, B, C = fx["H"], fx["W"], fx["B"], fx["C"]
    model = _build(H, W, False, DEV, rcf_amd.RCFModel, benched=B)
    assert model.decode_head2.dropout_ratio == DROP_P and model.backbone2.bn1.sync
    assert int((model.decode_head2.keep_mask == 0).sum()) == fx["dropped_head2"]
    assert int((model.decode_head3.keep_mask == 0).sum()) == fx["dropped_head3"]
    tr = rcf_amd.Trainer(model, lr=1e-4, weight_decay=1e-4, device=DEV)
    losses = tr.step(_batch(B, H, W, DEV))
    e_loss = {k: rel(float(losses[k]), v) for k, v in fx["loss_fp32"].items()}
    z = rcf_amd.ops.nhwc_to_nchw(model.last_logits, C).cpu().numpy()
    am = z.argmax(1).astype(np.uint8)
    sure = arr[tag + "_margin_fp32"].astype(np.float32) > TOL
    mism = int((am != arr[tag + "_argmax_fp32"])[sure].sum())
    if tag == "small":
        e_logits = rel(z, arr["small_logits_fp32"])
        e_logits64 = rel(z, arr["small_logits_f64"])
    else:
        e_logits = rel(torch.softmax(torch.from_numpy(z), dim=1).mean(dim=(2, 3)).numpy(), arr[tag + "_mask_mean_fp32"])
        e_logits64 = rel(z[:1], arr[tag + "_logits_f64_0"])
    named = dict(model.named_parameters())
    gn = {}
    for n, p in named.items():
        if p.grad is not None:
            gn[n.split(".")[0]] = gn.get(n.split(".")[0], 0.0) + float(p.grad.double().pow(2).sum())
    e_gn = {k: rel(np.sqrt(gn[k]), v) for k, v in fx["gradnorm_f64"].items()}
    lim_gn = {k: max(TOL, 4 * v) for k, v in fx["ref32_err_gradnorm"].items()}
    e_grad, lim_grad = {}, {}
    for i, (name, err) in enumerate(fx["ref32_err_grad"].items()):
        g = named[name].grad.detach().cpu().contiguous().numpy().ravel()[:256]
        e_grad[name] = rel(g, arr[f"{tag}_truth_grad_{i}"])
        lim_grad[name] = max(TOL, 4 * err)
    report(f"benched configuration (SyncBN, Dropout2d draw injected: {fx['dropped_head2']} + {fx['dropped_head3']} planes dropped) "
           f"[{tag}] {H}x{W} B={B} vs the reference: loss {e_loss} logits {'' if tag == 'small' else '(mask means) '}{e_logits:.2e} "
           f"vs f64 {e_logits64:.2e} (ref32 {fx['ref32_err_logits']:.2e}) argmax mismatches (sure px) {mism} of {int(sure.sum())} "
           f"gradnorm vs f64 {e_gn} (limits {lim_gn}) grad vs f64 {e_grad} (limits {lim_grad})")
    assert max(e_loss.values()) < TOL and e_logits < TOL and mism == 0
    assert e_logits64 < max(TOL, 4 * fx["ref32_err_logits"])
    assert all(e_gn[k] < lim_gn[k] for k in e_gn), (e_gn, lim_gn)
    assert all(e_grad[k] < lim_grad[k] for k in e_grad), (e_grad, lim_grad)
    # the draw matters: the same step without it lands elsewhere (guards against a keep_mask that is silently ignored)
    plain = _build(H, W, False, DEV, rcf_amd.RCFModel)
    l0 = float(rcf_amd.Trainer(plain, device=DEV).step(_batch(B, H, W, DEV))["loss"])
    assert abs(l0 - float(losses["loss"])) / abs(l0) > 1e-4


@pytest.mark.parametrize("variant", list(config.VARIANTS))
def test_config_variants_vs_reference_golden(variant, golden_dir, report):
    """Configuration variants of the training step against numbers captured from the reference
    (tests/golden/variants.json, written by make_golden_variants.py): FBMS (3 segments + affine), STv2
    (single-map head + compactness), both sharpen-loss branches, joint residual head, compactness on the
    object channel.  Losses: 1e-4 relative to the reference; gradient norms: against the float64 truth with
    4x the reference's own fp32 error as the limit (floor 1e-4)."""
    import json
    fx = json.load(open(os.path.join(golden_dir, "variants.json")))[variant]
    H, W, B = fx["H"], fx["W"], fx["B"]
    kw, oc = config.variant_model_kwargs(variant, H, W)
    assert oc == fx["object_channel"]
    args = _args()
    args.object_channel = oc
    m = rcf_amd.RCFModel(args, **copy.deepcopy(kw))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=fx["weight_seed"]).items()})
    tr = rcf_amd.Trainer(m.to(DEV), device=DEV)
    lh = tr.step(_batch(B, H, W, DEV))
    assert sorted(k for k in lh if "loss" in k) == sorted(fx["loss"]), (sorted(lh), sorted(fx["loss"]))
    e = {k: rel(float(lh[k]), v) for k, v in fx["loss"].items()}
    gn = {}
    for n, p in m.named_parameters():
        if p.grad is not None:
            gn[n.split(".")[0]] = gn.get(n.split(".")[0], 0.0) + float(p.grad.double().pow(2).sum())
    e_gn = {k: rel(np.sqrt(gn[k]), v) for k, v in fx["truth_gradnorm"].items()}
    lim = {k: max(TOL, 4 * v) for k, v in fx["ref32_err_gradnorm"].items()}
    report(f"{variant} vs reference: " + " ".join(f"{k} {v:.2e}" for k, v in e.items()) + " | gradnorm vs f64 " +
           " ".join(f"{k} {v:.2e} (lim {lim[k]:.1e})" for k, v in e_gn.items()))
    # total loss against the reference; every term against the float64 truth (a hinge over a handful of active
    # pixels -- sharpen_obj -- is ill-conditioned in fp32: the reference itself is 2.4e-3 off there)
    assert e["loss"] < TOL
    e64 = {k: rel(float(lh[k]), v) for k, v in fx["truth_loss"].items()}
    assert all(e64[k] < max(TOL, 4 * fx["ref32_err_loss"][k]) for k in e64), (e64, fx["ref32_err_loss"])
    assert all(e_gn[k] < lim[k] for k in e_gn), (e_gn, lim)


def test_commuted_upsample_conv_matches_plain_path(report):
    """decode_head2 with the 2x up-sampling commuted past the dilated conv (layers.commuted_concat_conv) against the
    plain resize-concat-conv path of the same module: logits, input gradients and parameter gradients"""
    from rcf_amd.backbone import FCNHead
    from rcf_amd.layers import Act, Tape
    g = torch.Generator().manual_seed(3)
    N, h, w = 2, 40, 58
    head = FCNHead([64, 128], 32, num_classes=4, num_convs=2, concat_input=False, dilation=6, in_index=[0, 1],
                   input_transform="resize_concat", dropout_ratio=0.0, norm_cfg=dict(type="BN", requires_grad=True),
                   align_corners=False).to(DEV).train()
    with torch.no_grad():
        for p in head.parameters():
            p.copy_(torch.randn(p.shape, generator=g).to(DEV) * 0.1 + (1.0 if p.dim() == 1 else 0.0))
    fa = torch.randn(N, h, w, 64, generator=g).to(DEV)
    fb = torch.randn(N, h // 2, w // 2, 128, generator=g).to(DEV)
    dl = torch.randn(N, h, w, 4, generator=g).to(DEV)
    res = {}
    for fast in (False, True):
        head.commute_upsample = fast
        for p in head.parameters():
            p.grad = None
        a, b, tape = Act(fa.clone()), Act(fb.clone()), Tape()
        out = head.fwd([a, b], tape)
        out.grad = dl.clone()
        tape.backward()
        res[fast] = dict(out=out.t.clone(), ga=a.grad.clone(), gb=b.grad.clone(),
                         **{"p." + n: p.grad.clone() for n, p in head.named_parameters()})
    e = {k: rel(res[True][k].cpu().numpy(), res[False][k].cpu().numpy()) for k in res[True]}
    report("commuted upsample conv vs plain path: " + " ".join(f"{k} {v:.2e}" for k, v in e.items()))
    assert max(e.values()) < 5e-5, e


def test_train_step_bitwise_reproducible(report):
    """the same step from the same state twice: identical losses and gradients, bit for bit (no floating-point atomics
    on the training path: fixed-order split-K and batch-norm reductions, integer max for the operand ranges; the
    second HIP stream changes the order kernels run in, not what they compute)"""
    H, W, B = 96, 160, 2
    out = []
    for _ in range(2):
        model = _build(H, W, False, DEV, rcf_amd.RCFModel)
        tr = rcf_amd.Trainer(model, lr=1e-4, weight_decay=1e-4, device=DEV)
        losses = tr.step(_batch(B, H, W, DEV))
        torch.cuda.synchronize()
        out.append(({k: float(v) for k, v in losses.items()},
                    {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None},
                    {n: p.detach().clone() for n, p in model.named_parameters()}))
    same_loss = out[0][0] == out[1][0]
    bad_g = [n for n in out[0][1] if not torch.equal(out[0][1][n], out[1][1][n])]
    bad_p = [n for n in out[0][2] if not torch.equal(out[0][2][n], out[1][2][n])]
    report(f"two identical steps: losses identical {same_loss}, gradients differing {len(bad_g)} of {len(out[0][1])}, "
           f"updated parameters differing {len(bad_p)}")
    assert same_loss and not bad_g and not bad_p, (bad_g[:5], bad_p[:5])


def test_bn_buffers_after_two_steps_vs_oracle(report):
    """running_mean / running_var / num_batches_tracked of every batch norm after two optimiser steps: the conv's statistics
    reduction finalizes the norm on the device (rcf_conv2d_fwd_bnstats_*) -- against the oracle model taking the same two
    steps with torch's BatchNorm2d and Adam"""
    import rcf_torch as orc
    H, W, B = 96, 160, 2
    hip = _build(H, W, False, DEV, rcf_amd.RCFModel)
    ora = _build(H, W, False, "cpu", orc.RCFModel)
    tr = rcf_amd.Trainer(hip, lr=1e-4, weight_decay=1e-4, device=DEV)
    opt = orc.make_optimizer(ora, 1e-4, 1e-4)
    ora.train()
    for i in range(2):
        tr.step(_batch(B, H, W, DEV))
        losses = ora(_batch(B, H, W, "cpu"))
        opt.zero_grad()
        losses["loss"].backward()
        opt.step()
    hs, os_ = hip.state_dict(), ora.state_dict()
    worst_m = worst_v = 0.0
    n = 0
    for k, v in os_.items():
        if k.endswith("num_batches_tracked"):
            assert int(hs[k]) == int(v) == 2, (k, int(hs[k]), int(v))
            n += 1
        elif k.endswith("running_mean"):
            worst_m = max(worst_m, float((hs[k].cpu() - v).abs().max() / (v.abs().max() + 1e-6)))
        elif k.endswith("running_var"):
            worst_v = max(worst_v, rel(hs[k].cpu().numpy(), v.numpy()))
    report(f"batch-norm buffers after 2 steps ({n} norms): running_mean {worst_m:.2e} running_var {worst_v:.2e}, num_batches_tracked exact")
    assert n >= 50 and worst_m < 2e-3 and worst_v < 2e-3


def _mem_available_gb():
    try:
        return [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 1e6
    except Exception:                                           # noqa: BLE001
        return 0.0


def _module_gradnorms(model):
    gn = {}
    for n, p in model.named_parameters():
        if p.grad is not None:
            gn[n.split(".")[0]] = gn.get(n.split(".")[0], 0.0) + float(p.grad.double().pow(2).sum())
    return {k: v ** 0.5 for k, v in gn.items()}


def test_fullsize_b8_losses_and_argmax_vs_oracle(report):
    """BASELINE configs[1] at its real batch: 8 pairs of 480x854 (16 frames through the backbone, batch-norm statistics over
    all of them) -- every loss term of one training-mode forward, the logits and the segment arg-max of all 16 frames against
    the oracle on the host in fp32 (the reference-generated fixture of this geometry holds one pair; the oracle is pinned to
    the reference on the small cases and on that pair).  Arg-max: identical on every pixel whose top-2 margin exceeds
    1e-4 x max |logit| (the criterion of the small cases).  The bf16 step (BASELINE configs[2] on one rank) on the same batch
    against the same oracle losses, at the reference's own autocast-vs-fp32 deviation of this geometry (tests/golden/bf16.json)."""
    import json
    import rcf_torch as orc
    H, W, B = 480, 854, 8
    ora = _build(H, W, False, "cpu", orc.RCFModel)
    ora.train()
    grabbed = {}
    hook = ora.decode_head2.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o.detach()))
    with torch.no_grad():
        lo = ora(_batch(B, H, W, "cpu"))
    hook.remove()
    lo = {k: float(v) for k, v in lo.items()}
    lg_o = grabbed["logits"].float().numpy()                           # [16, 4, 120, 214]
    del ora
    out, lg_h = {}, None
    for prec in ("fp32", "bf16"):
        hip = _build(H, W, False, DEV, rcf_amd.RCFModel)
        hip.precision = prec
        hip.train()
        with torch.no_grad():
            lh = hip(_batch(B, H, W, DEV))
        out[prec] = {k: float(v) for k, v in lh.items()}
        if prec == "fp32":
            lg_h = rcf_amd.ops.nhwc_to_nchw(hip.last_logits).cpu().numpy()
        del hip
        torch.cuda.empty_cache()
    e32 = {k: rel(out["fp32"][k], lo[k]) for k in lo}
    e16 = {k: rel(out["bf16"][k], lo[k]) for k in lo}
    e_logits = rel(lg_h, lg_o)
    top2 = np.sort(lg_o, axis=1)[:, -2:]
    sure = (top2[:, 1] - top2[:, 0]) > TOL * float(np.abs(lg_o).max())
    mism = int((lg_h.argmax(1) != lg_o.argmax(1))[sure].sum())
    report(f"480x854 b8 (16 frames): fp32 " + " ".join(f"{k} hip {out['fp32'][k]:.6f} oracle {lo[k]:.6f} ({e32[k]:.1e})" for k in lo) +
           f" | logits {e_logits:.2e}, arg-max mismatches on sure pixels {mism} of {int(sure.sum())} (unsure {int((~sure).sum())})"
           " | bf16 " + " ".join(f"{k} {e16[k]:.1e}" for k in lo))
    assert all(np.isfinite(v) for v in out["fp32"].values()) and max(e32.values()) < TOL
    assert e_logits < TOL and mism == 0
    ref16 = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16.json")))["480x854"]["ref_bf16_vs_fp32"]
    assert all(e16[k] < max(3 * ref16["loss"][k], 5e-3) for k in e16), e16


def test_fullsize_b8_gradients_vs_oracle(report):
    """the backward of the same batch against the oracle's FLOAT64 backward of this very batch (tests/golden/oracle_b8_selfdev.json,
    made by tools/oracle_b8_selfdev.py on the GPU box's host: the float64 step needs ~90 GB and minutes, so it is a committed
    fixture, not a live run; the fixture's loss ties it to the batch).
    The step is ill-conditioned at this size -- 57 train-mode batch norms, ReLU / max-pool / arg-max decisions: EVERY fp32
    evaluation's gradient VECTOR is ~1.5e-2 (backbone) from the float64 one, the oracle's own included, and its norm then lands
    anywhere within a fraction of that (the exact-fp32 HIP kernels in another summation order: 1.9e-3; the oracle on images
    perturbed by one ulp: 2e-4 at 8 pairs, 1.4e-3 at one; tools/grad_error_b8.py, tools/grad_bias_probe.py, profiles/r04_grad_*).
    So the check that means something is on the vectors: per module, the HIP step's relative vector error against float64
    (from `synth.grad_sketch` fingerprints: 32 signed sums per parameter tensor) must be within 1.5 x the WORST of the oracle's
    own fp32 evaluations (as is / images perturbed by one fp32 ulp, two seeds), floor 1e-4; the norms are reported beside the
    oracle's norm spread and must at least stay inside the oracle's vector error (a norm error is one projection of it).
    bf16: the reference's own autocast-vs-fp32 deviation of this geometry (limits 3x, floor 10 %)."""
    import json
    from rcf_amd import synth as _synth
    H, W, B = 480, 854, 8
    here = os.path.dirname(os.path.abspath(__file__))
    fx = json.load(open(os.path.join(here, "golden", "oracle_b8_selfdev.json")))
    assert (fx["B"], fx["H"], fx["W"]) == (B, H, W)
    go, spread, vec_ref, sk_truth = fx["gradnorm_f64"], fx["gradnorm_fp32_spread"], fx["vector_fp32_err"], fx["sketch_f64"]
    losses, out, vec = {}, {}, {}
    for prec in ("fp32", "bf16"):
        hip = _build(H, W, False, DEV, rcf_amd.RCFModel)
        hip.precision = prec
        hip.train()
        l = hip(_batch(B, H, W, DEV))
        l["loss"].backward()
        losses[prec] = float(l["loss"])
        out[prec] = _module_gradnorms(hip)
        if prec == "fp32":
            sk = _synth.grad_sketch({n: p.grad for n, p in hip.named_parameters() if p.grad is not None}, k=fx["sketch_k"])
            assert sorted(sk) == sorted(sk_truth), "parameter names differ from the oracle's"
            vec = {k: _synth.sketch_error(sk, sk_truth, k + ".") for k in go}
        del hip
        torch.cuda.empty_cache()
    assert rel(losses["fp32"], fx["loss_f64"]["loss"]) < TOL, (losses, fx["loss_f64"])       # same batch, same weights
    g32 = {k: rel(out["fp32"][k], v) for k, v in go.items()}
    g16 = {k: rel(out["bf16"][k], v) for k, v in go.items()}
    report("480x854 b8 gradients vs the oracle's float64: fp32 vector error (oracle's own fp32, worst of 3) " +
           " ".join(f"{k} {vec[k]:.1e} ({vec_ref[k]:.1e})" for k in go) + " | norm error (oracle's fp32 spread) " +
           " ".join(f"{k} {g32[k]:.1e} ({spread[k]:.1e})" for k in go) + " | bf16 norms " + " ".join(f"{k} {v:.1e}" for k, v in g16.items()))
    assert all(vec[k] < max(TOL, 1.5 * vec_ref[k]) for k in go), (vec, vec_ref)
    assert all(g32[k] < max(TOL, 4 * spread[k], vec_ref[k]) for k in go), (g32, spread, vec_ref)
    ref16 = json.load(open(os.path.join(here, "golden", "bf16.json")))["480x854"]["ref_bf16_vs_fp32"]
    assert all(g16[k] < max(3 * ref16["gradnorm"][k], 0.10) for k in g16), g16


def test_b8_gradients_vs_reference(golden_dir, report):
    """The backward at the REAL batch size against a REFERENCE-derived fixture (the 8 x 480x854 gradient
    fixture is made by the oracle alone).  tests/golden/b8_reference.json (make_golden_b8.py): the reference model itself on 8 pairs
    -- 16 frames through the backbone, batch-norm statistics over all of them -- at 192x320, the largest geometry whose float64
    truth fits the build container; its fp32 losses, and `synth.grad_sketch` fingerprints of the float64 truth and of the
    reference's own fp32 gradients (8 threads / 1 thread).  Criteria as at 480x854: losses 1e-4 of the reference; per module the
    HIP step's gradient VECTOR error against float64 within 1.5 x the reference's own fp32 vector error (floor 1e-4), its norm
    inside max(4 x the reference's norm error, the reference's vector error)."""
    import json
    fx = json.load(open(os.path.join(golden_dir, "b8_reference.json")))
    H, W, B = fx["H"], fx["W"], fx["B"]
    hip = _build(H, W, False, DEV, rcf_amd.RCFModel)
    hip.train()
    l = hip(_batch(B, H, W, DEV))
    l["loss"].backward()
    e_loss = {k: rel(float(l[k]), v) for k, v in fx["loss_ref_fp32"].items()}
    sk = synth.grad_sketch({n: p.grad for n, p in hip.named_parameters() if p.grad is not None}, k=fx["sketch_k"])
    assert sorted(sk) == sorted(fx["sketch_f64"]), "parameter names differ from the reference's"
    mods = sorted(fx["gradnorm_f64"])
    vec = {k: synth.sketch_error(sk, fx["sketch_f64"], k + ".") for k in mods}
    vec_vs_ref = {k: synth.sketch_error(sk, fx["sketch_ref_fp32"], k + ".") for k in mods}
    gn = _module_gradnorms(hip)
    e_gn = {k: rel(gn[k], fx["gradnorm_f64"][k]) for k in mods}
    vr, nr = fx["ref_fp32_vector_err"], fx["ref_fp32_norm_err"]
    report(f"{H}x{W} b8 against the REFERENCE: losses " + " ".join(f"{k} {v:.1e}" for k, v in e_loss.items()) +
           " | gradient vector error vs float64 (the reference's own fp32) " + " ".join(f"{k} {vec[k]:.1e} ({vr[k]:.1e})" for k in mods) +
           " | vs the reference's fp32 gradients " + " ".join(f"{k} {vec_vs_ref[k]:.1e}" for k in mods) +
           " | norm error vs float64 (the reference's) " + " ".join(f"{k} {e_gn[k]:.1e} ({nr[k]:.1e})" for k in mods))
    assert max(e_loss.values()) < TOL
    assert all(vec[k] < max(TOL, 1.5 * vr[k]) for k in mods), (vec, vr)
    assert all(e_gn[k] < max(TOL, 4 * nr[k], vr[k]) for k in mods), (e_gn, nr, vr)
