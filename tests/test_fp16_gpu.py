"""GPU: IEEE fp16 as the 16-bit storage type (librcf_hip_f16.so: csrc/rcf_common.h RCF_HALF_F16) -- Lightning's `precision: 16`
of the STv2 / FBMS configs (fp16 autocast + GradScaler, configs/rcf_stv2/rcf_stage1.yaml:57-60), which rounds 2-5 ran as bf16
storage.  Same kernels as the bf16 step with the other MFMA instruction and widening conversion, so the checks are: the kernels
against float64 (and closer to it than the bf16 build on the same data: 10 significand bits against 7), the training step against
the fp32 step and against the REFERENCE's own fp16-autocast deviation (tests/golden/variants_autocast.json), the loss-scaling
policy, torch's own GradScaler + Adam through the autograd bridge, and the three storage types side by side in one process."""
import copy
import json
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import rcf_amd
from rcf_amd import config, layers, ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H16 = {"bf16": torch.bfloat16, "fp16": torch.float16}


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _conv_errors(half, case, seed):
    """forward / data gradient / weight gradient of one conv on operands representable in BOTH 16-bit types, against float64"""
    N, Cin, Cout, k, stride, pad, dil, H, W = case
    g = torch.Generator().manual_seed(seed)
    both = lambda t: t.to(torch.bfloat16).to(torch.float16).float()          # 7 significand bits, fp16's range: exact in both
    x = both(torch.randn(N, Cin, H, W, generator=g))
    w = both(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    y64 = F.conv2d(x.double(), w.double(), None, stride, pad, dil)
    dy = both(torch.randn(y64.shape, generator=g))
    dx64 = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), stride, pad, dil)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), stride, pad, dil)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(half).to(DEV)
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last)
    with ops.half_storage(half):
        y = ops.conv2d_fwd_bf16(nhwc(x), wd, None, None, stride, pad, dil)
        dx = ops.conv2d_dgrad_bf16(nhwc(dy), wd, (N, H, W, Cin), stride, pad, dil)
        dw = torch.zeros_like(wd)
        ops.conv2d_wgrad_bf16(nhwc(x), nhwc(dy), wd, dw, stride, pad, dil, beta=0)
        assert y.dtype == half and dx.dtype == half
    return (relerr(y.float().permute(0, 3, 1, 2), y64), relerr(dx.float().permute(0, 3, 1, 2), dx64), relerr(dw, dw64))


@pytest.mark.parametrize("case", [(2, 256, 256, 3, 1, 2, 2, 30, 41), (2, 256, 1024, 1, 1, 0, 1, 33, 29), (2, 128, 128, 3, 2, 1, 1, 31, 45),
                                  (1, 64, 64, 3, 1, 1, 1, 30, 53)])
def test_fp16_convs_vs_float64(case, report):
    e16, eb = _conv_errors(torch.float16, case, 5), _conv_errors(torch.bfloat16, case, 5)
    report(f"16-bit convs {case} vs float64 (fwd / dgrad / wgrad): fp16 build {e16[0]:.1e} {e16[1]:.1e} {e16[2]:.1e}; bf16 build "
           f"{eb[0]:.1e} {eb[1]:.1e} {eb[2]:.1e}")
    # outputs rounded to fp16 (2^-11) resp. bf16 (2^-8); the weight gradient is fp32 in both
    assert e16[0] < 1e-3 and e16[1] < 1e-3 and e16[2] < 2e-5 and eb[2] < 2e-5
    assert e16[0] < 0.3 * eb[0] and e16[1] < 0.3 * eb[1]


def test_fp16_batchnorm_and_spatial_passes(report):
    """the streaming passes on fp16 tensors against their fp32 forms on the same (fp16-representable) values"""
    g = torch.Generator().manual_seed(3)
    N, H, W, C = 2, 23, 31, 64
    q = lambda t: t.to(torch.float16).float()
    x, res, dy = q(torch.randn(N, H, W, C, generator=g) * 2 + 0.3), q(torch.randn(N, H, W, C, generator=g)), q(torch.randn(N, H, W, C, generator=g))
    gam, bet = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    e = {}
    with ops.half_storage(torch.float16):
        xf, rf, df = x.to(DEV), res.to(DEV), dy.to(DEV)
        xh, rh, dh = xf.half(), rf.half(), df.half()
        sf, sh = ops.bn_stats(xf), ops.bn_stats(xh)
        e["stats"] = relerr(sh, sf)
        mean, invstd = ops.bn_finalize(sf, N * H * W, 1e-5, 0.1)
        yf = ops.bn_apply(xf, mean, invstd, gam, bet, True, residual=rf)
        yh = ops.bn_apply(xh, mean, invstd, gam, bet, True, residual=rh)
        assert yh.dtype == torch.float16
        e["apply"] = relerr(yh.float(), yf)
        s2f = ops.bn_bwd_reduce(df, xf, yf, mean, invstd, True)
        s2h = ops.bn_bwd_reduce(dh, xh, yh, mean, invstd, True)
        e["bwd_sums"] = relerr(s2h, s2f)
        pf = [torch.zeros(C, device=DEV) for _ in range(4)]
        dxf = ops.bn_bwd_apply(df, xf, yf, mean, invstd, gam, True, s2f, N * H * W, pf[0], pf[1])
        dxh = ops.bn_bwd_apply(dh, xh, yh, mean, invstd, gam, True, s2f, N * H * W, pf[2], pf[3])
        e["bwd_apply"] = relerr(dxh.float(), dxf)
        mf, af = ops.maxpool_fwd(xf)
        mh, ah = ops.maxpool_fwd(xh)
        e["maxpool"] = relerr(mh.float(), mf)
        of, oh = torch.zeros(N, 2 * H, 2 * W, C, device=DEV), torch.zeros(N, 2 * H, 2 * W, C, dtype=torch.float16, device=DEV)
        ops.resize_nhwc_fwd(xf, (2 * H, 2 * W), False, out=of)
        ops.resize_nhwc_fwd(xh, (2 * H, 2 * W), False, out=oh)
        e["resize"] = relerr(oh.float(), of)
        e["cast"] = relerr(ops.cast(ops.cast(xf, torch.float16), torch.float32), xf)
    report("fp16 streaming passes vs fp32 on the same values: " + " ".join(f"{k} {v:.1e}" for k, v in e.items()))
    assert e["stats"] < 1e-12 and e["maxpool"] == 0 and e["cast"] == 0 and e["bwd_sums"] < 1e-3
    assert max(e["apply"], e["bwd_apply"], e["resize"]) < 1.2e-3           # one rounding to 11 bits


def _model_and_batch(H, W, B, variant=None):
    if variant is None:
        kw, oc = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, norm="BN"), None
    else:
        kw, oc = config.variant_model_kwargs(variant, H, W)
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_fp16", object_channel=oc, eval_save=False, eval_export=False)
    m = rcf_amd.RCFModel(args, **copy.deepcopy(kw))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
    nb = synth.make_batch(B, H, W, config_id=1)
    batch = {k: [torch.from_numpy(np.ascontiguousarray(x)).to(DEV) for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
    return m, batch


def _gradnorms(m, scale=1.0):
    gn, finite = {}, True
    for n, p in m.named_parameters():
        if p.grad is not None:
            finite &= bool(torch.isfinite(p.grad).all())
            gn[n.split(".")[0]] = gn.get(n.split(".")[0], 0.0) + float((p.grad.double() / scale).pow(2).sum())
    return {k: v ** 0.5 for k, v in gn.items()}, finite


def test_fp16_step_tracks_fp32_and_the_scaler_steps(report):
    """one model in fp32, bf16 and fp16 storage on the same batch: the fp16 step's losses are closer to the fp32 step's than the
    bf16 step's; Trainer(precision="fp16") scales the loss (LossScaler = torch GradScaler's policy), backs off on a non-finite
    gradient, then trains (loss falls, everything finite)"""
    Hh, Ww, B = 64, 96, 2
    res = {}
    for prec in ("fp32", "bf16", "fp16"):
        m, batch = _model_and_batch(Hh, Ww, B)
        tr = rcf_amd.Trainer(m, device=DEV, precision=prec)
        if prec == "fp16":
            assert tr.scaler is not None and tr.scaler.scale == 2.0 ** 16        # GradScaler's default, as Lightning builds it
        ls = [float(tr.step(batch)["loss"]) for _ in range(12)]
        res[prec] = (ls, tr)
    d16 = abs(res["fp16"][0][0] - res["fp32"][0][0]) / abs(res["fp32"][0][0])
    db = abs(res["bf16"][0][0] - res["fp32"][0][0]) / abs(res["fp32"][0][0])
    sc = res["fp16"][1].scaler
    report(f"first-step loss vs the fp32 step: fp16 storage {d16:.2e}, bf16 storage {db:.2e}; fp16 trainer over 12 steps: losses "
           f"{[round(v, 3) for v in res['fp16'][0]]} (fp32: {[round(v, 3) for v in res['fp32'][0]]}), loss scale 2^{np.log2(sc.scale):.0f}, "
           f"{sc.skipped} skipped step(s), {res['fp16'][1].step_count} optimizer steps")
    assert d16 < 2e-3 and d16 < db
    assert all(np.isfinite(v) for v in res["fp16"][0]) and res["fp16"][0][-1] < res["fp16"][0][0]
    # this randomly initialised net at 64x96 has large gradients: the scale walks down from 2^16 (the reference's own fp16 run
    # overflows at 2^14: variants_autocast.json) and then the steps are taken
    assert res["fp16"][1].step_count + sc.skipped == 12 and res["fp16"][1].step_count >= 3 and sc.scale >= 2.0 ** 6
    assert res["fp16"][1].model._act_dtype == torch.float16


@pytest.mark.parametrize("variant", ["stv2", "fbms"])
def test_fp16_autocast_vs_reference_fp16_autocast(variant, golden_dir, report):
    """the STv2 / FBMS variants (the configs the reference trains with `precision: 16`) entered the way Lightning enters them --
    torch.autocast(float16), the loss times a GradScaler-style scale handed to backward -- now store fp16.  Yardstick: the
    REFERENCE's own fp16-autocast step of the variant (variants_autocast.json: loss terms 0.04-0.5 % from its fp32 run; its
    backbone gradient overflows at a scale of 2^14 -- the step GradScaler skips -- so here the scale backs off like GradScaler's
    until the gradients are finite).  Loss terms within 3x the reference's largest fp16 loss-term deviation (floor 2e-3); module
    gradient norms (unscaled, against float64) within max(3x the reference's 16-bit deviation of that module, 35 %) as in
    test_stv2_variant_under_autocast_precision -- its fp16 deviation, or where its fp16 run overflowed (NaN) its bf16 deviation: a
    64x96 batch of 2 is noise-dominated in 16 bits (the reference's own runs move these norms by 4-89 %)."""
    fx = json.load(open(os.path.join(golden_dir, "variants.json")))[variant]
    fa = json.load(open(os.path.join(golden_dir, "variants_autocast.json")))[variant]
    ref = fa["ref_autocast_vs_fp32"]["fp16"]
    Hh, Ww, B = fx["H"], fx["W"], fx["B"]
    scale, tries = 2.0 ** 14, 0
    while True:
        m, batch = _model_and_batch(Hh, Ww, B, variant)
        m.to(DEV).train()
        with torch.autocast("cuda", dtype=torch.float16):
            losses = m(batch)
        assert m._act_dtype == torch.float16, "autocast(float16) must select fp16 storage (SCHED.autocast_fp16_as_bf16 is off)"
        (losses["loss"] * scale).backward()
        gn, finite = _gradnorms(m, scale)
        if finite or tries >= 8:
            break
        scale, tries = scale * 0.5, tries + 1
    e = {k: abs(float(losses[k]) - v) / abs(v) for k, v in fx["loss"].items()}
    e_gn = {k: abs(gn[k] - v) / v for k, v in fx["truth_gradnorm"].items()}
    ref_b = fa["ref_autocast_vs_fp32"]["bf16"]["gradnorm"]
    ref_gn = {k: (v if v == v else ref_b[k]) for k, v in ref["gradnorm"].items()}            # NaN: the reference overflowed there
    report(f"{variant} under autocast(float16) -> fp16 storage, loss scale 2^{np.log2(scale):.0f} after {tries} back-off(s): losses vs the "
           "reference's fp32 " + " ".join(f"{k} {v:.1e}" for k, v in e.items()) + " | gradient norms vs float64 " +
           " ".join(f"{k} {v:.1e}" for k, v in e_gn.items()) + " | the reference's own fp16 autocast vs its fp32: losses " +
           " ".join(f"{k} {v:.1e}" for k, v in ref["loss"].items()) + " gradient norms " + " ".join(f"{k} {v:.1e}" for k, v in ref["gradnorm"].items()))
    assert finite and scale >= 2.0 ** 8
    lim_l = max(3 * max(ref["loss"].values()), 2e-3)
    assert all(v < lim_l for v in e.values()), (e, lim_l)
    assert all(e_gn[k] < max(3 * ref_gn[k], 0.35) for k in e_gn), (e_gn, ref["gradnorm"])


def test_fp16_autocast_with_torch_gradscaler_and_adam(report):
    """the unchanged main.py path at `precision: 16`: torch.autocast(float16) around the model, torch's GradScaler around
    torch.optim.Adam, `scaler.scale(loss).backward()` through the autograd bridge (main.py:158-178, 299-307)"""
    m, batch = _model_and_batch(64, 96, 2)
    m.to(DEV).train()
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 16, growth_interval=1000)
    ls, scales = [], []
    for _ in range(12):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            losses = m(batch)
        scaler.scale(losses["loss"]).backward()
        scaler.step(opt)
        scaler.update()
        ls.append(float(losses["loss"]))
        scales.append(scaler.get_scale())
    report(f"autocast(float16) + torch GradScaler + torch Adam through the bridge: losses {[round(v, 3) for v in ls]}, scale {[int(np.log2(s)) for s in scales]} (log2)")
    assert m._act_dtype == torch.float16 and all(np.isfinite(v) for v in ls) and ls[-1] < ls[0] and scales[-1] >= 2.0 ** 6


def test_three_storage_types_coexist_in_one_process(report):
    """fp32, bf16 and fp16 models take steps in alternation (two library builds in one process, ops.half_storage per pass) and each
    reproduces what it computes alone, bit for bit"""
    Hh, Ww, B = 64, 96, 2
    alone = {}
    for prec in ("fp32", "bf16", "fp16"):
        m, batch = _model_and_batch(Hh, Ww, B)
        tr = rcf_amd.Trainer(m, device=DEV, precision=prec, loss_scaler=rcf_amd.trainer.LossScaler(2.0 ** 8) if prec == "fp16" else None)
        alone[prec] = [float(tr.step(batch)["loss"]) for _ in range(2)]
    ms = {}
    for prec in ("fp32", "bf16", "fp16"):
        m, batch = _model_and_batch(Hh, Ww, B)
        ms[prec] = (rcf_amd.Trainer(m, device=DEV, precision=prec, loss_scaler=rcf_amd.trainer.LossScaler(2.0 ** 8) if prec == "fp16" else None), batch)
    mixed = {k: [] for k in ms}
    for _ in range(2):
        for prec in ("fp16", "bf16", "fp32"):
            tr, batch = ms[prec]
            mixed[prec].append(float(tr.step(batch)["loss"]))
    report(f"fp32 / bf16 / fp16 models interleaved: {mixed} alone: {alone}")
    assert mixed == alone and rcf_amd._lib.ACTIVE == "bf16"


def test_fp16_step_fullsize_vs_reference_fp32(golden_dir, report):
    """BASELINE's geometry (480x854, one pair) in fp16 storage against the REFERENCE's fp32 step (tests/golden/bf16.*): losses
    closer than the reference's own bf16-autocast run is, and far fewer arg-max decisions flipped than by that run (fp16 keeps 10
    significand bits, bf16 7) -- the statistic of test_bf16_step_vs_reference_autocast_golden, with the bf16 reference as the
    ceiling"""
    fx = json.load(open(os.path.join(golden_dir, "bf16.json")))["480x854"]
    arr = np.load(os.path.join(golden_dir, "bf16.npz"))
    Hh, Ww, B, C = fx["H"], fx["W"], fx["B"], fx["C"]
    m, batch = _model_and_batch(Hh, Ww, B)
    tr = rcf_amd.Trainer(m, device=DEV, precision="fp16", loss_scaler=rcf_amd.trainer.LossScaler(2.0 ** 8))
    losses = tr.step(batch)
    e_l = {k: abs(float(losses[k]) - v) / abs(v) for k, v in fx["loss_fp32"].items()}
    ref = fx["ref_bf16_vs_fp32"]
    z = ops.nhwc_to_nchw(m.last_logits, C).cpu()
    am32 = torch.from_numpy(arr["480x854_argmax_fp32"].astype(np.int64))
    flips = float((z.argmax(1) != am32).float().mean())
    gn, finite = _gradnorms(m, tr.scaler.scale if tr.scaler.skipped == 0 else 1.0)
    report(f"fp16 step at 480x854 B=1 vs the reference's fp32: losses " + " ".join(f"{k} {v:.1e}" for k, v in e_l.items()) +
           f" (the reference's bf16 autocast: " + " ".join(f"{v:.1e}" for v in ref["loss"].values()) + f"); arg-max decisions flipped "
           f"{flips:.4f} of the pixels (reference bf16 autocast {ref['argmax_mismatch_frac']:.4f}); gradients finite {finite}, {tr.scaler.skipped} skipped")
    assert all(e_l[k] < max(ref["loss"][k], 2e-3) for k in e_l), e_l
    assert flips < 0.5 * ref["argmax_mismatch_frac"] and finite
