"""fp16 pair planes (include/rcf_hip.h RCF_CONV_X_PLANES / RCF_CONV_DY_PLANES): the batch-norm passes write a conv's input and
its output gradient pre-split, the conv kernels take both operands by LDS-DMA (csrc/igemm_h2d.inc, igemm_h2dw.inc).
 * the plane kernels against the register-split kernels on the same values: the same three partial products in the same
   order -- BIT-identical (forward + fused statistics, data gradient overwrite / accumulate / strided, weight gradient), with the
   planes scaled by a loose upper bound instead of the exact range; and against float64;
 * the planes the batch norm writes: (h + m) 2^-k equals the fp32 output to 2^-21 of each element, the bound it leaves in
   the range slot really bounds the tensor, planes-only == planes of the both-mode, backward likewise;
 * one training step with the planes on and off: same losses, gradients at fp32 level (tests/test_model_gpu.py holds the
   default path -- planes on -- to the reference's fixtures)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import rcf_amd
from rcf_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def to_planes(x, bound_bits):
    """x [N,H,W,C] fp32 on the device -> the fp32-typed pair-plane buffer the kernels read, scaled like they scale: 2^k with
    k = 14 - floor(log2(bound))"""
    b = float(bound_bits.view(torch.float32))
    k = 14 - int(np.floor(np.log2(b)))
    t = x.double() * 2.0 ** k
    h = t.to(torch.float16)
    m = (t - h.double()).to(torch.float16)
    N, H, W, C = x.shape
    pl = torch.stack([h, m], dim=3).contiguous()                    # [N,H,W,2,C] fp16 = [pixel][h | m]
    return pl.view(torch.float32).reshape(N, H, W, C), k


def from_planes(buf, k):
    N, H, W, C = buf.shape
    hm = buf.contiguous().view(torch.float16).reshape(N, H, W, 2, C).double()
    return (hm[:, :, :, 0] + hm[:, :, :, 1]) * 2.0 ** -k


def cl_weight(w):
    return w.to(DEV).contiguous(memory_format=torch.channels_last)


def loose(amax, factor):
    return (amax.view(torch.float32) * factor).view(torch.int32)


@pytest.mark.parametrize("case", [
    # N, Cin, Cout, k, stride, pad, dil, H, W
    (2, 256, 256, 3, 1, 2, 2, 60, 107),      # layer3 conv2: 128 x 256 tiles, chunked K order
    (2, 256, 1024, 1, 1, 0, 1, 60, 107),     # conv3: 4 column tiles, 16 K-steps
    (2, 1024, 256, 1, 1, 0, 1, 33, 41),      # conv1: ragged row tile
    (3, 128, 128, 3, 2, 1, 1, 61, 107),      # layer2.0 conv2: stride 2 (strided data gradient), 128-wide tile
    (2, 64, 64, 3, 1, 1, 1, 30, 53),         # layer1 conv2: 64-wide tile, K = 576
    (2, 256, 64, 1, 1, 0, 1, 30, 53),        # layer1 conv1
    (1, 512, 2048, 1, 2, 0, 1, 30, 54),      # layer4-style downsample: 1x1 stride 2
])
def test_pair_plane_convs_match_register_split(case, report):
    N, Cin, Cout, k, stride, pad, dil, H, W = case
    g = torch.Generator().manual_seed(sum((i + 1) * v for i, v in enumerate(case)))
    x = torch.randn(N, H, W, Cin, generator=g).to(DEV)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    wg = cl_weight(w)
    Ho, Wo = ops.conv_out_size(H, k, stride, pad, dil), ops.conv_out_size(W, k, stride, pad, dil)
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).to(DEV)
    ax, aw, ag = ops.absmax(x), ops.absmax(ops.weight_rsck(wg)), ops.absmax(dy)
    wp, wpt = ops.weight_pairs(wg, aw), ops.weight_pairs_t(wg, aw)
    xn, dyn = x.permute(0, 3, 1, 2).double().cpu(), dy.permute(0, 3, 1, 2).double().cpu()
    yref = F.conv2d(xn, w.double(), None, stride, pad, dil)
    wref = torch.nn.grad.conv2d_weight(xn, tuple(w.shape), dyn, stride, pad, dil)
    y0, s0 = ops.conv2d_fwd_stats(x, wg, stride, pad, dil, amax=(ax, aw), w_pairs=wp)
    dx0 = ops.conv2d_dgrad(dy, wg, x.shape, stride, pad, dil, amax=(ag, aw), w_pairs_t=wpt)
    acc0 = dx0.clone()
    ops.conv2d_dgrad(dy, wg, x.shape, stride, pad, dil, out=acc0, beta=1, amax=(ag, aw), w_pairs_t=wpt)
    dw0 = torch.full_like(wg, 3.0)
    ops.conv2d_wgrad(x, dy, wg, dw0, stride, pad, dil, beta=0, amax=(ax, ag), small_tile=True)    # 128 x 256 tile: the same split-K plan
    dwa0 = dw0.clone()
    ops.conv2d_wgrad(x, dy, wg, dwa0, stride, pad, dil, beta=1, amax=(ax, ag), small_tile=True)
    msg = []
    # factor 1: planes at the scale the register-split kernels use -> the same h, m everywhere, also where m is an fp16 subnormal
    # (elements below 2^-18 of the range): BIT-identical.  Loose bounds, as the batch norm leaves them: the subnormal m of those
    # tiny elements are rounded at another place (2^-40 of the range) -- fp32-level agreement, float64 accuracy unchanged.
    for fx, fg in ((1.0, 1.0), (5.3, 1.9), (60.0, 33.0)):
        bx, bg = loose(ax, fx), loose(ag, fg)
        xp, _ = to_planes(x, bx)
        dyp, _ = to_planes(dy, bg)
        ya, da = ops.new_amax(DEV), ops.new_amax(DEV)
        y1, s1 = ops.conv2d_fwd_stats(xp, wg, stride, pad, dil, amax=(bx, aw), w_pairs=wp, x_planes=True, amax_y=ya)
        dx1 = ops.conv2d_dgrad(dyp, wg, x.shape, stride, pad, dil, amax=(bg, aw), w_pairs_t=wpt, dy_planes=True, amax_y=da)
        acc1 = dx1.clone()
        ops.conv2d_dgrad(dyp, wg, x.shape, stride, pad, dil, out=acc1, beta=1, amax=(bg, aw), w_pairs_t=wpt, dy_planes=True)
        dw1 = torch.full_like(wg, 3.0)
        ops.conv2d_wgrad(xp, dyp, wg, dw1, stride, pad, dil, beta=0, amax=(bx, bg), planes=True)
        dwa1 = dw1.clone()
        ops.conv2d_wgrad(xp, dyp, wg, dwa1, stride, pad, dil, beta=1, amax=(bx, bg), planes=True)
        assert int(ya) == int(ops.absmax(y1)) and int(da) == int(ops.absmax(dx1)), "the epilogue's range of its output"
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
        d = [rel(y1, y0), rel(dx1, dx0), rel(acc1, acc0), rel(dw1, dw0), rel(dwa1, dwa0)]
        e_s = rel(s1, s0)
        e_y = float((y1.permute(0, 3, 1, 2).double().cpu() - yref).abs().max() / yref.abs().max())
        e_w = float((dw1.cpu().double() - wref).abs().max() / wref.abs().max())
        msg.append(f"bounds x{fx:g}/x{fg:g}: vs register split fwd {d[0]:.1e} dgrad {d[1]:.1e} acc {d[2]:.1e} wgrad {d[3]:.1e} acc {d[4]:.1e} "
                   f"stats {e_s:.1e}; vs float64 fwd {e_y:.2e} wgrad {e_w:.2e}")
        if fx == 1.0:
            assert max(d) == 0.0, d
        assert max(d) < 1e-6 and e_s < 1e-7 and e_y < 2e-5 and e_w < 2e-5, (d, e_s, e_y, e_w)
    report(f"pair-plane convs {case}: " + " | ".join(msg))


@pytest.mark.parametrize("res", [False, True])
def test_batchnorm_writes_pair_planes(res, report):
    g = torch.Generator().manual_seed(11 + res)
    N, H, W, C = 2, 33, 41, 256
    x = (torch.randn(N, H, W, C, generator=g) * torch.exp2(4 * torch.rand(C, generator=g)) + 0.3).to(DEV)
    r = torch.randn(N, H, W, C, generator=g).to(DEV) * 3 if res else None
    dy = torch.randn(N, H, W, C, generator=g).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    count = N * H * W
    mean, invstd = ops.bn_finalize(ops.bn_stats(x), count, 1e-5, 0.1)
    ax, ar, ag = ops.absmax(x), (ops.absmax(r) if res else None), ops.absmax(dy)
    rm0 = torch.empty(x.numel() // 4, dtype=torch.uint8, device=DEV)
    y0 = ops.bn_apply(x, mean, invstd, gamma, beta, True, residual=r, relu_mask=rm0, amax_out=ops.new_amax(DEV))
    out = {}
    for only in (False, True):
        rm = torch.empty_like(rm0)
        bound, pl = ops.new_amax(DEV), torch.empty_like(x)
        y = ops.bn_apply(x, mean, invstd, gamma, beta, True, residual=r, relu_mask=rm, amax_out=bound, planes=pl, planes_only=only,
                         amax_x=ax, amax_res=ar)
        assert (y is None) == only and torch.equal(rm, rm0)
        if not only:
            assert torch.equal(y, y0)
        out[only] = (pl, bound)
    assert torch.equal(out[False][0], out[True][0]) and int(out[False][1]) == int(out[True][1])
    b = float(out[True][1].view(torch.float32))
    k = 14 - int(np.floor(np.log2(b)))
    dec = from_planes(out[True][0], k)
    ymax = float(y0.abs().max())
    # element-wise: 22 significand bits wherever the element is not tiny against the bound
    big = y0.abs().double() > b * 2.0 ** -16
    e_el = float(((dec - y0.double()).abs() / y0.abs().double().clamp_min(1e-30))[big].max())
    e_abs = float((dec - y0.double()).abs().max() / b)
    # backward
    s2 = ops.bn_bwd_reduce(dy, x, None, mean, invstd, True, relu_mask=rm0)
    dg0, db0, dg1, db1 = (torch.zeros(C, device=DEV) for _ in range(4))
    dres0 = torch.full_like(x, 0.25) if res else None
    dres1 = torch.full_like(x, 0.25) if res else None
    dx0 = ops.bn_bwd_apply(dy, x, None, mean, invstd, gamma, True, s2, count, dg0, db0, dres=dres0, res_beta=1 if res else 0, relu_mask=rm0,
                           amax_out=ops.new_amax(DEV))
    gb = ops.new_amax(DEV)
    dxp = ops.bn_bwd_apply(dy, x, None, mean, invstd, gamma, True, s2, count, dg1, db1, dres=dres1, res_beta=1 if res else 0, relu_mask=rm0,
                           amax_out=gb, dx_planes=True, amax_x=ax, amax_dy=ag)
    bb = float(gb.view(torch.float32))
    kb = 14 - int(np.floor(np.log2(bb)))
    ddec = from_planes(dxp, kb)
    dmax = float(dx0.abs().max())
    bigd = dx0.abs().double() > bb * 2.0 ** -16
    e_del = float(((ddec - dx0.double()).abs() / dx0.abs().double().clamp_min(1e-30))[bigd].max())
    report(f"batch norm pair planes (residual {res}): forward bound / max |y| = {b / ymax:.2f}, element error {e_el:.1e} (2^-21 = 4.8e-7), "
           f"absolute {e_abs:.1e} of the bound; backward bound / max |dx| = {bb / dmax:.2f}, element error {e_del:.1e}")
    assert b >= ymax and b < 64 * ymax and e_el < 2.0 ** -21 and e_abs < 2.0 ** -22
    assert bb >= dmax and bb < 64 * dmax and e_del < 2.0 ** -21
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1) and (not res or torch.equal(dres0, dres1))


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_join_normalises_its_residual_on_the_fly(dt, report):
    """rcf_bn_apply_res_mp: relu(bn3(z3) + bn_ds(z_ds)) in ONE pass over z3 and the RAW z_ds (a stage's downsample branch,
    models/resnet.py:293-294) is bit-identical to the two passes it replaces -- output, sign bits, and (fp32) the pair planes,
    whose bound starts from the raw residual's range."""
    g = torch.Generator().manual_seed(5)
    N, H, W, C = 2, 29, 37, 256
    tdt = torch.float32 if dt == "fp32" else torch.bfloat16
    z3 = (torch.randn(N, H, W, C, generator=g) * torch.exp2(3 * torch.rand(C, generator=g)) + 0.2).to(DEV).to(tdt)
    zd = (torch.randn(N, H, W, C, generator=g) * 2.5 - 0.7).to(DEV).to(tdt)
    ga3, be3 = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    gad, bed = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    count = N * H * W
    m3, i3 = ops.bn_finalize(ops.bn_stats(z3), count, 1e-5, 0.1)
    md, idd = ops.bn_finalize(ops.bn_stats(zd), count, 1e-5, 0.1)
    idt = ops.bn_apply(zd, md, idd, gad, bed, False)
    rm0, rm1 = (torch.empty(z3.numel() // 4, dtype=torch.uint8, device=DEV) for _ in range(2))
    y0 = ops.bn_apply(z3, m3, i3, ga3, be3, True, residual=idt, relu_mask=rm0)
    y1 = ops.bn_apply(z3, m3, i3, ga3, be3, True, residual=zd, relu_mask=rm1, res_norm=(md, idd, gad, bed))
    assert torch.equal(y0, y1) and torch.equal(rm0, rm1)
    msg = f"join with its residual normalised on the fly ({dt}): output and sign bits identical to the two passes"
    if dt == "fp32":
        ax, ar_raw, ar = ops.absmax(z3), ops.absmax(zd), ops.absmax(idt)
        b0, p0, b1, p1 = ops.new_amax(DEV), torch.empty_like(z3), ops.new_amax(DEV), torch.empty_like(z3)
        ops.bn_apply(z3, m3, i3, ga3, be3, True, residual=idt, relu_mask=rm0, amax_out=b0, planes=p0, planes_only=True, amax_x=ax, amax_res=ar)
        ops.bn_apply(z3, m3, i3, ga3, be3, True, residual=zd, relu_mask=rm1, amax_out=b1, planes=p1, planes_only=True, amax_x=ax,
                     amax_res=ar_raw, res_norm=(md, idd, gad, bed))
        f0, f1 = float(b0.view(torch.float32)), float(b1.view(torch.float32))
        k = 14 - int(np.floor(np.log2(f1)))
        dec = from_planes(p1, k)
        ymax = float(y0.abs().max())
        big = y0.abs().double() > f1 * 2.0 ** -16
        e_el = float(((dec - y0.double()).abs() / y0.abs().double().clamp_min(1e-30))[big].max())
        msg += f"; planes: bound {f1 / ymax:.2f} x max |y| (two passes: {f0 / ymax:.2f}), element error {e_el:.1e}"
        assert f1 >= ymax and f1 < 64 * ymax and e_el < 2.0 ** -21
    report(msg)


@pytest.mark.parametrize("dt,planes", [("fp32", False), ("fp32", True), ("bf16", False)])
def test_join_and_downsample_norm_share_their_backward_passes(dt, planes, report):
    """rcf_bn_bwd_reduce2_mp / rcf_bn_bwd_apply2_mp: the join of a stage's first block and its downsample norm take their backward of
    the same masked gradient in ONE reduction and ONE apply pass -- sums, both input gradients (fp32 / bf16 / pair planes with their
    bounds) and both norms' parameter gradients bit-identical to the four separate passes."""
    g = torch.Generator().manual_seed(9)
    N, H, W, C = 2, 31, 45, 256
    tdt = torch.float32 if dt == "fp32" else torch.bfloat16
    z3 = (torch.randn(N, H, W, C, generator=g) * 1.7 + 0.2).to(DEV).to(tdt)
    zd = (torch.randn(N, H, W, C, generator=g) * 2.5 - 0.7).to(DEV).to(tdt)
    dy = torch.randn(N, H, W, C, generator=g).to(DEV).to(tdt)
    ga3, gad = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    mask = torch.randint(0, 16, (N * H * W * C // 4,), generator=g, dtype=torch.uint8).to(DEV)
    count = N * H * W
    m3, i3 = ops.bn_finalize(ops.bn_stats(z3), count, 1e-5, 0.1)
    md, idd = ops.bn_finalize(ops.bn_stats(zd), count, 1e-5, 0.1)
    s3 = ops.bn_bwd_reduce(dy, z3, None, m3, i3, True, relu_mask=mask)
    sd = ops.bn_bwd_reduce(dy, zd, None, md, idd, True, relu_mask=mask)
    s4 = ops.bn_bwd_reduce2(dy, z3, zd, m3, i3, md, idd, mask)
    assert torch.equal(s4[:2 * C], s3) and torch.equal(s4[2 * C:], sd)
    ax3, axd, ag = ops.absmax(z3.float()), ops.absmax(zd.float()), ops.absmax(dy.float())
    kw = dict(dx_planes=True, amax_dy=ag) if planes else {}
    pg = [torch.zeros(C, device=DEV) for _ in range(8)]
    a0, a1, b0, b1 = (ops.new_amax(DEV) for _ in range(4))
    dx3 = ops.bn_bwd_apply(dy, z3, None, m3, i3, ga3, True, s3, count, pg[0], pg[1], relu_mask=mask, amax_out=a0,
                           amax_x=ax3 if planes else None, **kw)
    dxd = ops.bn_bwd_apply(dy, zd, None, md, idd, gad, True, sd, count, pg[2], pg[3], relu_mask=mask, amax_out=a1,
                           amax_x=axd if planes else None, **kw)
    e3, ed = torch.empty_like(z3), torch.empty_like(zd)
    ops.bn_bwd_apply2(dy, z3, m3, i3, ga3, mask, s4[:2 * C], count, pg[4], pg[5], e3, zd, md, idd, gad, s4[2 * C:], pg[6], pg[7], ed,
                      amax_out=b0, amax_out2=b1, dx_planes=planes, amax_x=ax3 if planes else None, amax_x2=axd if planes else None,
                      amax_dy=ag if planes else None)
    assert torch.equal(e3, dx3) and torch.equal(ed, dxd)
    assert int(a0) == int(b0) and int(a1) == int(b1)
    assert all(torch.equal(pg[i], pg[i + 4]) for i in range(4))
    report(f"join + downsample norm, shared backward passes ({dt}, planes {planes}): sums, gradients, bounds and parameter gradients identical")


def test_training_step_with_and_without_pair_planes(report):
    """the same step with the planes on (default) and off (every conv splits its fp32 operands in registers): what changes is
    where the split happens and the tile of some weight gradients, not the arithmetic"""
    import copy
    import types
    from rcf_amd import config, layers, synth
    H, W, B = 96, 160, 2
    res = {}
    saved = layers.SCHED.planes
    try:
        for on in (True, False):
            layers.SCHED.planes = on
            kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, norm="BN")
            kw.update(log_interval=10 ** 9, train_iter=1)
            args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_test", object_channel=None, eval_save=False, eval_export=False)
            m = rcf_amd.RCFModel(args, **copy.deepcopy(kw))
            shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
            m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
            m.to(DEV)
            tr = rcf_amd.Trainer(m, lr=1e-4, weight_decay=1e-4, device=DEV)
            nb = synth.make_batch(B, H, W, config_id=1)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
            batch = {"imgs": [t(a) for a in nb["imgs"]], "gt_fw_flows": [t(a) for a in nb["gt_fw_flows"]],
                     "gt_bw_flows": [t(a) for a in nb["gt_bw_flows"]], "seq_ids": nb["seq_ids"], "seq_names": nb["seq_names"],
                     "paths": nb["paths"]}
            calls = {"n": 0}
            orig = ops.conv2d_wgrad

            def counting(*a, **k):
                calls["n"] += bool(k.get("planes"))
                return orig(*a, **k)
            ops.conv2d_wgrad = counting
            try:
                losses = tr.step(batch)
            finally:
                ops.conv2d_wgrad = orig
            gn = {}
            for n, p in m.named_parameters():
                if p.grad is not None:
                    gn[n.split(".")[0]] = gn.get(n.split(".")[0], 0.0) + float(p.grad.double().pow(2).sum())
            res[on] = ({k: float(v) for k, v in losses.items()}, {k: v ** 0.5 for k, v in gn.items()}, calls["n"])
    finally:
        layers.SCHED.planes = saved
    e_l = max(abs(res[True][0][k] - res[False][0][k]) / abs(res[False][0][k]) for k in res[True][0])
    e_g = {k: abs(res[True][1][k] - v) / v for k, v in res[False][1].items()}
    report(f"training step, pair planes on vs off: {res[True][2]} / {res[False][2]} weight gradients on the plane kernel; losses {e_l:.1e}; "
           "module gradient norms " + " ".join(f"{k} {v:.1e}" for k, v in e_g.items()))
    assert res[True][2] >= 40 and res[False][2] == 0
    assert e_l < 1e-5 and max(e_g.values()) < 2e-3


def test_training_step_lazy_norm_with_all_joins_as_planes(report):
    """with SCHED.join_planes = "all" a stage's first-block join writes pair planes AND (lazy downsample norm)
    normalises its residual on the fly -- the one place where bn_apply_kernel calls block_max_f twice in a row (the residual
    norm's bound, then its own), which raced on the shared scratch before block_max_f got its trailing barrier.  The step with
    the lazy norm on must be reproducible bit for bit (a race is timing) and agree with the step with it off (the two-pass form;
    the planes' bound starts from the raw residual's range there, so the scale 2^k may differ and the two are equal to rounding,
    not bit for bit) -- a plane scale off by a power of two in some workgroups would be a gross error."""
    import copy
    import types
    from rcf_amd import config, layers, synth
    H, W, B = 96, 160, 2
    res = {}
    old = layers.SCHED.set(join_planes="all")
    try:
        for lazy in (True, False):
            layers.SCHED.set(lazy_downsample_norm=lazy)
            kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, norm="BN")
            kw.update(log_interval=10 ** 9, train_iter=1)
            args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_test", object_channel=None, eval_save=False, eval_export=False)
            m = rcf_amd.RCFModel(args, **copy.deepcopy(kw))
            shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
            m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
            m.to(DEV)
            tr = rcf_amd.Trainer(m, lr=1e-4, weight_decay=1e-4, device=DEV)
            nb = synth.make_batch(B, H, W, config_id=1)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
            batch = {k: [t(a) for a in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
            runs = []
            for _ in range(3):                                  # a race shows as run-to-run differences too
                tr.fp.zero_grad()
                m.train()
                l = m(batch)
                l["loss"].backward()
                runs.append((float(l["loss"]), tr.fp.grad.clone()))
            assert all(r[0] == runs[0][0] and torch.equal(r[1], runs[0][1]) for r in runs[1:]), "the step is not reproducible"
            res[lazy] = runs[0]
    finally:
        layers.SCHED.set(**{k: old[k] for k in ("join_planes",)}, lazy_downsample_norm=True)
    e_l = abs(res[True][0] - res[False][0]) / abs(res[False][0])
    e_g = float((res[True][1] - res[False][1]).double().norm() / res[False][1].double().norm())
    report(f"join_planes='all': step with the downsample norm applied inside the join vs as its own pass: reproducible over 3 runs; "
           f"loss {e_l:.1e}, flat gradient {e_g:.1e} (norm-relative)")
    assert e_l < 1e-5 and e_g < 2e-3


@pytest.mark.parametrize("case", [
    # N, Cin, Cout, k, stride, pad, dil, H, W   (Cin = the batch norm's channels = the data gradient's output columns)
    (2, 256, 256, 3, 1, 2, 2, 60, 107),      # bn1 -> conv2 of layer3: 256-wide tile
    (2, 256, 1024, 1, 1, 0, 1, 33, 41),      # bn2 -> conv3: ragged row tile
    (2, 1024, 256, 1, 1, 0, 1, 33, 41),      # join -> the next block's conv1: four column tiles, accumulating
    (3, 128, 128, 3, 2, 1, 1, 61, 107),      # layer2.0 conv2: strided data gradient, 128-wide tile
    (2, 64, 64, 3, 1, 1, 1, 30, 53),         # layer1: 64-wide tile
    (1, 64, 256, 1, 1, 0, 1, 20, 30),        # few rows: the 64 x 64 tile of the register-split family
])
def test_data_gradient_delivers_batchnorm_backward_sums(case, report):
    """rcf_conv2d_dgrad_bnsums_f32: the data gradient whose output is the gradient of a batch norm + ReLU's output also delivers
    that norm's two backward sums (sum g, sum g xhat; g masked by the norm's sign bits) from its epilogue.  Against
    rcf_bn_bwd_reduce_mp run on the tensor the same data gradient wrote: dx BIT-identical to the plain launch, sums to fp32
    summation-order level (a lane's <= 64 values per column are added in fp32 here, in fp64 there) -- overwrite and accumulate,
    register-split and pair-plane operand, whole model shapes of every tile width."""
    N, Cin, Cout, k, stride, pad, dil, H, W = case
    g = torch.Generator().manual_seed(7 + sum((i + 1) * v for i, v in enumerate(case)))
    xbn = (torch.randn(N, H, W, Cin, generator=g) * torch.exp2(2 * torch.rand(Cin, generator=g)) + 0.4).to(DEV)   # the norm's input
    gamma, beta_ = (torch.rand(Cin, generator=g) + 0.5).to(DEV), (0.3 * torch.randn(Cin, generator=g)).to(DEV)
    mean, invstd = ops.bn_finalize(ops.bn_stats(xbn), N * H * W, 1e-5, 0.1)
    rmask = torch.empty(xbn.numel() // 4, dtype=torch.uint8, device=DEV)
    y = ops.bn_apply(xbn, mean, invstd, gamma, beta_, True, relu_mask=rmask)            # the conv's input
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    wg = cl_weight(w)
    Ho, Wo = ops.conv_out_size(H, k, stride, pad, dil), ops.conv_out_size(W, k, stride, pad, dil)
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).to(DEV)
    prev = torch.randn(N, H, W, Cin, generator=g).to(DEV)                                # an earlier writer of the same gradient
    aw, ag = ops.absmax(ops.weight_rsck(wg)), ops.absmax(dy)
    wpt = ops.weight_pairs_t(wg, aw)
    dyp, _ = to_planes(dy, loose(ag, 1.7))
    msg = []
    for planes in (False, True):
        src, bound = (dyp, loose(ag, 1.7)) if planes else (dy, ag)
        for beta in (0, 1):
            ref = prev.clone()
            ops.conv2d_dgrad(src, wg, y.shape, stride, pad, dil, out=ref, beta=beta, amax=(bound, aw), w_pairs_t=wpt, dy_planes=planes)
            s_ref = ops.bn_bwd_reduce(ref, xbn, y, mean, invstd, True, relu_mask=rmask)
            out, rng = prev.clone(), ops.new_amax(DEV)
            got, s2 = ops.conv2d_dgrad(src, wg, y.shape, stride, pad, dil, out=out, beta=beta, amax=(bound, aw), w_pairs_t=wpt,
                                       dy_planes=planes, amax_y=rng, bn_bwd=(xbn, rmask, mean, invstd))
            assert s2 is not None, "this shape must take the fused epilogue"
            assert torch.equal(got, ref), "dx differs from the plain data gradient"
            assert int(rng) == int(ops.absmax(ref))
            scale = s_ref.abs().reshape(2, Cin).max(dim=1).values.repeat_interleave(Cin)
            e = float(((s2 - s_ref).abs() / scale).max())
            msg.append(f"{'planes' if planes else 'split'} beta {beta}: {e:.1e}")
            assert e < 2e-6, e
    report(f"data gradient + batch-norm backward sums {case}: max |sums - reduce pass| / max |sums| " + ", ".join(msg))


def test_data_gradient_bnsums_refuses_what_it_cannot_do():
    """a shape without the fused epilogue (Cin not a whole column tile) returns (dx, None): nothing half-done"""
    g = torch.Generator().manual_seed(3)
    N, H, W, Cin, Cout = 1, 12, 20, 96, 64
    xbn = torch.randn(N, H, W, Cin, generator=g).to(DEV)
    mean, invstd = ops.bn_finalize(ops.bn_stats(xbn), N * H * W, 1e-5, 0.1)
    rmask = torch.empty(xbn.numel() // 4, dtype=torch.uint8, device=DEV)
    y = ops.bn_apply(xbn, mean, invstd, torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV), True, relu_mask=rmask)
    wg = cl_weight(torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5)
    dy = torch.randn(N, H, W, Cout, generator=g).to(DEV)
    aw, ag = ops.absmax(ops.weight_rsck(wg)), ops.absmax(dy)
    wpt = ops.weight_pairs_t(wg, aw)
    ref = ops.conv2d_dgrad(dy, wg, y.shape, amax=(ag, aw), w_pairs_t=wpt)
    got, s2 = ops.conv2d_dgrad(dy, wg, y.shape, amax=(ag, aw), w_pairs_t=wpt, bn_bwd=(xbn, rmask, mean, invstd))
    assert s2 is None and torch.equal(got, ref)


def test_training_step_with_and_without_fused_bn_backward_sums(report):
    """the same step with the batch-norm backward sums taken from the data gradients' epilogues (default) and by the reduction
    pass (SCHED.fuse_bn_bwd off): identical forward, gradients at summation-order level; how many norms still run the pass"""
    import copy
    import types
    from rcf_amd import config, layers, synth
    H, W, B = 96, 160, 2
    res = {}
    saved = layers.SCHED.fuse_bn_bwd
    try:
        for on in (True, False):
            layers.SCHED.fuse_bn_bwd = on
            kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, norm="BN")
            kw.update(log_interval=10 ** 9, train_iter=1)
            args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_test", object_channel=None, eval_save=False, eval_export=False)
            m = rcf_amd.RCFModel(args, **copy.deepcopy(kw))
            shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
            m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
            m.to(DEV).train()
            nb = synth.make_batch(B, H, W, config_id=1)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
            batch = {"imgs": [t(a) for a in nb["imgs"]], "gt_fw_flows": [t(a) for a in nb["gt_fw_flows"]],
                     "gt_bw_flows": [t(a) for a in nb["gt_bw_flows"]], "seq_ids": nb["seq_ids"], "seq_names": nb["seq_names"],
                     "paths": nb["paths"]}
            calls = {"n": 0}
            orig, orig2 = ops.bn_bwd_reduce, ops.bn_bwd_reduce2

            def counting(*a, **k):
                calls["n"] += 1
                return orig(*a, **k)

            def counting2(*a, **k):                  # a stage's first join + its downsample norm: one pass serves two norms
                calls["n"] += 2
                return orig2(*a, **k)
            ops.bn_bwd_reduce, ops.bn_bwd_reduce2 = counting, counting2
            try:
                losses = m(batch)
                losses["loss"].backward()
            finally:
                ops.bn_bwd_reduce, ops.bn_bwd_reduce2 = orig, orig2
            grads = {n: p.grad.detach().double().clone() for n, p in m.named_parameters() if p.grad is not None}
            res[on] = ({k: float(v) for k, v in losses.items()}, grads, calls["n"])
    finally:
        layers.SCHED.fuse_bn_bwd = saved
    assert res[True][0] == res[False][0], "the forward pass does not depend on the switch"
    worst = max((float((res[True][1][n] - g).norm() / g.norm()), n) for n, g in res[False][1].items() if float(g.norm()) > 0)
    report(f"training step, batch-norm backward sums from the data gradients' epilogues: {res[False][2]} -> {res[True][2]} reduction passes; "
           f"worst parameter-gradient difference {worst[0]:.1e} ({worst[1]})")
    assert res[False][2] >= 55 and res[True][2] <= 20       # (norms served; the joins of the stages' first blocks keep their shared pass)
    assert worst[0] < 1e-4


@pytest.mark.parametrize("case", [
    # N, Cin, Cout, k, H, W    (conv1 of an identity bottleneck: 1x1, Cin = the join's channels)
    (2, 1024, 256, 1, 33, 41),
    (2, 256, 64, 1, 30, 53),
    (1, 128, 128, 3, 20, 31),
    (2, 64, 64, 1, 16, 24),
])
def test_data_gradient_with_masked_addend(case, report):
    """rcf_conv2d_dgrad_add_f32: dx = data gradient + (mask ? add : 0) in one launch -- BIT-identical to writing the masked tensor
    (rcf_relu_mask_copy_mp, what rcf_bn_bwd_apply_mp's dres holds) and accumulating the data gradient onto it; register-split and
    pair-plane operand; with the batch-norm backward sums riding along."""
    N, Cin, Cout, k, H, W = case
    g = torch.Generator().manual_seed(41 + sum((i + 1) * v for i, v in enumerate(case)))
    pad = k // 2
    xbn = torch.randn(N, H, W, Cin, generator=g).to(DEV)
    mean, invstd = ops.bn_finalize(ops.bn_stats(xbn), N * H * W, 1e-5, 0.1)
    rm_in = torch.empty(xbn.numel() // 4, dtype=torch.uint8, device=DEV)
    ops.bn_apply(xbn, mean, invstd, torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV), True, relu_mask=rm_in)
    wg = cl_weight(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    dy = torch.randn(N, H, W, Cout, generator=g).to(DEV)
    add = torch.randn(N, H, W, Cin, generator=g).to(DEV)                                   # the join's output gradient
    jmask = torch.randint(0, 16, (N * H * W * Cin // 4,), generator=g, dtype=torch.uint8).to(DEV)   # the join's sign bits
    aw, ag = ops.absmax(ops.weight_rsck(wg)), ops.absmax(dy)
    wpt = ops.weight_pairs_t(wg, aw)
    dyp, _ = to_planes(dy, loose(ag, 2.3))
    bits = torch.stack([(jmask >> e) & 1 for e in range(4)], dim=1).reshape(N, H, W, Cin).bool()
    masked = ops.relu_mask_copy(add, jmask)
    assert torch.equal(masked, torch.where(bits, add, torch.zeros_like(add)))
    acc = masked.clone()
    ops.relu_mask_copy(add, jmask, out=acc, beta=1)
    assert torch.equal(acc, masked + masked)
    for planes in (False, True):
        src, bound = (dyp, loose(ag, 2.3)) if planes else (dy, ag)
        assert ops.dgrad_takes_addend(wg, xbn.shape, 1, pad, 1, Cout, (bound, aw), wpt, planes)
        ref = masked.clone()
        ops.conv2d_dgrad(src, wg, xbn.shape, 1, pad, 1, out=ref, beta=1, amax=(bound, aw), w_pairs_t=wpt, dy_planes=planes)
        rng = ops.new_amax(DEV)
        got = ops.conv2d_dgrad(src, wg, xbn.shape, 1, pad, 1, amax=(bound, aw), w_pairs_t=wpt, dy_planes=planes, amax_y=rng, addend=(add, jmask))
        assert torch.equal(got, ref) and int(rng) == int(ops.absmax(ref))
        s_ref = ops.bn_bwd_reduce(ref, xbn, None, mean, invstd, True, relu_mask=rm_in)
        got2, s2 = ops.conv2d_dgrad(src, wg, xbn.shape, 1, pad, 1, amax=(bound, aw), w_pairs_t=wpt, dy_planes=planes, addend=(add, jmask),
                                    bn_bwd=(xbn, rm_in, mean, invstd))
        assert torch.equal(got2, ref) and s2 is not None
        scale = s_ref.abs().reshape(2, Cin).max(dim=1).values.repeat_interleave(Cin)
        assert float(((s2 - s_ref).abs() / scale).max()) < 2e-6
    report(f"data gradient + masked addend {case}: bit-identical to mask copy + accumulate (register split and pair planes), sums ride along")


def test_training_step_with_and_without_deferred_residual_gradient(report):
    """the same step with the identity branches' gradients added in conv1's data-gradient epilogue (default) and written by the
    join's batch-norm backward (SCHED.defer_residual off): every parameter gradient BIT-identical; how many joins defer"""
    import copy
    import types
    from rcf_amd import config, layers, synth
    H, W, B = 96, 160, 2
    res = {}
    saved = layers.SCHED.defer_residual
    try:
        for on in (True, False):
            layers.SCHED.defer_residual = on
            kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, norm="BN")
            kw.update(log_interval=10 ** 9, train_iter=1)
            args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_test", object_channel=None, eval_save=False, eval_export=False)
            m = rcf_amd.RCFModel(args, **copy.deepcopy(kw))
            shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
            m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
            m.to(DEV).train()
            nb = synth.make_batch(B, H, W, config_id=1)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
            batch = {"imgs": [t(a) for a in nb["imgs"]], "gt_fw_flows": [t(a) for a in nb["gt_fw_flows"]],
                     "gt_bw_flows": [t(a) for a in nb["gt_bw_flows"]], "seq_ids": nb["seq_ids"], "seq_names": nb["seq_names"],
                     "paths": nb["paths"]}
            calls = {"add": 0, "copy": 0}
            o_d, o_c = ops.conv2d_dgrad, ops.relu_mask_copy

            def dgrad(*a, **k):
                calls["add"] += k.get("addend") is not None
                return o_d(*a, **k)

            def mcopy(*a, **k):
                calls["copy"] += 1
                return o_c(*a, **k)
            ops.conv2d_dgrad, ops.relu_mask_copy = dgrad, mcopy
            try:
                losses = m(batch)
                losses["loss"].backward()
            finally:
                ops.conv2d_dgrad, ops.relu_mask_copy = o_d, o_c
            res[on] = ({k: float(v) for k, v in losses.items()}, {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None},
                       dict(calls))
    finally:
        layers.SCHED.defer_residual = saved
    assert res[True][0] == res[False][0]
    diff = [n for n, g in res[False][1].items() if not torch.equal(g, res[True][1][n])]
    report(f"training step, deferred residual gradients: {res[True][2]['add']} of 16 joins add in conv1's epilogue "
           f"({res[True][2]['copy']} materialised), {len(diff)} parameter gradients differ")
    assert res[True][2]["add"] == 12 and res[True][2]["copy"] == 0 and res[False][2]["add"] == 0
    assert not diff, diff[:5]


def test_deferred_residual_gradient_is_materialised_for_other_readers():
    """Act.pending_add (a join's output gradient + sign bits left for conv1's data gradient) reaches every OTHER kind of reader as
    the tensor the join would have written: take_grad, take_grad_range and a producer that does not take addends"""
    from rcf_amd.layers import Act
    g = torch.Generator().manual_seed(9)
    N, H, W, C = 2, 9, 13, 64
    t = torch.randn(N, H, W, C, generator=g).to(DEV)
    dy = torch.randn(N, H, W, C, generator=g).to(DEV)
    mask = torch.randint(0, 16, (N * H * W * C // 4,), generator=g, dtype=torch.uint8).to(DEV)
    bits = torch.stack([(mask >> e) & 1 for e in range(4)], dim=1).reshape(N, H, W, C).bool()
    want = torch.where(bits, dy, torch.zeros_like(dy))
    a = Act(t)
    a.pending_add = (dy, mask)
    assert torch.equal(a.take_grad(), want) and a.pending_add is None
    a = Act(t)
    a.pending_add = (dy, mask)
    assert int(a.take_grad_range()) == int(ops.absmax(want)) and torch.equal(a.grad, want)
    a = Act(t)
    a.pending_add = (dy, mask)
    buf, beta = a.grad_slot()                       # an ordinary producer: gets the materialised tensor to accumulate onto
    assert beta == 1 and torch.equal(buf, want) and a.pending_add is None
    a = Act(t)
    a.pending_add = (dy, mask)
    buf, beta = a.grad_slot(takes_addend=True)      # conv1's data gradient: first writer, takes the pair itself
    pend = a.take_pending()
    assert beta == 0 and pend[0] is dy and pend[1] is mask and a.pending_add is None
    a = Act(t)
    a.grad = want.clone()
    a.pending_add = (dy, mask)
    buf, beta = a.grad_slot(takes_addend=True)      # somebody wrote before: accumulate, the pending pair is folded in first
    assert beta == 1 and torch.equal(buf, want + want) and a.pending_add is None
