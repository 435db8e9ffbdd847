"""GPU parity of the individual HIP kernels against plain PyTorch fp64 on CPU (the float kernels'
reference) and against the golden vectors captured from the reference (warp family)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import rcf_amd  # noqa: F401  (package alias)
from rcf_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def to_nhwc(x):  # [N,C,H,W] cpu -> [N,H,W,C] gpu contiguous
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def from_nhwc(y):
    return y.permute(0, 3, 1, 2).cpu()


def cl_weight(w):  # OIHW cpu -> channels_last gpu
    return w.to(DEV).contiguous(memory_format=torch.channels_last)


CONV_CASES = [
    # N, Cin, Cout, k, stride, pad, dil, H, W, bias, act
    (2, 64, 256, 1, 1, 0, 1, 13, 17, False, 0),
    (2, 64, 64, 3, 1, 1, 1, 13, 17, False, 0),
    (1, 128, 128, 3, 1, 2, 2, 15, 22, False, 0),
    (2, 32, 48, 3, 1, 6, 6, 20, 27, False, 0),
    (2, 64, 128, 3, 2, 1, 1, 13, 18, False, 0),
    (2, 64, 128, 1, 2, 0, 1, 13, 18, False, 0),
    (2, 4, 64, 7, 2, 3, 1, 30, 41, False, 0),
    (2, 256, 4, 1, 1, 0, 1, 12, 14, True, 0),
    (2, 4, 64, 3, 1, 1, 1, 12, 14, True, 1),
    (1, 260, 136, 3, 1, 4, 4, 11, 23, False, 0),
    (3, 512, 256, 1, 1, 0, 1, 9, 31, False, 0),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, report):
    N, Cin, Cout, k, stride, pad, dil, H, W, has_bias, act = case
    g = torch.Generator().manual_seed(sum((i + 1) * v for i, v in enumerate(case)))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) if has_bias else None
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    conv = F.conv2d(xd, wd, b.double() if has_bias else None, stride=stride, padding=pad, dilation=dil)
    yref = F.leaky_relu(conv, 0.1) if act else conv
    dy = torch.randn(yref.shape, generator=g)
    yref.backward(dy.double())
    xg, wg = to_nhwc(x), cl_weight(w)
    y = ops.conv2d_fwd(xg, wg, b.to(DEV) if has_bias else None, stride, pad, dil, act=act, slope=0.1)
    e_f = relerr(from_nhwc(y), yref)
    # the backward kernels are linear maps of the gradient w.r.t. the conv output (pre-activation)
    gpre = dy.double() * torch.where(conv.detach() > 0, 1.0, 0.1) if act else dy.double()
    gg = to_nhwc(gpre.float())
    dx = ops.conv2d_dgrad(gg, wg, xg.shape, stride, pad, dil)
    e_d = relerr(from_nhwc(dx), xd.grad)
    dw = torch.zeros_like(wg)
    ops.conv2d_wgrad(xg, gg, wg, dw, stride, pad, dil, beta=1)
    e_w = relerr(dw.cpu(), wd.grad)
    dx2 = ops.conv2d_dgrad(gg, wg, xg.shape, stride, pad, dil, out=dx.clone(), beta=1)
    e_acc = relerr(from_nhwc(dx2), 2 * xd.grad)
    report(f"conv {case}: fwd {e_f:.2e} dgrad {e_d:.2e} wgrad {e_w:.2e} acc {e_acc:.2e}")
    assert e_f < 2e-5 and e_d < 2e-5 and e_w < 2e-5 and e_acc < 2e-5


@pytest.mark.parametrize("xs,ws,gs,heavy", [(1.0, 1.0, 1.0, False), (1e-6, 30.0, 1e-9, False), (3e4, 1e-3, 1e5, False),
                                            (1.0, 1.0, 1e-7, True)])
@pytest.mark.parametrize("case", [CONV_CASES[i] for i in (0, 1, 2, 4, 6, 9, 10)] + [(2, 256, 512, 3, 1, 1, 1, 24, 40, False, 0)])
def test_conv_fp16_pairs(case, xs, ws, gs, heavy, report):
    """the fp16-pair kernels (operand ranges given: x*2^k = h + m in fp16, 3 partial products) against float64, over
    operand magnitudes from 1e-9 to 1e5 and a heavy-tailed gradient; the yardstick is torch's own fp32 conv error"""
    N, Cin, Cout, k, stride, pad, dil, H, W, _, _ = case
    g = torch.Generator().manual_seed(sum((i + 1) * v for i, v in enumerate(case)) + int(heavy))
    x = torch.randn(N, Cin, H, W, generator=g) * xs
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5 * ws
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yref = F.conv2d(xd, wd, None, stride=stride, padding=pad, dilation=dil)
    dy = torch.randn(yref.shape, generator=g) * gs
    if heavy:                                   # a few entries 1e4 times larger than the rest set the range
        dy = dy * torch.where(torch.rand(dy.shape, generator=g) < 1e-3, 1e4, 1.0)
    yref.backward(dy.double())
    x32, w32 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y32 = F.conv2d(x32, w32, None, stride=stride, padding=pad, dilation=dil)
    y32.backward(dy)

    def rms(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float(((a - b) ** 2).mean().sqrt() / ((b ** 2).mean().sqrt() + 1e-300))

    def mxe(a, b):
        """largest element error in units of the result's rms value"""
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).abs().max() / ((b ** 2).mean().sqrt() + 1e-300))
    xg, wg, gg = to_nhwc(x), cl_weight(w), to_nhwc(dy)
    ax, aw, ag = ops.absmax(xg), ops.absmax(ops.weight_rsck(wg)), ops.absmax(gg)
    assert float(ax.view(torch.float32)) == float(x.abs().max()) and float(ag.view(torch.float32)) == float(dy.abs().max())
    y = ops.conv2d_fwd(xg, wg, None, stride, pad, dil, amax=(ax, aw))
    # the weights split once beforehand (two fp16 planes) instead of in every row tile: the same bits
    assert torch.equal(y, ops.conv2d_fwd(xg, wg, None, stride, pad, dil, amax=(ax, aw), w_pairs=ops.weight_pairs(wg, aw)))
    dx = ops.conv2d_dgrad(gg, wg, xg.shape, stride, pad, dil, amax=(ag, aw))
    dw = torch.zeros_like(wg)
    ops.conv2d_wgrad(xg, gg, wg, dw, stride, pad, dil, beta=1, amax=(ax, ag))
    e = (rms(from_nhwc(y), yref), rms(from_nhwc(dx), xd.grad), rms(dw.cpu(), wd.grad))
    r = (rms(y32, yref), rms(x32.grad, xd.grad), rms(w32.grad, wd.grad))
    y6 = ops.conv2d_fwd(xg, wg, None, stride, pad, dil)
    e6 = rms(from_nhwc(y6), yref)
    report(f"conv fp16 pairs {case[:9]} scales x{xs:g} w{ws:g} dy{gs:g} heavy={heavy}: rms error vs float64 "
           f"fwd {e[0]:.2e} dgrad {e[1]:.2e} wgrad {e[2]:.2e} | torch fp32 {r[0]:.2e} {r[1]:.2e} {r[2]:.2e} | "
           f"bf16 triples fwd {e6:.2e}")
    for ei, ri in zip(e, r):
        assert ei < max(4 * ri, 5e-7)
    # ... and the LARGEST element error (in units of the result's rms), against the same yardstick: a kernel that is right
    # on average and wrong on a few elements (a dropped partial product on one tile edge, an overflowing part) fails here
    em = (mxe(from_nhwc(y), yref), mxe(from_nhwc(dx), xd.grad), mxe(dw.cpu(), wd.grad))
    rm = (mxe(y32, yref), mxe(x32.grad, xd.grad), mxe(w32.grad, wd.grad))
    report(f"   max element error / rms(result): fwd {em[0]:.2e} dgrad {em[1]:.2e} wgrad {em[2]:.2e} | torch fp32 "
           f"{rm[0]:.2e} {rm[1]:.2e} {rm[2]:.2e}")
    for ei, ri in zip(em, rm):
        assert ei < max(4 * ri, 1e-5)      # measured 0.8e-6 ... 6.2e-6 (torch fp32: 1.2e-6 ... 1.5e-5); a dropped partial product is >= 5e-4


@pytest.mark.parametrize("case", [(2, 256, 4, 33, 41), (1, 256, 8, 120, 214), (2, 256, 16, 30, 27), (3, 64, 4, 17, 19), (2, 512, 16, 12, 11),
                                  (2, 128, 8, 21, 9)])
def test_thin_1x1_convs(case, report):
    """csrc/thin.hip: 1x1 convs with 4 / 8 / 16 output channels (the decode heads' classifiers, models/fcn_head.py conv_seg) as streaming
    fp32 passes -- forward (+ bias, + accumulate), data gradient (+ accumulate, range of the result) and weight gradient (+ accumulate)
    against float64 and against the GEMM kernels they replace (RCF_CONV_NO_THIN); exact products: at or below torch's fp32 error"""
    from rcf_amd import _lib
    N, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g)
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double()
    yref = F.conv2d(xd, wd, bd)
    dy = torch.randn(yref.shape, generator=g)
    yref.backward(dy.double())
    x32, w32 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y32 = F.conv2d(x32, w32, b)
    y32.backward(dy)

    def rms(a, b_):
        a, b_ = a.detach().double().cpu(), b_.detach().double().cpu()
        return float(((a - b_) ** 2).mean().sqrt() / ((b_ ** 2).mean().sqrt() + 1e-300))
    xg, wg, gg, bg = to_nhwc(x), cl_weight(w), to_nhwc(dy), b.to(DEV)
    ax, aw, ag = ops.absmax(xg), ops.absmax(ops.weight_rsck(wg)), ops.absmax(gg)
    out = {}
    for thin in (True, False):
        old = ops.set_conv_flags(0 if thin else _lib.CONV_NO_THIN)
        try:
            y = ops.conv2d_fwd(xg, wg, bg, 1, 0, 1, amax=(ax, aw))
            y2 = ops.conv2d_fwd(xg, wg, bg, 1, 0, 1, amax=(ax, aw), out=y.clone(), beta=1)
            ry = ops.new_amax(DEV)
            dx = ops.conv2d_dgrad(gg, wg, xg.shape, 1, 0, 1, amax=(ag, aw), amax_y=ry)
            dx2 = ops.conv2d_dgrad(gg, wg, xg.shape, 1, 0, 1, amax=(ag, aw), out=dx.clone(), beta=1)
            dw = torch.full_like(wg, 0.5)
            ops.conv2d_wgrad(xg, gg, wg, dw, 1, 0, 1, beta=1, amax=(ax, ag))
        finally:
            ops.set_conv_flags(old)
        out[thin] = (y, y2, dx, dx2, dw, ry)
    y, y2, dx, dx2, dw, ry = out[True]
    e = (rms(from_nhwc(y), yref), rms(from_nhwc(dx), xd.grad), rms(dw.cpu() - 0.5, wd.grad))
    r = (rms(y32, yref), rms(x32.grad, xd.grad), rms(w32.grad, wd.grad))
    eg = (rms(from_nhwc(out[False][0]), yref), rms(from_nhwc(out[False][2]), xd.grad), rms(out[False][4].cpu() - 0.5, wd.grad))
    report(f"thin 1x1 conv {case}: rms error vs float64 fwd {e[0]:.1e} dgrad {e[1]:.1e} wgrad {e[2]:.1e} | GEMM kernels {eg[0]:.1e} {eg[1]:.1e} "
           f"{eg[2]:.1e} | torch fp32 {r[0]:.1e} {r[1]:.1e} {r[2]:.1e}")
    for ei, ri in zip(e, r):
        assert ei < max(2 * ri, 2e-7)
    # accumulate forms: exactly twice the plain result less the bias once (y + (y) and dx + dx in fp32: one rounding)
    assert rms(y2 - y, y) < 1e-6 and rms(dx2, 2 * dx) < 1e-6
    assert float(ry.view(torch.float32)) == float(dx.abs().max())


def test_conv_fp16_pairs_range_edges(report):
    """operand ranges at the edges: an all-zero tensor (range 0 -> scale 1, exact zeros out), magnitudes spread
    log-uniformly over 30 binary orders (elements far below the tensor's maximum lose bits, but only at 2^-39 of the
    maximum: the error relative to the output stays at fp32 level), and a range that is an upper bound 2^10 too large"""
    g = torch.Generator().manual_seed(5)
    N, Cin, Cout, H, W = 2, 128, 256, 17, 23
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    wg = cl_weight(w)
    aw = ops.absmax(ops.weight_rsck(wg))
    z = torch.zeros(N, H, W, Cin, device=DEV)
    yz = ops.conv2d_fwd(z, wg, None, 1, 1, 1, amax=(ops.absmax(z), aw))
    assert float(yz.abs().max()) == 0.0
    mag = torch.exp2(-30 * torch.rand(N, Cin, H, W, generator=g))
    x = torch.randn(N, Cin, H, W, generator=g) * mag
    ref = F.conv2d(x.double(), w.double(), None, 1, 1, 1)
    xg = to_nhwc(x)
    y = ops.conv2d_fwd(xg, wg, None, 1, 1, 1, amax=(ops.absmax(xg), aw))
    e_spread = float(((from_nhwc(y).double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
    loose = (ops.absmax(xg).view(torch.float32) * 1024.0).view(torch.int32)          # a valid, needlessly large bound
    y2 = ops.conv2d_fwd(xg, wg, None, 1, 1, 1, amax=(loose, aw))
    e_loose = float(((from_nhwc(y2).double() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
    report(f"conv fp16 pairs range edges: zero tensor exact, 30 binary orders of magnitude rms {e_spread:.2e}, "
           f"range 2^10 too large rms {e_loose:.2e}")
    assert e_spread < 1e-6 and e_loose < 1e-6


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_conv_kernel_variants(variant, report):
    """the tuning variants of the implicit-GEMM kernel (K-step 16/32, K-major / row-major LDS) agree"""
    from rcf_amd import _lib
    try:
        ops.set_conv_flags(_lib.CONV_FP32_MFMA(variant))        # per-call flag (RCF_CONV_FP32_MFMA), OR-ed into every launch
        for case in (CONV_CASES[0], CONV_CASES[3], CONV_CASES[4], CONV_CASES[6], CONV_CASES[9], CONV_CASES[7]):
            N, Cin, Cout, k, stride, pad, dil, H, W, has_bias, act = case
            g = torch.Generator().manual_seed(variant * 100 + Cin)
            x = torch.randn(N, Cin, H, W, generator=g)
            w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
            xd, wd = x.double().requires_grad_(True), w.double()
            yref = F.conv2d(xd, wd, None, stride=stride, padding=pad, dilation=dil)
            dy = torch.randn(yref.shape, generator=g)
            yref.backward(dy.double())
            xg, wg = to_nhwc(x), cl_weight(w)
            e_f = relerr(from_nhwc(ops.conv2d_fwd(xg, wg, None, stride, pad, dil)), yref)
            e_d = relerr(from_nhwc(ops.conv2d_dgrad(to_nhwc(dy), wg, xg.shape, stride, pad, dil)), xd.grad)
            report(f"conv variant {variant} {case}: fwd {e_f:.2e} dgrad {e_d:.2e}")
            assert e_f < 2e-5 and e_d < 2e-5
    finally:
        ops.set_conv_flags(0)


@pytest.mark.parametrize("case", [
    # N, Cin, Cout, k, pad, dil, H, W, region: row counts that give every sub-tile height (4..8 row blocks, masked < 4)
    (2, 64, 256, 3, 2, 2, 60, 107, None),        # 12 840 rows = 402 blocks: ranges of 1 / 2 blocks (masked 4-block tiles)
    (8, 64, 256, 3, 1, 1, 64, 107, None),        # 54 784 rows = 1 712 blocks: ranges of 6 / 7
    (12, 32, 512, 3, 3, 3, 60, 107, None),       # 77 040 rows: ranges of 9 / 10 -> 5 + 4, 5 + 5; two column tiles; K = 288
    (16, 16, 256, 1, 0, 1, 60, 107, None),       # 102 720 rows: 12 / 13 -> 8 + 4, 8 + 5; K = 16: ONE K-step (ring padded to 4)
    (3, 48, 256, 3, 6, 6, 60, 107, (0, 0, 60, 107, 7)),      # border frame (the commuted decode-head conv's band)
    (3, 48, 256, 3, 6, 6, 60, 107, (5, 9, 40, 70)),          # rectangle
    # 1x1 convs (short K-loops: the built-in rule keeps them on the 128 x 256 kernel)
    (16, 256, 1024, 1, 0, 1, 60, 107, None),     # 12 / 13 row blocks per range: 3 + 1 sub-tiles (one of them 1 block), 4 column tiles, 16 K-steps
    (5, 192, 256, 1, 0, 1, 60, 107, None),       # K = 192: 12 K-steps (no plain steps); ranges of 3 / 4 blocks: a single tile per workgroup
    (2, 512, 512, 1, 0, 1, 33, 41, None),        # 2 706 rows = 85 blocks on 256 workgroups: most of them have NO tile; 32 K-steps
    (9, 320, 768, 1, 0, 1, 60, 107, None),       # 3 column tiles (workgroups cannot share ranges: gn = 1), K = 320: 20 K-steps
])
def test_conv_h2p_matches_x3(case, report):
    """the persistent LDS-DMA kernel (csrc/igemm_h2p.inc: one workgroup per CU, 4-stage ring, weights by DMA, balanced row
    ranges) against the 128 x 256 kernel it replaces on the deep 3x3 layers: forward (+ fused batch-norm statistics),
    data gradient (overwrite and accumulate), whole tensors and regions -- the same arithmetic in the same order, so the
    outputs must be BIT-identical; and against float64 at fp32 level"""
    N, Cin, Cout, k, pad, dil, H, W, reg = case
    g = torch.Generator().manual_seed(sum((i + 1) * v for i, v in enumerate(case[:8])))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    dy = torch.randn(N, Cout, H, W, generator=g)
    xg, wg, gg = to_nhwc(x), cl_weight(w), to_nhwc(dy)
    ax, aw, ag = ops.absmax(xg), ops.absmax(ops.weight_rsck(wg)), ops.absmax(gg)
    res = {}
    try:
        from rcf_amd import _lib
        for mode in (0, 1):
            ops.set_conv_flags(_lib.CONV_H2P_ALWAYS if mode else _lib.CONV_H2P_NEVER)
            wp, wpt = ops.weight_pairs(wg, aw), ops.weight_pairs_t(wg, aw)
            y = torch.full((N, H, W, Cout), 3.0, device=DEV)
            ops.conv2d_fwd(xg, wg, None, 1, pad, dil, out=y, amax=(ax, aw), w_pairs=wp, region=reg)
            dx = torch.full((N, H, W, Cin), 5.0, device=DEV)
            # dgrad: Ncol = Cin must be a multiple of 256 for the persistent kernel -> swap the roles: gradient of a conv
            # whose INPUT has Cout channels (weights transposed by viewing the same tensor as [Cin', Cout'])
            dxw = torch.full((N, H, W, Cout), 5.0, device=DEV)
            wT = cl_weight(w.permute(1, 0, 2, 3).contiguous())                     # [Cin, Cout, k, k]: a Cout -> Cin conv
            awT = ops.absmax(ops.weight_rsck(wT))
            wptT = ops.weight_pairs_t(wT, awT)
            ops.conv2d_dgrad(xg, wT, (N, H, W, Cout), 1, pad, dil, out=dxw, amax=(ax, awT), w_pairs_t=wptT, region=reg)
            acc = dxw.clone()
            ops.conv2d_dgrad(xg, wT, (N, H, W, Cout), 1, pad, dil, out=acc, beta=1, amax=(ax, awT), w_pairs_t=wptT, region=reg)
            st = None
            if reg is None:
                ys, st = ops.conv2d_fwd_stats(xg, wg, 1, pad, dil, amax=(ax, aw), w_pairs=wp)
                assert torch.equal(ys, y)
            res[mode] = (y, dxw, acc, st)
    finally:
        ops.set_conv_flags(0)
    a, b = res[0], res[1]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    if reg is None:
        e_s = float((a[3] - b[3]).abs().max() / a[3].abs().max())
        assert e_s < 1e-7, e_s                     # the partial sums are grouped differently (row ranges, not 128-row tiles)
        yref = F.conv2d(x.double(), w.double(), None, 1, pad, dil)
        e_f = relerr(from_nhwc(b[0]), yref)
        dref = F.conv_transpose2d(x.double(), w.permute(1, 0, 2, 3).double(), None, 1, pad, 0, 1, dil)
        e_d = relerr(from_nhwc(b[1]), dref)
        sref = torch.cat([yref.sum((0, 2, 3)), (yref ** 2).sum((0, 2, 3))])
        e_st = float((b[3].cpu() - sref).abs().max() / sref.abs().max())
        report(f"conv h2p {case[:8]}: identical to the 128x256 kernel; vs float64 fwd {e_f:.2e} dgrad {e_d:.2e} stats {e_st:.2e}")
        assert e_f < 2e-5 and e_d < 2e-5 and e_st < 1e-6
    else:
        report(f"conv h2p region {case}: identical to the 128x256 kernel (fwd, dgrad, accumulate)")


@pytest.mark.parametrize("case", [
    # N, Cin, Cout, k, stride, pad, dil, H, W, region (y0, x0, h, w) in output coords (fwd/wgrad) / input coords (dgrad)
    (2, 32, 48, 3, 1, 6, 6, 20, 27, (7, 7, 6, 13)),
    (2, 32, 48, 3, 1, 6, 6, 20, 27, (0, 0, 7, 27)),          # full-width strip: rows wrap inside the rectangle
    (3, 64, 136, 3, 1, 3, 3, 11, 23, (2, 20, 9, 3)),         # 3-pixel-wide strip (narrower than a K-step)
    (2, 256, 256, 1, 1, 0, 1, 9, 14, (1, 2, 5, 9)),
    (2, 32, 48, 3, 1, 6, 6, 20, 27, (0, 0, 20, 27, 7)),      # border frame of thickness 7 of the whole image
    (3, 64, 136, 3, 1, 3, 3, 11, 23, (1, 2, 9, 20, 2)),      # frame of a sub-rectangle
])
def test_conv_region(case, report):
    """the rectangle-restricted forward / data-gradient / weight-gradient (rcf_conv2d_*_region_f32) against
    float64 torch on the same rectangle; pixels outside the rectangle must stay untouched"""
    N, Cin, Cout, k, stride, pad, dil, H, W, reg = case
    y0, x0, rh, rw = reg[:4]
    t = reg[4] if len(reg) > 4 else 0

    def rect_mask(shape):
        m = torch.zeros(shape)
        m[:, :, y0:y0 + rh, x0:x0 + rw] = 1
        if t:
            m[:, :, y0 + t:y0 + rh - t, x0 + t:x0 + rw - t] = 0
        return m
    g = torch.Generator().manual_seed(sum((i + 1) * v for i, v in enumerate(case[:9])))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yref = torch.nn.functional.conv2d(xd, wd, None, stride, pad, dil)
    dy = torch.randn(yref.shape, generator=g)
    mask = rect_mask(dy.shape)
    yref.backward((dy * mask).double())                          # only the rectangle's output pixels contribute
    xg, wg, gg = to_nhwc(x), cl_weight(w), to_nhwc(dy)
    out = torch.full((N, yref.shape[2], yref.shape[3], Cout), 7.0, device=DEV)
    ops.conv2d_fwd(xg, wg, None, stride, pad, dil, out=out, region=reg)
    o = from_nhwc(out)
    e_f = relerr(o * mask, yref.detach() * mask)
    outside = float(((o - 7.0).abs() * (1 - mask)).max())
    dw = torch.zeros_like(wg)
    ops.conv2d_wgrad(xg, gg, wg, dw, stride, pad, dil, beta=1, region=reg)
    e_w = relerr(dw.cpu(), wd.grad)
    # dgrad: rectangle in INPUT coordinates, full dy
    xd2 = x.double().requires_grad_(True)
    torch.nn.functional.conv2d(xd2, w.double(), None, stride, pad, dil).backward(dy.double())
    dx = torch.full((N, H, W, Cin), 7.0, device=DEV)
    ops.conv2d_dgrad(gg, wg, (N, H, W, Cin), stride, pad, dil, out=dx, region=reg)
    d = from_nhwc(dx)
    imask = rect_mask((N, Cin, H, W))
    e_d = relerr(d * imask, xd2.grad * imask)
    outside_d = float(((d - 7.0).abs() * (1 - imask)).max())
    report(f"conv region {case}: fwd {e_f:.2e} wgrad {e_w:.2e} dgrad {e_d:.2e} outside {outside} {outside_d}")
    assert e_f < 2e-5 and e_w < 2e-5 and e_d < 2e-5 and outside == 0.0 and outside_d == 0.0


@pytest.mark.parametrize("case", [
    # N, Cin, Cout, k, stride, pad, dil, H, W, region in output coordinates or None
    (2, 256, 256, 1, 1, 0, 1, 9, 14, (1, 2, 5, 9)),          # region, one tap per 256-column tile
    (2, 256, 136, 3, 1, 3, 3, 19, 23, (0, 0, 19, 23, 5)),    # frame of the whole image (the commuted decode-head conv's shape)
    (3, 64, 136, 3, 1, 3, 3, 11, 23, (1, 2, 9, 20, 2)),      # region, a 256-column tile spans four taps
    (3, 64, 64, 3, 1, 1, 1, 11, 23, (2, 20, 9, 3)),          # 3-pixel-wide strip, 64 output channels (half-empty tile rows)
    (2, 128, 200, 1, 1, 0, 1, 17, 19, None),                 # 128 columns: the 128 x 128 tile, one tap
    (2, 192, 136, 1, 1, 0, 1, 17, 19, None),                 # 192 columns: the 128 x 128 tile, ragged second column tile
    (2, 128, 72, 3, 2, 1, 1, 21, 37, None),                  # stride 2, two taps per 256-column tile, ragged last tile (1152 columns)
    (1, 320, 136, 3, 1, 2, 2, 13, 16, None),                 # Cin = 5 x 64: column tiles straddle taps at a 64-channel boundary
    (2, 64, 256, 1, 1, 0, 1, 17, 19, None),                  # 64 columns: half of a 128-column tile
    (2, 256, 64, 1, 1, 0, 1, 17, 19, None),                  # 64 output channels: half of the tile's rows
    (2, 64, 72, 1, 2, 0, 1, 21, 23, None),                   # both, stride 2
])
def test_conv_wgrad_fp16_pairs_columns(case, report):
    """the fp16-pair weight gradient with (tap, channel) pairs as GEMM columns (igemm_wgrad_h2t_kernel: whole tensors and
    regions, one tap or several per column tile, both tile widths) against float64; yardstick: torch's own fp32 conv"""
    N, Cin, Cout, k, stride, pad, dil, H, W, reg = case
    g = torch.Generator().manual_seed(sum((i + 1) * v for i, v in enumerate(case[:9])))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
    xd, wd = x.double(), w.double().requires_grad_(True)
    yref = F.conv2d(xd, wd, None, stride, pad, dil)
    dy = torch.randn(yref.shape, generator=g)
    mask = torch.ones(dy.shape)
    if reg is not None:
        y0, x0, rh, rw = reg[:4]
        t = reg[4] if len(reg) > 4 else 0
        mask = torch.zeros(dy.shape)
        mask[:, :, y0:y0 + rh, x0:x0 + rw] = 1
        if t:
            mask[:, :, y0 + t:y0 + rh - t, x0 + t:x0 + rw - t] = 0
    yref.backward((dy * mask).double())
    w32 = w.clone().requires_grad_(True)
    F.conv2d(x, w32, None, stride, pad, dil).backward(dy * mask)

    def rms(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float(((a - b) ** 2).mean().sqrt() / ((b ** 2).mean().sqrt() + 1e-300))
    xg, wg, gg = to_nhwc(x), cl_weight(w), to_nhwc(dy)
    ax, ag = ops.absmax(xg), ops.absmax(gg)
    dw = torch.full_like(wg, 3.0)
    ops.conv2d_wgrad(xg, gg, wg, dw, stride, pad, dil, beta=0, region=reg, amax=(ax, ag))
    e0 = rms(dw.cpu(), wd.grad)
    ops.conv2d_wgrad(xg, gg, wg, dw, stride, pad, dil, beta=1, region=reg, amax=(ax, ag))      # accumulate: 2 x
    e1 = rms(dw.cpu(), 2 * wd.grad)
    r = rms(w32.grad, wd.grad)
    report(f"conv wgrad fp16 pairs, columns {case}: rms error vs float64 {e0:.2e} (accumulated {e1:.2e}) | torch fp32 {r:.2e}")
    assert e0 < max(4 * r, 5e-7) and e1 < max(4 * r, 5e-7)


@pytest.mark.parametrize("case", [
    # N, Cin, Cout, k, pad, dil, H, W, region
    (4, 256, 256, 3, 2, 2, 60, 107, None),                 # 3x3: 9 column tiles of one tap each
    (3, 512, 256, 1, 0, 1, 60, 107, None),                 # 1x1, two column tiles
    (2, 256, 512, 3, 3, 3, 33, 41, (0, 0, 33, 41, 6)),     # frame region, two row tiles
])
def test_conv_wgrad_256x256_tile_matches_128x256(case, report):
    """igemm_wgrad_h2t_kernel<.., MR = 4> (256 x 256 tile, one workgroup per CU: the default where Cout and Cin are multiples
    of 256) against the 128 x 256 tile (RCF_CONV_WGRAD_TILE_128, what the overlapped launches of the training step take): every element is the same sum over the same pixel chunks in
    the same order when the split counts agree, and fp32-level agreement otherwise; both against float64"""
    N, Cin, Cout, k, pad, dil, H, W, reg = case
    g = torch.Generator().manual_seed(sum((i + 1) * v for i, v in enumerate(case[:8])))
    x = torch.randn(N, H, W, Cin, generator=g).to(DEV)
    dy = torch.randn(N, H, W, Cout, generator=g).to(DEV)
    wg = cl_weight(torch.randn(Cout, Cin, k, k, generator=g))
    ax, ag = ops.absmax(x), ops.absmax(dy)
    res = {}
    try:
        for mode in (0, 1):
            dw = torch.full_like(wg, 3.0)
            ops.conv2d_wgrad(x, dy, wg, dw, 1, pad, dil, beta=0, region=reg, amax=(ax, ag), small_tile=not mode)
            acc = dw.clone()
            ops.conv2d_wgrad(x, dy, wg, acc, 1, pad, dil, beta=1, region=reg, amax=(ax, ag), small_tile=not mode)
            res[mode] = (dw, acc)
    finally:
        pass
    d = [float((a - b).abs().max() / a.abs().max()) for a, b in zip(res[0], res[1])]
    ref = torch.nn.grad.conv2d_weight(from_nhwc(x).double(), tuple(wg.shape), (from_nhwc(dy) * _region_mask(dy, reg).cpu()).double(),
                                      1, pad, dil)
    e = relerr(res[1][0], ref)
    report(f"conv wgrad 256x256 tile {case}: max difference to the 128x256 tile {d[0]:.1e} (accumulated {d[1]:.1e}); vs float64 {e:.2e}")
    assert max(d) < 2e-6 and e < 2e-5


def test_conv_column_tile_xcd_mapping_is_a_permutation(report):
    """csrc/rcf_common.h rcf_conv_tile: convs whose weight operand exceeds L2 many times over (the data gradient of a 3x3 conv
    with 2048 input channels: 19 MB of fp16 pairs, 16 column tiles) give every XCD its own column tiles of all row tiles
    instead of a band of row tiles -- only which workgroup computes which tile changes: bit-identical, fp32 pairs and bf16,
    data gradient (the byte model takes the new mapping) and forward of the mirrored shape (2048 output channels)"""
    g = torch.Generator().manual_seed(5)
    N, Cin, Cout, H, W = 1, 2048, 256, 47, 107            # 5 029 rows: a ragged last row tile
    x = torch.randn(N, H, W, Cin, generator=g).to(DEV)
    dy = torch.randn(N, H, W, Cout, generator=g).to(DEV)
    w = cl_weight(torch.randn(Cout, Cin, 3, 3, generator=g) * 0.02)
    wT = cl_weight(torch.randn(Cin, Cout, 3, 3, generator=g) * 0.02)       # a 256 -> 2048 conv: wide FORWARD
    ag, aw, ax, awT = ops.absmax(dy), ops.absmax(ops.weight_rsck(w)), ops.absmax(x), ops.absmax(ops.weight_rsck(wT))
    dyb = dy.bfloat16()
    res = {}
    try:
        from rcf_amd import _lib
        for mode in (0, 1):
            ops.set_conv_flags(0 if mode else _lib.CONV_NO_COLMAP)
            dx = ops.conv2d_dgrad(dy, w, x.shape, 1, 3, 3, amax=(ag, aw), w_pairs_t=ops.weight_pairs_t(w, aw))
            acc = dx.clone()
            ops.conv2d_dgrad(dy, w, x.shape, 1, 3, 3, out=acc, beta=1, amax=(ag, aw), w_pairs_t=ops.weight_pairs_t(w, aw))
            y = ops.conv2d_fwd(dy, wT, None, 1, 3, 3, amax=(ag, awT), w_pairs=ops.weight_pairs(wT, awT))
            dxb = ops.conv2d_dgrad_bf16(dyb, w, x.shape, 1, 3, 3, w_t_bf16=ops.weight_bf16(w, transpose=True))
            res[mode] = (dx, acc, y, dxb)
    finally:
        ops.set_conv_flags(0)
    same = [torch.equal(a, b) for a, b in zip(res[0], res[1])]
    ref = F.conv_transpose2d(from_nhwc(dy).double(), w.cpu().double(), None, 1, 3, 0, 1, 3)
    e = relerr(from_nhwc(res[1][0]), ref)
    report(f"conv column-tile XCD mapping: identical dgrad / accumulate / forward / bf16 dgrad {same}; dgrad vs float64 {e:.2e}")
    assert all(same) and e < 2e-5


@pytest.mark.parametrize("case", [
    # N, Cin, Cout, k, pad, dil, H, W, region
    (4, 256, 256, 3, 2, 2, 60, 107, None),                 # 18 tiles x many splits: total % 8 != 0
    (2, 512, 136, 1, 0, 1, 33, 41, None),                  # 2 x 2 tiles, ragged rows
    (16, 64, 256, 1, 0, 1, 60, 107, None),                 # ONE column tile of 128: 2 tiles, splits >> tiles
    (3, 256, 136, 3, 3, 3, 19, 23, (0, 0, 19, 23, 5)),     # frame region
    (1, 64, 64, 3, 1, 1, 9, 11, None),                     # fewer than 16 workgroups: plain order
])
def test_conv_wgrad_xcd_mapping_is_a_permutation(case, report):
    """rcf_wgrad_item (csrc/rcf_common.h) only re-orders which workgroup computes which (output tile, pixel range): the
    weight gradient must be BIT-identical with the mapping on and off, fp16 pairs and bf16, overwrite and accumulate"""
    N, Cin, Cout, k, pad, dil, H, W, reg = case
    g = torch.Generator().manual_seed(sum((i + 1) * v for i, v in enumerate(case[:8])))
    x = torch.randn(N, H, W, Cin, generator=g).to(DEV)
    dy = torch.randn(N, H, W, Cout, generator=g).to(DEV)
    wg = cl_weight(torch.randn(Cout, Cin, k, k, generator=g))
    ax, ag = ops.absmax(x), ops.absmax(dy)
    xb, dyb = x.bfloat16(), dy.bfloat16()
    res = {}
    try:
        from rcf_amd import _lib
        for mode in (0, 1):
            ops.set_conv_flags(0 if mode else _lib.CONV_NO_WGRAD_XCD)
            d32 = torch.full_like(wg, 3.0)
            ops.conv2d_wgrad(x, dy, wg, d32, 1, pad, dil, beta=0, region=reg, amax=(ax, ag))
            a32 = d32.clone()
            ops.conv2d_wgrad(x, dy, wg, a32, 1, pad, dil, beta=1, region=reg, amax=(ax, ag))
            d16 = torch.full_like(wg, 3.0)
            ops.conv2d_wgrad_bf16(xb, dyb, wg, d16, 1, pad, dil, beta=0, region=reg)
            res[mode] = (d32, a32, d16)
    finally:
        ops.set_conv_flags(0)
    same = [torch.equal(a, b) for a, b in zip(res[0], res[1])]
    ref = torch.nn.grad.conv2d_weight(from_nhwc(x).double(), tuple(wg.shape),
                                      (from_nhwc(dy) * _region_mask(dy, reg).cpu()).double(), 1, pad, dil)
    e = relerr(res[1][0], ref)
    report(f"conv wgrad XCD mapping {case}: identical fp16-pairs / accumulate / bf16 {same}; vs float64 {e:.2e}")
    assert all(same) and e < 2e-5


def _region_mask(dy_nhwc, reg):
    """[N,C,H,W] 0/1 mask of the contributing output pixels"""
    N, H, W, C = dy_nhwc.shape
    m = torch.ones(N, C, H, W, device=dy_nhwc.device)
    if reg is not None:
        y0, x0, rh, rw = reg[:4]
        t = reg[4] if len(reg) > 4 else 0
        m = torch.zeros(N, C, H, W, device=dy_nhwc.device)
        m[:, :, y0:y0 + rh, x0:x0 + rw] = 1
        if t:
            m[:, :, y0 + t:y0 + rh - t, x0 + t:x0 + rw - t] = 0
    return m


@pytest.mark.parametrize("N,Cin,Cout,k,stride,pad,dil,H,W", [
    (2, 64, 64, 1, 1, 0, 1, 31, 45),           # 64-wide tiles, ragged last row tile
    (2, 4, 64, 7, 2, 3, 1, 60, 107),           # stem
    (3, 64, 256, 1, 1, 0, 1, 23, 29),          # 128 x 256 tiles
    (2, 128, 136, 3, 1, 2, 2, 20, 27),         # ragged column tile
    (9, 32, 32, 3, 2, 1, 1, 40, 40),           # several row tiles per image and images per tile
])
def test_conv_fwd_fused_bn_stats(N, Cin, Cout, k, stride, pad, dil, H, W, report):
    """rcf_conv2d_fwd_stats_f32: the output is bit-identical to the plain forward and the epilogue's per-channel
    sums equal the float64 column sums of that output to fp32 rounding of a lane-level partial sum (as the separate pass)"""
    g = torch.Generator().manual_seed(N * 1000 + Cin + Cout + k)
    x = torch.randn(N, Cin, H, W, generator=g) + 0.3
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
    xg, wg = to_nhwc(x), cl_weight(w)
    y0 = ops.conv2d_fwd(xg, wg, None, stride, pad, dil)
    y1, sums = ops.conv2d_fwd_stats(xg, wg, stride, pad, dil)
    assert torch.equal(y0, y1)
    ref = ops.bn_stats(y0)
    yd = y0.double().reshape(-1, Cout)
    truth = torch.cat([yd.sum(0), (yd * yd).sum(0)])
    scale = torch.cat([yd.abs().sum(0), (yd * yd).sum(0)]).clamp_min(1e-30)
    e_sep = float(((ref - truth).abs() / scale).max())
    e_fused = float(((sums - truth).abs() / scale).max())
    report(f"conv+bn-stats N{N} {Cin}->{Cout} k{k}: fused sums rel err {e_fused:.2e} (separate pass {e_sep:.2e})")
    assert e_fused < 5e-7 and e_sep < 5e-7


def test_conv_large_wgrad_splitk(report):
    """enough pixels for the split-K path (workspace + deterministic reduce)"""
    g = torch.Generator().manual_seed(77)
    N, Cin, Cout, H, W = 4, 64, 64, 60, 107
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.04
    dy = torch.randn(N, Cout, H, W, generator=g)
    wd = w.double().requires_grad_(True)
    F.conv2d(x.double(), wd, padding=1).backward(dy.double())
    xg, gg, wg = to_nhwc(x), to_nhwc(dy), cl_weight(w)
    dw1, dw2 = torch.zeros_like(wg), torch.zeros_like(wg)
    ops.conv2d_wgrad(xg, gg, wg, dw1, 1, 1, 1)
    ops.conv2d_wgrad(xg, gg, wg, dw2, 1, 1, 1)
    e = relerr(dw1.cpu(), wd.grad)
    report(f"wgrad split-K: {e:.2e} deterministic={torch.equal(dw1, dw2)}")
    assert e < 2e-5 and torch.equal(dw1, dw2)


def test_conv_pitched_slices(report):
    """input read from / output written into channel slices of wider NHWC buffers (concat layout)."""
    g = torch.Generator().manual_seed(5)
    N, H, W = 2, 9, 11
    x = torch.randn(N, 64, H, W, generator=g)
    w = torch.randn(128, 64, 3, 3, generator=g) * 0.05
    big_in = torch.zeros(N, H, W, 96, device=DEV)
    big_in[..., 16:80] = to_nhwc(x)
    big_out = torch.full((N, H, W, 200), 7.0, device=DEV)
    ops.conv2d_fwd(big_in[..., 16:80], cl_weight(w), None, 1, 1, 1, out=big_out[..., 8:136])
    ref = F.conv2d(x.double(), w.double(), padding=1)
    e = relerr(from_nhwc(big_out[..., 8:136]), ref)
    report(f"conv pitched: {e:.2e}")
    assert e < 2e-5
    assert float(big_out[..., :8].min()) == 7.0 and float(big_out[..., 136:].max()) == 7.0


@pytest.mark.parametrize("C,relu,res,drop", [(64, True, False, False), (256, True, True, False),
                                             (128, False, False, False), (256, True, False, True),
                                             (2048, True, True, False)])
def test_batchnorm_train(C, relu, res, drop, report):
    g = torch.Generator().manual_seed(C + relu)
    N, H, W = 3, 7, 9
    x = (torch.randn(N, C, H, W, generator=g) * 2 + 0.7)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    r = torch.randn(N, C, H, W, generator=g) if res else None
    keep = (torch.rand(N, C, generator=g) > 0.3).float() / 0.7 if drop else None
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rd = r.double().requires_grad_(True) if res else None
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    y = F.batch_norm(xd, rm, rv, gd, bd, training=True, momentum=0.1, eps=1e-5)
    if res:
        y = y + rd
    if relu:
        y = F.relu(y)
    if drop:
        y = y * keep.double()[:, :, None, None]
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())

    xg = to_nhwc(x)
    rmean, rvar = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    sums = ops.bn_stats(xg)
    count = N * H * W
    mean, invstd = ops.bn_finalize(sums, count, 1e-5, 0.1, rmean, rvar)
    gg, bg = gamma.to(DEV), beta.to(DEV)
    rg = to_nhwc(r) if res else None
    kg = keep.to(DEV) if drop else None
    rmask = torch.empty(xg.numel() // 4, dtype=torch.uint8, device=DEV) if relu else None
    yg = ops.bn_apply(xg, mean, invstd, gg, bg, relu, residual=rg, chan_scale=kg, relu_mask=rmask)
    e_y = relerr(from_nhwc(yg), y)
    e_rm, e_rv = relerr(rmean, rm), relerr(rvar, rv)
    dyg = to_nhwc(dy)
    s2 = ops.bn_bwd_reduce(dyg, xg, yg, mean, invstd, relu, chan_scale=kg)
    dgam, dbet = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dres = torch.empty_like(xg) if res else None
    dx = ops.bn_bwd_apply(dyg, xg, yg, mean, invstd, gg, relu, s2, count, dgam, dbet, dres=dres, chan_scale=kg)
    if relu:
        # the backward kernels fed the sign-bit mask instead of y: identical results
        s2m = ops.bn_bwd_reduce(dyg, xg, None, mean, invstd, relu, chan_scale=kg, relu_mask=rmask)
        dresm = torch.empty_like(xg) if res else None
        dxm = ops.bn_bwd_apply(dyg, xg, None, mean, invstd, gg, relu, s2m, count, torch.zeros(C, device=DEV),
                               torch.zeros(C, device=DEV), dres=dresm, chan_scale=kg, relu_mask=rmask)
        assert torch.equal(s2m, s2) and torch.equal(dxm, dx) and (not res or torch.equal(dresm, dres))
    e_dx, e_dg, e_db = relerr(from_nhwc(dx), xd.grad), relerr(dgam, gd.grad), relerr(dbet, bd.grad)
    e_dr = relerr(from_nhwc(dres), rd.grad) if res else 0.0
    report(f"bn C={C} relu={relu} res={res} drop={drop}: y {e_y:.2e} rm {e_rm:.2e} rv {e_rv:.2e} dx {e_dx:.2e} "
           f"dgamma {e_dg:.2e} dbeta {e_db:.2e} dres {e_dr:.2e}")
    assert max(e_y, e_rm, e_rv, e_dx, e_dg, e_db, e_dr) < 2e-5


@pytest.mark.parametrize("C,rows_shape,res,bf16", [(64, (3, 60, 107), False, False), (256, (2, 60, 107), True, False),
                                                  (1024, (1, 83, 107), True, False), (48, (2, 60, 107), False, False),
                                                  (512, (2, 60, 107), True, True)])
def test_batchnorm_cache_aware_row_order(C, rows_shape, res, bf16, report):
    """csrc/bn.hip struct Sweep: the banded row orders of the streaming batch-norm kernels visit every row exactly once -- the
    element-wise outputs (y, ReLU mask, range, dx, dres) are BIT-identical to the front-to-back walk, the backward sums
    agree to fp64 rounding; row counts that do not fill the eight bands, a channel count whose row group does not
    divide the band (C = 48: falls back to the plain order), fp32 and bf16 storage"""
    N, H, W = rows_shape
    g = torch.Generator().manual_seed(C + N)
    dt = torch.bfloat16 if bf16 else torch.float32
    x = (torch.randn(N, H, W, C, generator=g) * 2 + 0.7).to(DEV).to(dt)
    r = torch.randn(N, H, W, C, generator=g).to(DEV).to(dt) if res else None
    dy = torch.randn(N, H, W, C, generator=g).to(DEV).to(dt)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    count = N * H * W
    out = {}
    try:
        from rcf_amd import _lib
        for mode in (0, 2):                      # 2: the banded order whatever the tensor's size (default: from 192 MB up)
            ops.BN_FLAGS = _lib.BN_SWEEP_ALWAYS if mode else _lib.BN_SWEEP_OFF        # per-call flags (RCF_BN_SWEEP_*)
            rmean, rvar = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            mean, invstd = ops.bn_finalize(ops.bn_stats(x), count, 1e-5, 0.1, rmean, rvar)
            rmask = torch.empty(x.numel() // 4, dtype=torch.uint8, device=DEV)
            am = ops.new_amax(DEV) if not bf16 else None
            y = ops.bn_apply(x, mean, invstd, gamma, beta, True, residual=r, relu_mask=rmask, amax_out=am)
            s2 = ops.bn_bwd_reduce(dy, x, None, mean, invstd, True, relu_mask=rmask)
            dgam, dbet = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
            dres = torch.full_like(x, 0.25) if res else None
            gam = ops.new_amax(DEV) if not bf16 else None
            dx = ops.bn_bwd_apply(dy, x, None, mean, invstd, gamma, True, s2, count, dgam, dbet, dres=dres, res_beta=1 if res else 0,
                                  relu_mask=rmask, amax_out=gam)
            out[mode] = (y, rmask, am, s2, dx, dres, gam, dgam, dbet)
    finally:
        ops.BN_FLAGS = 0
    a, b = out[0], out[2]
    same = [torch.equal(a[i], b[i]) for i in (0, 1, 4)] + [a[2] is None or torch.equal(a[2], b[2]), not res or torch.equal(a[5], b[5])]
    e_s = float((a[3] - b[3]).abs().max() / a[3].abs().max())
    # dx: given the SAME sums it is bit-identical; with re-grouped fp64 sums it may move by an ulp
    report(f"bn row order C={C} rows={count} res={res} bf16={bf16}: y / mask / dx / range / dres identical {same}; sums rel diff {e_s:.1e}")
    assert same[0] and same[1] and same[3] and e_s < 1e-12
    tol = 2e-2 if bf16 else 1e-6
    assert float((a[4].float() - b[4].float()).abs().max() / a[4].float().abs().max()) < tol
    if res:
        assert torch.equal(a[5], b[5])


def test_maxpool(report):
    g = torch.Generator().manual_seed(3)
    for (N, C, H, W) in [(2, 64, 15, 22), (1, 8, 240, 427), (2, 4, 8, 8)]:
        x = torch.relu(torch.randn(N, C, H, W, generator=g))   # many exact ties at 0, like post-ReLU maps
        xd = x.double().requires_grad_(True)
        y = F.max_pool2d(xd, 3, 2, 1)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy.double())
        yg, am = ops.maxpool_fwd(to_nhwc(x))
        dx = ops.maxpool_bwd(to_nhwc(dy), am, (N, H, W, C))
        assert torch.equal(from_nhwc(yg).double(), y.detach())
        e = relerr(from_nhwc(dx), xd.grad)
        report(f"maxpool {(N, C, H, W)}: dx {e:.2e}")
        assert e < 1e-6


@pytest.mark.parametrize("Hi,Wi,Ho,Wo,align", [(60, 107, 120, 214, False), (15, 27, 30, 54, False),
                                               (7, 9, 20, 31, False), (24, 40, 6, 10, False), (12, 16, 24, 33, True),
                                               (30, 54, 15, 27, True), (5, 5, 5, 5, False)])
def test_resize_nhwc(Hi, Wi, Ho, Wo, align, report):
    g = torch.Generator().manual_seed(Hi * Wo)
    N, C = 2, 8
    x = torch.randn(N, C, Hi, Wi, generator=g)
    xd = x.double().requires_grad_(True)
    y = F.interpolate(xd, (Ho, Wo), mode="bilinear", align_corners=align)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    yg = ops.resize_nhwc_fwd(to_nhwc(x), (Ho, Wo), align)
    dx = ops.resize_nhwc_bwd(to_nhwc(dy), (Hi, Wi), align)
    e_y, e_dx = relerr(from_nhwc(yg), y), relerr(from_nhwc(dx), xd.grad)
    yp = ops.resize_nchw(x.to(DEV), (Ho, Wo), align)
    e_p = relerr(yp, y)
    report(f"resize {(Hi, Wi)}->{(Ho, Wo)} align={align}: y {e_y:.2e} dx {e_dx:.2e} planar {e_p:.2e}")
    assert max(e_y, e_dx, e_p) < 2e-5


@pytest.mark.parametrize("N,Hi,Wi,C,bf16", [(2, 60, 107, 64, False), (1, 2, 2, 8, False), (3, 7, 33, 24, False), (2, 15, 27, 64, True)])
def test_resize_exact_2x_kernels_are_bit_identical(N, Hi, Wi, C, bf16, report):
    """csrc/spatial.hip resize2x_*: the exact-2x forms of the bilinear resize (a thread makes the 2 x 2 outputs of one source
    pixel from 9 loads instead of 16) against the general kernels they replace: same taps, weights and expression tree, so
    the outputs must be BIT-identical, borders included (2 x 2 sources: every output clamps somewhere), fp32 and bf16"""
    g = torch.Generator().manual_seed(N * Hi + Wi)
    x = torch.randn(N, Hi, Wi, C, generator=g).to(DEV)
    dy = torch.randn(N, 2 * Hi, 2 * Wi, C, generator=g).to(DEV)
    if bf16:
        x, dy = x.bfloat16(), dy.bfloat16()
    res = {}
    for mode in (0, 1):                         # frame = -1: the general kernels on the whole tensor (a per-call choice)
        res[mode] = (ops.resize_nhwc_fwd(x, (2 * Hi, 2 * Wi), False, frame=0 if mode else -1),
                     ops.resize_nhwc_bwd(dy, (Hi, Wi), False, frame=0 if mode else -1))
    same = [torch.equal(a, b) for a, b in zip(res[0], res[1])]
    ref = F.interpolate(x.float().permute(0, 3, 1, 2).double(), scale_factor=2, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    e = float((res[1][0].double() - ref).abs().max() / ref.abs().max())
    report(f"resize exact 2x {(N, Hi, Wi, C)} bf16={bf16}: forward / backward identical to the general kernels {same}; forward vs float64 {e:.1e}")
    assert all(same) and e < (1e-2 if bf16 else 1e-6)


def test_layout_copy_colsum(report):
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 3, 10, 13, generator=g)
    nhwc = ops.nchw_to_nhwc(x.to(DEV), 4)
    assert torch.equal(nhwc[..., :3].cpu(), x.permute(0, 2, 3, 1)) and float(nhwc[..., 3].abs().max()) == 0.0
    back = ops.nhwc_to_nchw(nhwc, 3)
    assert torch.equal(back.cpu(), x)
    a = torch.randn(50, 16, generator=g).to(DEV)
    dst = torch.zeros(50, 40, device=DEV)
    ops.copy2d(a, 16, dst[:, 8:], 40, 50, 16)
    ops.copy2d(a, 16, dst[:, 8:], 40, 50, 16, beta=1)
    assert torch.equal(dst[:, 8:24], 2 * a) and float(dst[:, :8].abs().max()) == 0
    assert float(dst[:, 24:].abs().max()) == 0
    t = torch.randn(2, 31, 17, 16, generator=g)
    out = torch.ones(16, device=DEV)
    ops.colsum(t.to(DEV), out, beta=1)
    e = relerr(out, 1 + t.double().sum(dim=(0, 1, 2)))
    report(f"colsum: {e:.2e}")
    assert e < 1e-6


def test_adam_and_ema(report):
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(10007, generator=g)
    pr = p0.clone().double().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-4, weight_decay=1e-4)
    pg = p0.clone().to(DEV)
    m, v = torch.zeros_like(pg), torch.zeros_like(pg)
    for step in range(1, 4):
        gr = torch.randn(10007, generator=g)
        pr.grad = gr.double()
        opt.step()
        ops.adam_step(pg, gr.to(DEV), m, v, 1e-4, step, weight_decay=1e-4)
    e = float((pg.cpu().double() - pr.detach()).abs().max() / 3e-4)   # relative to the total update size
    d, s = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    dg = d.clone().to(DEV)
    ops.ema_update(dg, s.to(DEV), 0.999)
    e2 = relerr(dg, d * 0.999 + s * (1 - 0.999))
    report(f"adam: {e:.2e} (of update) ema: {e2:.2e}")
    assert e < 5e-3 and e2 < 1e-6


def test_warp_family_vs_golden(golden_dir, report):
    fx = np.load(os.path.join(golden_dir, "warp.npz"))
    x, y = torch.from_numpy(fx["x"]).to(DEV), torch.from_numpy(fx["y"]).to(DEV)
    for name in ("random", "integer", "outofrange"):
        f12 = torch.from_numpy(fx[f"{name}_f12"]).to(DEV)
        f21 = torch.from_numpy(fx[f"{name}_f21"]).to(DEV)
        wb = ops.flow_warp(x, f12, "border")
        wz = ops.flow_warp(x, f12, "zeros")
        e_b = float((wb.cpu() - torch.from_numpy(fx[f"{name}_warp_border"])).abs().max())
        e_z = float((wz.cpu() - torch.from_numpy(fx[f"{name}_warp_zeros"])).abs().max())
        ob = ops.occu_mask_backward(f21, 0.2)
        obi = ops.occu_mask_bidirection(f12, f21)
        m_b = float((ob.cpu() != torch.from_numpy(fx[f"{name}_occ_back"]).float()).float().mean())
        m_bi = float((obi.cpu() != torch.from_numpy(fx[f"{name}_occ_bidir"]).float()).float().mean())
        occ_ref = 1 - torch.from_numpy(fx[f"{name}_occ_back"]).float().to(DEV)
        wb_ref = torch.from_numpy(fx[f"{name}_warp_border"])
        ph = ops.photometric_loss(y, wb_ref.to(DEV), occ_ref)
        e_p = abs(float(ph) - float(fx[f"{name}_photo"])) / abs(float(fx[f"{name}_photo"]))
        l1 = ops.warp_l1_residual(y, x, f12, occ_ref, "border").cpu()
        l1_ref = ((y.cpu() - wb_ref).abs().sum(1, keepdim=True) * occ_ref.cpu()).double().sum()
        e_l1 = abs(float(l1[0]) - float(l1_ref)) / float(l1_ref)
        report(f"warp {name}: border {e_b:.2e} zeros {e_z:.2e} occ_back mism {m_b:.2e} occ_bidir mism {m_bi:.2e} "
               f"photo {e_p:.2e} fusedL1 {e_l1:.2e}")
        assert e_b < 2e-5 and e_z < 2e-5 and e_p < 1e-4 and e_l1 < 1e-5
        assert m_b == 0.0 and m_bi == 0.0, "occlusion masks are 0/1 decisions: bit-exact against the reference's"


def test_resize_frame_variants(report):
    """frame-restricted bilinear resize: fwd writes exactly the frame's pixels of the full resize; bwd equals the
    full bwd of a gradient zeroed off the frame (and never reads it there: NaNs planted in the interior)"""
    g = torch.Generator().manual_seed(9)
    N, hi, wi, C, t = 2, 9, 13, 8, 4
    x = torch.randn(N, hi, wi, C, generator=g).to(DEV)
    full = ops.resize_nhwc_fwd(x, (2 * hi, 2 * wi), False)
    out = torch.full_like(full, 7.0)
    ops.resize_nhwc_fwd(x, (2 * hi, 2 * wi), False, out=out, frame=t)
    m = torch.ones(2 * hi, 2 * wi, dtype=torch.bool)
    m[t:-t, t:-t] = False
    m = m.to(DEV)
    ok_f = bool(torch.equal(out[:, m], full[:, m])) and float((out[:, ~m] - 7.0).abs().max()) == 0.0
    dy = torch.randn(N, 2 * hi, 2 * wi, C, generator=g).to(DEV)
    masked = dy.clone()
    masked[:, ~m] = 0
    ref = ops.resize_nhwc_bwd(masked, (hi, wi), False)
    poisoned = dy.clone()
    poisoned[:, ~m] = float("nan")
    got = ops.resize_nhwc_bwd(poisoned, (hi, wi), False, frame=t)
    e_b = float((got - ref).abs().max())
    report(f"resize frame: fwd exact {ok_f}, bwd |d| {e_b:.2e}")
    assert ok_f and e_b < 1e-6


@pytest.mark.parametrize("hi,wi,t,dt", [(24, 30, 5, torch.float32), (25, 31, 6, torch.float32), (60, 107, 13, torch.float32),
                                        (24, 30, 5, torch.bfloat16), (33, 28, 7, torch.bfloat16)])
def test_resize2x_frame_forms(hi, wi, t, dt, report):
    """round 6: the border frame of an exact 2x up-sampling (decode_head2's commuted conv) runs on the 2x kernels over the source
    pixels / 2 x 2 input blocks that reach the frame (grids of the frame's size): forward = exactly the frame's pixels of the whole
    resize (bit for bit, the rest untouched); backward, accumulating = the whole backward of a gradient zeroed off the frame added to
    what was there (NaNs planted off the frame are never read), bit for bit against the general kernel's frame form on the same data"""
    g = torch.Generator().manual_seed(hi * 100 + wi + t)
    N, C = 2, 16
    x = torch.randn(N, hi, wi, C, generator=g).to(DEV).to(dt)
    full = ops.resize_nhwc_fwd(x, (2 * hi, 2 * wi), False)
    out = torch.full_like(full, 7.0)
    ops.resize_nhwc_fwd(x, (2 * hi, 2 * wi), False, out=out, frame=t)
    m = torch.ones(2 * hi, 2 * wi, dtype=torch.bool)
    m[t:-t, t:-t] = False
    m = m.to(DEV)
    ok_f = bool(torch.equal(out[:, m], full[:, m])) and float((out[:, ~m].float() - 7.0).abs().max()) == 0.0
    dy = torch.randn(N, 2 * hi, 2 * wi, C, generator=g).to(DEV).to(dt)
    masked = dy.clone()
    masked[:, ~m] = 0
    base = torch.randn(N, hi, wi, C, generator=g).to(DEV).to(dt)
    ref = base.clone()
    ops.resize_nhwc_bwd(masked, (hi, wi), False, out=ref, beta=1)                       # whole tensor, 2x kernel
    poisoned = dy.clone()
    poisoned[:, ~m] = float("nan")
    got = base.clone()
    ops.resize_nhwc_bwd(poisoned, (hi, wi), False, out=got, beta=1, frame=t)
    same = bool(torch.equal(got, ref))
    e_b = float((got.float() - ref.float()).abs().max())
    report(f"resize 2x frame forms ({hi}x{wi}, frame {t}, {dt}): forward exact {ok_f}; backward (accumulating) identical to the whole "
           f"backward of the masked gradient: {same} (max |d| {e_b:.1e})")
    assert ok_f and same


@pytest.mark.parametrize("H,W", [(6, 10), (7, 9), (12, 854)])
def test_warp_odd_shapes_vs_oracle(H, W, report):
    """flow_warp / fused warp+L1 on shapes whose rows are not multiples of the wavefront or of 4 (and the real
    854-pixel row), against the oracle restatement of utils/warp_utils.py:84-94"""
    import rcf_torch as orc
    g = torch.Generator().manual_seed(H * 1000 + W)
    B, C = 2, 3
    x, y = torch.rand(B, C, H, W, generator=g), torch.rand(B, C, H, W, generator=g)
    fl = torch.randn(B, 2, H, W, generator=g) * 2.5
    occ = (torch.rand(B, 1, H, W, generator=g) > 0.3).float()
    for pad in ("border", "zeros"):
        ref = orc.flow_warp(x, fl, pad=pad)
        got = ops.flow_warp(x.to(DEV), fl.to(DEV), pad).cpu()
        e = float((got - ref).abs().max())
        l1 = ops.warp_l1_residual(y.to(DEV), x.to(DEV), fl.to(DEV), occ.to(DEV), pad).cpu()
        l1_ref = ((y - ref).abs().sum(1, keepdim=True) * occ).double().sum()
        e1 = abs(float(l1[0]) - float(l1_ref)) / float(l1_ref)
        report(f"warp paths {H}x{W} {pad}: warp {e:.2e} fusedL1 {e1:.2e} occ sum {float(l1[1])} vs {float(occ.sum())}")
        assert e < 2e-5 and e1 < 1e-5 and float(l1[1]) == float(occ.sum())


@pytest.mark.parametrize("H,W", [(33, 300), (16, 256), (50, 857), (7, 9)])
def test_warp_tile_kernels_are_bit_identical_to_the_per_pixel_kernels(H, W, report):
    """the RGB / border tile kernels (exact 3-operation division, folded border taps, packed position arithmetic) against
    the per-pixel kernels: same bits, including positions far outside the image, NaN flows and exact-integer positions"""
    from rcf_amd import _lib
    g = torch.Generator().manual_seed(H * 7 + W)
    B = 3
    x, y = torch.rand(B, 3, H, W, generator=g).to(DEV), torch.rand(B, 3, H, W, generator=g).to(DEV)
    fl = torch.randn(B, 2, H, W, generator=g) * 6.0
    fl[0, :, :3, :5] = 1e9
    fl[1, :, -3:, -5:] = -1e9
    fl[2, 0, 2, 2] = float("nan")
    fl[2, :, 4:6] = fl[2, :, 4:6].round()                          # integer positions: weights exactly 0 / 1
    fl[2, 0, :, -1] = 0.0                                          # x = W-1 exactly: the folded tap
    fl[2, 1, -1, :] = 0.0
    fl = fl.to(DEV)
    occ = (torch.rand(B, 1, H, W, generator=g) > 0.3).float().to(DEV)
    res = {}
    for v in (0, 1):                        # 0: pad_mode | RCF_WARP_PER_PIXEL, a per-call choice
        pad = "border" if v else "border_per_pixel"
        res[v] = (ops.flow_warp(x, fl, pad).cpu(), ops.warp_l1_residual(y, x, fl, occ, pad).cpu(),
                  ops.warp_l1_residual(y, x, fl, None, pad).cpu())
    same_w = torch.equal(res[0][0], res[1][0])
    rel = [abs(float(res[0][k][0]) - float(res[1][k][0])) / abs(float(res[0][k][0])) for k in (1, 2)]
    report(f"warp tile vs per-pixel {H}x{W}: warped identical {same_w}, fused L1 rel diff {rel[0]:.1e} / {rel[1]:.1e} (fp64 sums, order differs)")
    assert same_w and max(rel) < 1e-12 and float(res[0][1][1]) == float(res[1][1][1])


def test_warp_backward(report):
    g = torch.Generator().manual_seed(21)
    B, C, H, W = 2, 3, 17, 23
    x = torch.randn(B, C, H, W, generator=g)
    fl = torch.randn(B, 2, H, W, generator=g) * 3
    dout = torch.randn(B, C, H, W, generator=g)
    for pad in ("border", "zeros"):
        xd, fd = x.double().requires_grad_(True), fl.double().requires_grad_(True)
        xs = torch.arange(W, dtype=torch.float64).view(1, 1, W).expand(B, H, W)
        ys = torch.arange(H, dtype=torch.float64).view(1, H, 1).expand(B, H, W)
        gr = torch.stack([xs, ys], 1) + fd
        gn = torch.stack([2.0 * gr[:, 0] / (W - 1) - 1.0, 2.0 * gr[:, 1] / (H - 1) - 1.0], dim=-1)
        out = F.grid_sample(xd, gn, mode="bilinear", padding_mode=pad, align_corners=True)
        out.backward(dout.double())
        dx, dfl = ops.flow_warp_bwd(x.to(DEV), fl.to(DEV), dout.to(DEV), pad)
        e1, e2 = relerr(dx, xd.grad), relerr(dfl, fd.grad)
        report(f"warp bwd {pad}: dx {e1:.2e} dflow {e2:.2e}")
        assert e1 < 2e-5 and e2 < 2e-4


def test_dropout2d_scale_draw(report):
    """rcf_dropout2d_scale_f32 (round 6: the step's Dropout2d draw no longer goes through torch.bernoulli): values in {0, 1/(1-p)},
    the dropped fraction within 4 sigma of p, a (seed, size) pair always gives the same draw, another seed another one, no visible
    correlation between neighbouring elements; and torch.manual_seed governs the draw an FCNHead takes in training mode."""
    n, C, p = 64, 256, 0.1
    a = ops.dropout2d_scale(n, C, p, 1234, DEV)
    b = ops.dropout2d_scale(n, C, p, 1234, DEV)
    c = ops.dropout2d_scale(n, C, p, 1235, DEV)
    vals = torch.unique(a).cpu().tolist()
    frac = float((a == 0).float().mean())
    sigma = (p * (1 - p) / (n * C)) ** 0.5
    z = (a == 0).float()
    corr = float(((z[:, 1:] - p) * (z[:, :-1] - p)).mean() / (p * (1 - p)))
    big = ops.dropout2d_scale(4096, 512, 0.5, 7, DEV)
    frac_big = float((big == 0).float().mean())
    report(f"dropout2d draw: values {vals}, dropped {frac:.4f} (p = {p}, sigma {sigma:.4f}); same seed identical {bool(torch.equal(a, b))}, "
           f"other seed differs on {float((a != c).float().mean()):.3f} of the planes; lag-1 correlation {corr:+.4f}; p = 0.5 over 2 M planes: {frac_big:.5f}")
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1 / (1 - p)) < 1e-6
    assert abs(frac - p) < 4 * sigma and torch.equal(a, b) and not torch.equal(a, c) and abs(corr) < 0.05
    assert abs(frac_big - 0.5) < 4 * (0.25 / (4096 * 512)) ** 0.5
    # the model's heads: the draw follows torch's host generator
    from rcf_amd import backbone
    seen = []
    orig = ops.dropout2d_scale
    ops.dropout2d_scale = lambda *a_, **k_: (seen.append(orig(*a_, **k_)) or seen[-1])
    try:
        head = backbone.FCNHead(64, 32, num_classes=4, num_convs=1, concat_input=False, dropout_ratio=0.1,
                                norm_cfg=dict(type="BN", requires_grad=True)).to(DEV).train()
        from rcf_amd.layers import Act, Tape
        x = torch.randn(2, 12, 16, 64, device=DEV)
        for s_ in (5, 5, 6):
            torch.manual_seed(s_)
            head.fwd([Act(x, needs_grad=False)], Tape(enabled=False))
    finally:
        ops.dropout2d_scale = orig
    assert len(seen) == 3 and torch.equal(seen[0], seen[1]) and not torch.equal(seen[0], seen[2])


def test_ema_update_of_a_state_dict_in_one_launch(report):
    """round 6: momentum_update_param_and_buffer (utils/model_utils.py:33-38) runs every entry of the state dict through ONE launch
    (rcf_ema_update_multi: chunk table built once per module pair).  Float entries: the bits of the per-tensor kernel
    (d m + s (1.0f - m), no fused multiply-add); channels_last conv weights in their own memory order; the int64
    num_batches_tracked counters exactly as torch's int64 * python-float arithmetic leaves them (the reference's truncation);
    the table follows a re-allocation of the tensors."""
    from rcf_amd import layers
    from rcf_amd.model import momentum_update_param_and_buffer
    torch.manual_seed(3)

    def make():
        m = torch.nn.Sequential()
        m.add_module("c1", layers.Conv2d(8, 16, 3))
        m.add_module("b1", layers.BatchNorm2d(16))
        m.add_module("c2", layers.Conv2d(16, 200, 1))
        m.add_module("b2", layers.BatchNorm2d(200))
        return m.to(DEV)
    src, dst = make(), make()
    with torch.no_grad():
        for mod, seed in ((src, 1), (dst, 2)):
            g = torch.Generator().manual_seed(seed)
            for t in list(mod.parameters()) + list(mod.buffers()):
                if t.dtype == torch.float32:
                    t.copy_(torch.randn(t.shape, generator=g).to(DEV))
        src.b1.num_batches_tracked.fill_(1234567)
        dst.b1.num_batches_tracked.fill_(41)
        src.b2.num_batches_tracked.fill_(3)
        dst.b2.num_batches_tracked.fill_(100000)
    for rnd, m in enumerate((0.999, 0.9, 0.5)):
        want = {}
        mf = torch.tensor(m, dtype=torch.float32, device=DEV)
        for (k, s_), (_, d_) in zip(src.state_dict().items(), dst.state_dict().items()):
            if d_.dtype == torch.float32:
                want[k] = d_ * mf + s_ * (torch.tensor(1.0, dtype=torch.float32, device=DEV) - mf)
            else:
                want[k] = (d_ * m + s_ * (1.0 - m)).to(d_.dtype)            # the reference's expression on the int64 counter
        if rnd == 2:                                                        # re-allocate: the cached table must be rebuilt
            with torch.no_grad():
                dst.c2.weight.data = dst.c2.weight.data.clone()
                want["c2.weight"] = dst.c2.weight.data * mf + src.c2.weight.data * (torch.tensor(1.0, dtype=torch.float32, device=DEV) - mf)
        momentum_update_param_and_buffer(src, dst, m)
        got = dst.state_dict()
        bad = [k for k in want if not torch.equal(got[k], want[k])]
        report(f"EMA of a state dict in one launch, m = {m}: {len(want)} entries ({dst._ema_plan.count} chunks), mismatching: {bad}; "
               f"counters {int(got['b1.num_batches_tracked'])}, {int(got['b2.num_batches_tracked'])}")
        assert not bad
