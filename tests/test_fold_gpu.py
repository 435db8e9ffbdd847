"""GPU: the folded 1x1 conv -> training-mode batch norm (-> + residual -> ReLU) of the bf16 step (csrc/foldbn.hip, the fused
epilogues of csrc/igemm_bf16.hip, layers.conv_bn_fold) against float64 torch autograd, and against the three-pass form it
replaces (conv / statistics / apply; reduce / apply / data and weight gradient): the fold must be at least as close to float64."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import rcf_amd
from rcf_amd import backbone, layers, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def q(t):
    return t.to(BF).to(t.dtype)


def nhwc(x_nchw, dtype=BF):
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


@pytest.mark.parametrize("case", [(2, 64, 256, 19, 23, True, True), (1, 256, 1024, 17, 31, True, True), (2, 128, 512, 9, 13, False, True),
                                  (2, 64, 64, 11, 15, True, False), (1, 512, 2048, 12, 11, True, True), (2, 32, 128, 10, 9, False, False)])
def test_conv_affine_epilogue(case, report):
    """y = [relu](conv1x1(x, w) * scale + shift [+ residual]) in the conv's epilogue vs float64"""
    N, K, C, H, W, relu, with_res = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = q(torch.randn(N, K, H, W, generator=g))
    w = q(torch.randn(C, K, 1, 1, generator=g) * (2.0 / K) ** 0.5)
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    res = q(torch.randn(N, C, H, W, generator=g))
    ref = F.conv2d(x.double(), w.double()) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    if with_res:
        ref = ref + res.double()
    if relu:
        ref = ref.clamp_min(0)
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last)
    y = ops.conv2d_fwd_affine_bf16(nhwc(x), wd, ops.weight_bf16(wd), scale.to(DEV), shift.to(DEV), nhwc(res) if with_res else None, relu)
    e = relerr(y.float(), ref.permute(0, 2, 3, 1))
    report(f"conv+affine epilogue {case}: {e:.2e}")
    assert e < 5e-3
    if relu:
        assert float(y.float().min()) >= 0.0
        # the sign bits the tile leaves (one ballot per accumulator register), used as the mask of a data gradient over the same
        # [rows][C] tensor: bit-identical to masking by y itself
        y2, bits = ops.conv2d_fwd_affine_bf16(nhwc(x), wd, ops.weight_bf16(wd), scale.to(DEV), shift.to(DEV), nhwc(res) if with_res else None,
                                              relu, want_bits=True)
        assert torch.equal(y2, y)
        Cd = 64
        dy = q(torch.randn(N, Cd, H, W, generator=g))
        w2 = q(torch.randn(Cd, C, 1, 1, generator=g) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last)
        outs = []
        for mb in (None, bits):
            o = torch.zeros((N, H, W, C), dtype=BF, device=DEV)
            _, cs = ops.conv2d_dgrad_masked_bf16(nhwc(dy), w2, (N, H, W, C), ops.weight_bf16(w2, True), y, o, beta=0, mask_bits=mb)
            outs.append((o, cs))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("case", [(2, 64, 256, 19, 23, 0), (1, 256, 1024, 17, 31, 1), (2, 512, 2048, 7, 9, 1), (1, 64, 64, 21, 17, 0),
                                  (2, 128, 128, 13, 11, 1)])
def test_masked_dgrad_and_mask_colsum(case, report):
    """dx = y > 0 ? dgrad (+ dx) : 0 with its column sums from the data gradient's epilogue, and the stand-alone mask + colsum pass"""
    N, Cout, Cin, H, W, beta = case
    g = torch.Generator().manual_seed(sum(case))
    dy = q(torch.randn(N, Cout, H, W, generator=g))
    w = q(torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cout) ** 0.5)
    ymask = q(torch.relu(torch.randn(N, Cin, H, W, generator=g)))
    old = q(torch.randn(N, Cin, H, W, generator=g))
    ref = F.conv_transpose2d(dy.double(), w.double())
    if beta:
        ref = ref + old.double()
    ref = torch.where(ymask.double() > 0, ref, torch.zeros_like(ref))
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last)
    out = nhwc(old).clone()
    _, cs = ops.conv2d_dgrad_masked_bf16(nhwc(dy), wd, (N, H, W, Cin), ops.weight_bf16(wd, True), nhwc(ymask), out, beta=beta)
    e = relerr(out.float(), ref.permute(0, 2, 3, 1))
    e_cs = relerr(cs[:Cin], ref.sum(dim=(0, 2, 3)))
    # the stand-alone pass, out of place and in place
    d2 = nhwc(old)
    g2, cs2 = ops.relu_mask_colsum(d2, nhwc(ymask))
    ref2 = torch.where(ymask.double() > 0, old.double(), torch.zeros_like(old.double()))
    e2 = relerr(g2.float(), ref2.permute(0, 2, 3, 1))
    e2_cs = relerr(cs2[:Cin], ref2.sum(dim=(0, 2, 3)))
    g3, cs3 = ops.relu_mask_colsum(d2, nhwc(ymask), out=d2)
    assert g3.data_ptr() == d2.data_ptr() and torch.equal(g3, g2) and torch.equal(cs3, cs2)
    report(f"masked dgrad {case}: dx {e:.2e} colsum {e_cs:.2e}; mask pass {e2:.2e} colsum {e2_cs:.2e}")
    assert e < 6e-3 and e_cs < 2e-3 and e2 == 0.0 and e2_cs < 1e-6


@pytest.mark.parametrize("case", [(3000, 64, 256), (2111, 256, 1024), (1500, 512, 2048), (4000, 128, 64), (2500, 128, 512)])
def test_fold_small_kernels_vs_float64(case, report):
    """Gram / P / statistics / finalize / backward sums / backward operands against the same algebra in float64"""
    n, K, C = case
    g = torch.Generator().manual_seed(n + K)
    X = q(torch.relu(torch.randn(n, K, generator=g) + 0.3))
    Wm = torch.randn(C, K, generator=g) * (2.0 / K) ** 0.5            # fp32 master: the kernels round it to bf16 themselves
    Wq = q(Wm).double()
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    eps, mom = 1e-5, 0.1
    Xd = X.double()
    S64, A64 = Xd.t() @ Xd, Xd.sum(0)
    P64 = Wq @ S64
    sz, szz = Wq @ A64, (Wq * P64).sum(1)
    mu, var = sz / n, szz / n - (sz / n) ** 2
    inv = (var + eps).rsqrt()
    xa = X.to(BF).to(DEV).view(1, 1, n, K)
    wd = Wm.view(C, K, 1, 1).to(DEV).contiguous(memory_format=torch.channels_last)
    S = ops.gram_bf16(xa)
    A1 = ops.bn_stats(xa)
    P, sums = ops.fold_fwd(S, A1, wd, rows=n)
    bn = layers.BatchNorm2d(C).to(DEV)
    bn.weight.data.copy_(gam)
    bn.bias.data.copy_(bet)
    mean, invstd, scale, shift = ops.fold_finalize(sums, n, bn)
    # the one-launch form (local statistics): the same constants, the same running statistics
    bn2 = layers.BatchNorm2d(C).to(DEV)
    bn2.weight.data.copy_(gam)
    bn2.bias.data.copy_(bet)
    P2, outs2 = ops.fold_fwd(S, A1, wd, bn2, n)
    assert torch.equal(P2, P) and all(torch.equal(a_, b_) for a_, b_ in zip(outs2, (mean, invstd, scale, shift)))
    assert torch.equal(bn2.running_mean, bn.running_mean) and torch.equal(bn2.running_var, bn.running_var) and int(bn2.num_batches_tracked) == 1
    # P is the CENTRED product W (S - A1 A1^T / n) (round 6), followed by the local means of z
    e_S, e_P = relerr(S.view(K, K), S64), relerr(P[:C * K].view(C, K), P64 - mu[:, None] * A64[None])
    assert relerr(P[C * K:], mu) < 1e-6
    e_sums = relerr(sums, torch.cat([sz, szz]))
    e_mu, e_inv = relerr(mean, mu), relerr(invstd, inv)
    e_aff = max(relerr(scale, gam.double() * inv), relerr(shift, bet.double() - mu * gam.double() * inv))
    rm_ref, rv_ref = mom * mu, (1 - mom) + mom * var * n / (n - 1)
    e_run = max(relerr(bn.running_mean, rm_ref), relerr(bn.running_var, rv_ref))
    assert int(bn.num_batches_tracked) == 1
    # backward
    gq = q(torch.randn(n, C, generator=g) * (torch.rand(n, C, generator=g) > 0.5))
    gd = gq.double()
    sg = gd.sum(0)
    G64 = gd.t() @ Xd
    sgz = inv * ((Wq * G64).sum(1) - mu * sg)
    a = gam.double() * inv
    m, qq = sg / n, sgz / n
    dW64 = a[:, None] * (G64 - m[:, None] * A64[None] - (qq * inv)[:, None] * (P64 - mu[:, None] * A64[None]))
    d = a * inv * qq
    T64 = Wq.t() @ (d[:, None] * Wq)
    c064 = (d * mu - a * m) @ Wq
    ga = gq.to(BF).to(DEV).view(1, 1, n, C)
    G = torch.empty((C, K, 1, 1), dtype=torch.float32, device=DEV)
    ops.conv2d_wgrad_bf16(xa, ga, wd, G, 1, 0, 1, beta=0)
    cs = torch.cat([sg, torch.zeros(C, dtype=torch.float64)]).to(DEV)
    sums2 = ops.fold_bwd_sums(G, wd, cs, mean, invstd)
    dW = torch.zeros_like(wd)
    dgam, dbet = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    negT, c0 = ops.fold_bwd_prepare(G, P, A1, wd, sums2, None, n, mean, invstd, bn.weight, dW, dgam, dbet)
    wg_t = ops.fold_wg(wd, scale)
    e_s2 = relerr(sums2, torch.cat([sg, sgz]))
    e_dW = relerr(dW.view(C, K), dW64)
    e_dg, e_db = relerr(dgam, sgz), relerr(dbet, sg)
    e_c0 = relerr(c0, c064)
    # the derived bf16 operands, read back through the layouts the conv kernels read: (row j, k) at ((k >> 5) rows + j) 32 + (k & 31)
    wgt = wg_t.view(BF).view(C // 32, K, 32).permute(1, 0, 2).reshape(K, C).float().cpu()        # [k][c] = a_c W[c][k]
    e_wg = relerr(wgt, (a[:, None] * Wq).t())
    nT = negT.view(BF).view(K // 32, K, 32).permute(1, 0, 2).reshape(K, K).float().cpu()         # [row k][kk j] = -T[j][k]
    e_T = relerr(nT, -T64.t())
    report(f"fold kernels {case}: S {e_S:.1e} P {e_P:.1e} sums {e_sums:.1e} mean {e_mu:.1e} invstd {e_inv:.1e} affine {e_aff:.1e} running "
           f"{e_run:.1e} | sums2 {e_s2:.1e} dW {e_dW:.1e} dgamma {e_dg:.1e} dbeta {e_db:.1e} c0 {e_c0:.1e} Wg^T {e_wg:.1e} -T {e_T:.1e}")
    assert max(e_S, e_P, e_sums, e_mu, e_inv, e_aff, e_run) < 2e-5
    assert max(e_s2, e_dW, e_dg, e_db, e_c0) < 2e-4
    assert e_wg < 5e-3 and e_T < 5e-3


def test_fold_small_kernels_two_rank_recombination(report):
    """the SyncBN form of the fold without a process group: the rows split in two "ranks", each runs rcf_fold_fwd_f32 on ITS
    moments (sums only), the sums are added as the all-reduce would, rcf_fold_finalize_f32 finalizes on the global count, and each
    rank's rcf_fold_bwd_prepare_f32 re-centres its P on the GLOBAL mean: statistics and the summed dW / dgamma / dbeta against the
    float64 algebra over all rows (this branch had no test of its own)."""
    n, K, C = 3001, 128, 256
    g = torch.Generator().manual_seed(12)
    X = q(torch.relu(torch.randn(n, K, generator=g) + 0.8))
    X[n // 2:] += 0.5                                          # the two halves have different means: mloc != the global mean
    X = q(X)
    Wm = torch.randn(C, K, generator=g) * (2.0 / K) ** 0.5
    Wq, Xd = q(Wm).double(), X.double()
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    Z = Xd @ Wq.t()
    mu, var = Z.mean(0), Z.var(0, unbiased=False)
    inv = (var + 1e-5).rsqrt()
    gq = q(torch.randn(n, C, generator=g) * (torch.rand(n, C, generator=g) > 0.5))
    gd = gq.double()
    sg, G64, A64 = gd.sum(0), gd.t() @ Xd, Xd.sum(0)
    sgz = inv * ((Wq * G64).sum(1) - mu * sg)
    a, m, qq = gam.double() * inv, sg / n, sgz / n
    dW64 = a[:, None] * (G64 - m[:, None] * A64[None] - (qq * inv)[:, None] * ((Wq @ (Xd.t() @ Xd)) - mu[:, None] * A64[None]))
    wd = Wm.view(C, K, 1, 1).to(DEV).contiguous(memory_format=torch.channels_last)
    bn = layers.BatchNorm2d(C).to(DEV)
    bn.weight.data.copy_(gam)
    bn.bias.data.copy_(bet)
    halves = [(0, n // 2), (n // 2, n)]
    ranks, sums = [], None
    for lo, hi in halves:
        xa = X[lo:hi].to(BF).to(DEV).view(1, 1, hi - lo, K).contiguous()
        S, A1 = ops.gram_bf16(xa), ops.bn_stats(xa)
        P, s_r = ops.fold_fwd(S, A1, wd, rows=hi - lo)
        sums = s_r.clone() if sums is None else sums + s_r
        ranks.append((xa, A1, P))
    mean, invstd, scale, shift = ops.fold_finalize(sums, n, bn)
    e_mu, e_inv = relerr(mean, mu), relerr(invstd, inv)
    s2, parts = None, []
    for (lo, hi), (xa, A1, P) in zip(halves, ranks):
        ga = gq[lo:hi].to(BF).to(DEV).view(1, 1, hi - lo, C).contiguous()
        G = torch.empty((C, K, 1, 1), dtype=torch.float32, device=DEV)
        ops.conv2d_wgrad_bf16(xa, ga, wd, G, 1, 0, 1, beta=0)
        cs = torch.cat([gd[lo:hi].sum(0), torch.zeros(C, dtype=torch.float64)]).to(DEV)
        s2_r = ops.fold_bwd_sums(G, wd, cs, mean, invstd)
        s2 = s2_r.clone() if s2 is None else s2 + s2_r
        parts.append((G, s2_r))
    dW = torch.zeros_like(wd)
    dgam, dbet = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    for (xa, A1, P), (G, s2_r) in zip(ranks, parts):
        ops.fold_bwd_prepare(G, P, A1, wd, s2, s2_r, n, mean, invstd, bn.weight, dW, dgam, dbet)
    e_dW, e_dg, e_db = relerr(dW.view(C, K), dW64), relerr(dgam, sgz), relerr(dbet, sg)
    report(f"fold kernels over two 'ranks' ({n // 2} + {n - n // 2} rows, means apart): mean {e_mu:.1e} invstd {e_inv:.1e} | summed dW {e_dW:.1e} "
           f"dgamma {e_dg:.1e} dbeta {e_db:.1e}")
    assert max(e_mu, e_inv) < 2e-5 and max(e_dW, e_dg, e_db) < 2e-4


def test_fold_statistics_of_badly_conditioned_channels(report):
    """the folded norm took its variance as w^T S w / n - mean^2 with S and P = W S accumulated in fp32 -- for a
    channel whose |mean| dwarfs its standard deviation the subtraction cancels what fp32 kept.  Since round 6 the Gram matrix is
    centred in fp64 before the contraction (P = W (S - A1 A1^T / n)), so var = w . P / n has nothing to cancel.  Channels built to
    hurt: an average of all inputs (mean^2 / var ~ K x 9), the same with a large offset input, a nearly dead channel (var ~ eps),
    gamma = 0 and 1e-8; against float64, and against the UNFOLDED path (conv -> bn_stats -> finalize on the bf16 z)."""
    n, K, C = 6000, 256, 64
    g = torch.Generator().manual_seed(91)
    X = q(torch.relu(torch.randn(n, K, generator=g) + 3.0))                  # mean 3, std 1: every input well away from zero
    Wm = torch.randn(C, K, generator=g) * (2.0 / K) ** 0.5
    Wm[0] = 1.0 / K                                                           # z_0 = the average of 256 inputs: mean 3, std 1/16
    Wm[1] = 1.0 / K + torch.randn(K, generator=g) * 1e-3
    Wm[2] = torch.randn(K, generator=g) * 1e-4                                # nearly dead: var ~ 1e-6, below eps = 1e-5
    Wm[3] = 4.0 / K                                                           # mean 12, std 1/4
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    gam[4], gam[5] = 0.0, 1e-8
    Wq, Xd = q(Wm).double(), X.double()
    Z = Xd @ Wq.t()
    mu, var = Z.mean(0), Z.var(0, unbiased=False)
    inv = (var + 1e-5).rsqrt()
    xa = X.to(BF).to(DEV).view(1, 1, n, K)
    wd = Wm.view(C, K, 1, 1).to(DEV).contiguous(memory_format=torch.channels_last)
    bn = layers.BatchNorm2d(C).to(DEV)
    bn.weight.data.copy_(gam)
    bn.bias.data.copy_(bet)
    S, A1 = ops.gram_bf16(xa), ops.bn_stats(xa)
    P, (mean, invstd, scale, shift) = ops.fold_fwd(S, A1, wd, bn, n)
    # the same statistics from the uncentred fp32 moments, as the kernels computed them until round 5 (restated in torch fp32)
    S32, A32, W32 = S.view(K, K).float().cpu(), A1[:K].float().cpu(), q(Wm)
    P32 = W32 @ S32
    var_old = ((W32 * P32).sum(1).double() / n - ((W32.double() @ A1[:K].cpu()) / n) ** 2).clamp_min(0)
    inv_old = (var_old + 1e-5).rsqrt()
    # the unfolded path: statistics of the bf16-rounded conv output
    zb = q(Z.float())
    inv_unf = (zb.double().var(0, unbiased=False) + 1e-5).rsqrt()
    rel = lambda a, b: ((a.double().cpu() - b).abs() / b.abs()).numpy()
    e_new, e_old, e_unf = rel(invstd, inv), rel(inv_old, inv), rel(inv_unf, inv)
    ratio = (mu ** 2 / var)[:4]
    report(f"fold statistics, badly conditioned channels (mean^2 / var = {[f'{v:.0f}' for v in ratio.tolist()]}): invstd error vs float64, channels 0-3: "
           f"centred {[f'{v:.1e}' for v in e_new[:4]]}, uncentred fp32 form {[f'{v:.1e}' for v in e_old[:4]]}, unfolded (bf16 z) "
           f"{[f'{v:.1e}' for v in e_unf[:4]]}; worst over all {C} channels: centred {e_new.max():.1e}, uncentred {e_old.max():.1e}; "
           f"mean {relerr(mean, mu):.1e}; scale of gamma = 0 / 1e-8: {float(scale[4]):.1e} / {float(scale[5]):.1e}")
    assert relerr(mean, mu) < 1e-6
    assert e_new.max() < 2e-4, e_new.max()                    # (the uncentred form: percent on channels 0, 1, 3)
    assert e_new[[0, 1, 3]].max() < 0.1 * max(e_old[[0, 1, 3]].max(), 1e-3)
    assert float(scale[4]) == 0.0 and abs(float(scale[5]) - 1e-8 * float(inv[5])) < 1e-12 and torch.isfinite(shift).all()


class RefBottleneck(torch.nn.Module):
    """float64 torch restatement of models/resnet.py:262-302 (train-mode norms), weights copied from the HIP module"""

    def __init__(self, blk):
        super().__init__()
        self.blk = blk

    def forward(self, x):
        b = self.blk

        def cbn(conv, norm, t, relu):
            z = F.conv2d(t, self.p[conv.weight], None, conv.stride, conv.padding, conv.dilation)
            z = F.batch_norm(z, None, None, self.p[norm.weight], self.p[norm.bias], True, 0.1, norm.eps)
            return z.clamp_min(0) if relu else z
        o = cbn(b.conv1, b.bn1, x, True)
        o = cbn(b.conv2, b.bn2, o, True)
        o = cbn(b.conv3, b.bn3, o, False)
        idt = x if b.downsample is None else cbn(getattr(b.downsample, "0"), getattr(b.downsample, "1"), x, False)
        return (o + idt).clamp_min(0)


def _run_stage(stage, x, dy, fold):
    saved = layers.SCHED.fold_bn
    layers.SCHED.fold_bn = fold
    try:
        for p_ in stage.parameters():
            p_.grad = None
        ops.weights_changed()
        tape = layers.Tape(act_dtype=BF)
        xa = layers.Act(nhwc(x))
        ya = stage.fwd(xa, tape, None)
        ya.grad = nhwc(dy)
        tape.backward()
        torch.cuda.synchronize()
        grads = {n: p_.grad.detach().float().cpu().clone() for n, p_ in stage.named_parameters()}
        return ya.t.float().cpu(), xa.grad.float().cpu(), grads
    finally:
        layers.SCHED.fold_bn = saved


@pytest.mark.parametrize("geom", [(2, 64, 64, 1, 1, 24, 31, 3), (2, 256, 128, 1, 2, 16, 21, 2), (1, 512, 256, 1, 2, 14, 17, 2)])
def test_bottleneck_stage_fold_vs_float64_and_three_pass(geom, report):
    """a ResNet stage (first block with the 1x1 downsample, then identity blocks) in the bf16 step: the folded form and the
    three-pass form against float64 autograd on the same bf16-representable weights and inputs -- outputs, input gradient and
    every parameter gradient; the fold may not be further from float64 than the form it replaces (it rounds less)"""
    N, inplanes, planes, stride, dil, H, W, nblocks = geom
    torch.manual_seed(sum(geom))
    cfg = dict(type="BN", requires_grad=True)
    down = backbone.Downsample(layers.Conv2d(inplanes, planes * 4, 1, stride=stride), backbone.make_norm(cfg, planes * 4))
    blocks = [backbone.Bottleneck(inplanes, planes, stride, dil, down, cfg)]
    blocks += [backbone.Bottleneck(planes * 4, planes, 1, dil, None, cfg) for _ in range(nblocks - 1)]
    stage = backbone.Stage(blocks).to(DEV)
    with torch.no_grad():
        for n_, p_ in stage.named_parameters():
            if p_.dim() == 4:
                p_.copy_(q(p_))
            elif n_.endswith("weight"):
                p_.copy_(torch.rand_like(p_) * 0.5 + 0.5)
            else:
                p_.copy_(torch.randn_like(p_) * 0.2)
    stage.train()
    x = q(torch.relu(torch.randn(N, inplanes, H, W)))
    # float64 reference
    params = {p_: p_.detach().double().cpu().requires_grad_(True) for p_ in stage.parameters()}
    xr = x.double().requires_grad_(True)
    t = xr
    for b in stage.children():
        rb = RefBottleneck(b)
        rb.p = params
        t = rb(t)
    dy = q(torch.randn(t.shape))
    t.backward(dy.double())
    y64, dx64 = t.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1)
    res = {}
    for fold in (False, True):
        y, dx, grads = _run_stage(stage, x, dy, fold)
        e_y, e_dx = relerr(y, y64), relerr(dx, dx64)
        e_p = {n_: relerr(grads[n_], params[p_].grad) for n_, p_ in stage.named_parameters()}
        res[fold] = (e_y, e_dx, e_p)
    worst = lambda d: max(d.items(), key=lambda kv: kv[1])
    report(f"stage {geom}: y three-pass {res[False][0]:.2e} fold {res[True][0]:.2e}; dx {res[False][1]:.2e} / {res[True][1]:.2e}; "
           f"worst parameter gradient {worst(res[False][2])} / {worst(res[True][2])}")
    # a ResNet stage in bf16 storage sits ~1e-2 from float64; the fold must not be worse than the three-pass form by more than noise
    assert res[True][0] <= 1.3 * res[False][0] + 2e-3 and res[True][1] <= 1.3 * res[False][1] + 2e-3
    for n_ in res[True][2]:
        assert res[True][2][n_] <= 1.5 * res[False][2][n_] + 5e-3, (n_, res[True][2][n_], res[False][2][n_])
    assert res[True][0] < 4e-2             # (the gradients of a bf16 stage at these tiny sizes are ~0.2 from float64 in BOTH forms)
