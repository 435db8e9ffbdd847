"""GPU parity of the hand-written flow head (csrc/flowhead.hip) against the REFERENCE's own outputs and
gradients (tests/golden/head_*.npz): piecewise-constant + free residual, affine, quadratic affine and
the outlier-robust loss."""
import os
import types

import numpy as np
import pytest
import torch

import rcf_amd
from rcf_amd import ops, synth
from rcf_amd.layers import Act

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def to_nhwc(x, cpad=None):
    return ops.nchw_to_nhwc(torch.from_numpy(np.ascontiguousarray(x)).to(DEV), cpad)


@pytest.mark.parametrize("tag", ["head_free", "head_affine", "head_affine_quad", "head_free_robust"])
def test_flow_head_vs_reference_golden(tag, golden_dir, report):
    fx = np.load(os.path.join(golden_dir, tag + ".npz"))
    B, C, h, w = (int(fx[k]) for k in ("B", "C", "h", "w"))
    head = rcf_amd.FlowAggregationHeadWithResidual(
        args=None, create_flownet=True, mask_layer=C, mask_size=(h, w), clamp_flow_t=20.,
        free_residual=not bool(fx["affine"]), free_residual_with_affine=bool(fx["affine"]),
        free_residual_with_affine_quadratic=bool(fx["quadratic"]), allow_residual_resize=True,
        outlier_robust_loss=bool(fx["robust"]))
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    head.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=11).items()})
    head = head.to(DEV)
    model = types.SimpleNamespace(w_seg=1.0, w_entropy=0.0, w_pl=0, w_crf=0, compactness_head=None, w_sharpen=0, t_sharpen=0.25,
                                  object_aware_sharpening=False,
                                  args=types.SimpleNamespace(object_channel=None))
    logits = Act(to_nhwc(fx["logits"].reshape(B * 2, C, h, w)))                 # n = b*2 + i
    res = Act(to_nhwc(np.concatenate([fx["rfw"], fx["rbw"]], axis=1)))
    gfw, gbw = torch.from_numpy(fx["gfw"]).to(DEV), torch.from_numpy(fx["gbw"]).to(DEV)
    for p in head.parameters():
        p.grad = None
    losses, seed = head.loss_and_grads(model, logits, res, gfw, gbw, {}, B, 2, want_flows=True)
    seed(1.0)
    e = {"seg": rel(float(losses["loss_warp_seg"]), float(fx["seg"]))}
    # the head's outputs themselves (reference: flows dict of models/flow_aggregation_head_with_residual.py:370-399,
    # normalised [B,4,h,w]) through the nn.Module surface, which runs the same kernels on softmax(log p) = p
    masks = torch.softmax(torch.from_numpy(fx["logits"]), dim=2).to(DEV)
    flows, lf = head(torch.zeros(B, 2, 3, 4, 4, device=DEV), masks, gfw, gbw, torch.from_numpy(fx["rfw"]).to(DEV),
                     torch.from_numpy(fx["rbw"]).to(DEV))
    for k in ("pred_flow", "agg_flow", "residual_adj", "affine_flow"):
        if "flow_" + k in fx:
            e["flow." + k] = rel(flows[k][0].cpu().numpy(), fx["flow_" + k])
    e["seg_surface"] = rel(float(lf["seg"]), float(fx["seg"]))
    e["masks"] = rel(head.last_flows["masks"].cpu().numpy().reshape(B, 2, C, h, w), masks.cpu().numpy())
    e["dlogits"] = rel(ops.nhwc_to_nchw(logits.grad, C).cpu().numpy().reshape(B, 2, C, h, w), fx["dlogits"])
    dres = ops.nhwc_to_nchw(res.grad).cpu().numpy()
    e["dres_fw"], e["dres_bw"] = rel(dres[:, :2 * C], fx["dres_fw"]), rel(dres[:, 2 * C:], fx["dres_bw"])
    for n, p in head.named_parameters():
        e["d" + n] = rel(p.grad.cpu().contiguous().numpy(), fx["dparam_" + n.replace(".", "_")])
    report(f"flow head {tag}: " + " ".join(f"{k} {v:.2e}" for k, v in e.items()))
    # the robust loss has an unbounded second derivative near |d| -> 0: its gradients are conditioned worse
    tol = 2e-3 if bool(fx["robust"]) else 2e-4
    assert e["seg"] < 1e-5
    assert max(v for k, v in e.items() if k != "seg") < tol, e


def test_flow_head_entropy_and_targets_vs_torch(report):
    """entropy (double softmax) and pl / crf target terms with an upstream gradient scale: HIP tail vs the oracle head"""
    import torch.nn.functional as F
    g = np.random.Generator(np.random.PCG64(5))
    B, C, h, w = 2, 4, 20, 28
    head = rcf_amd.FlowAggregationHeadWithResidual(args=None, create_flownet=True, mask_layer=C, mask_size=(h, w),
                                                   clamp_flow_t=20., free_residual=True, allow_residual_resize=True).to(DEV)
    lg = g.normal(0, 2, size=(B * 2, C, h, w)).astype(np.float32)
    rs = g.normal(0, 6, size=(B, 4 * C, h, w)).astype(np.float32)
    gfw = torch.from_numpy(g.normal(0, 8, size=(B, 1, 2, h, w)).astype(np.float32)).to(DEV)
    gbw = torch.from_numpy(g.normal(0, 8, size=(B, 1, 2, h, w)).astype(np.float32)).to(DEV)
    pl = torch.from_numpy(g.random((B, 2, h, w)).astype(np.float32)).to(DEV)
    crf = torch.from_numpy((g.random((B, 2, h, w)) > 0.5).astype(np.float32)).to(DEV)
    model = types.SimpleNamespace(w_seg=1.0, w_entropy=0.05, w_pl=3.0, pl_pos_weight=2.0, pl_neg_weight=1.0,
                                  pl_mask_pos_th=0.35, w_crf=10.0, crf_pos_weight=2.0, crf_neg_weight=1.0,
                                  crf_mask_pos_th=-1.0, compactness_head=None, w_sharpen=0, t_sharpen=0.25,
                                  object_aware_sharpening=False,
                                  args=types.SimpleNamespace(object_channel=2))
    logits, res = Act(to_nhwc(lg)), Act(to_nhwc(rs))
    for p in head.parameters():
        p.grad = None
    losses, seed = head.loss_and_grads(model, logits, res, gfw, gbw, {"pl_masks": pl, "crf_masks": crf}, B, 2)
    seed(0.5)                                           # an upstream scale, like loss.backward(gradient=0.5)
    # reference of the same tail: the ORACLE head (pinned to the reference, tests/test_oracle_cpu.py) on the CPU with this
    # head's weights, torch autograd
    import rcf_torch as orc
    ohead = orc.FlowAggregationHeadWithResidual(args=None, create_flownet=True, mask_layer=C, mask_size=(h, w),
                                                clamp_flow_t=20., free_residual=True, allow_residual_resize=True)
    ohead.load_state_dict({k: v.detach().cpu().contiguous() for k, v in head.state_dict().items()})
    l = torch.from_numpy(lg).requires_grad_(True)
    r = torch.from_numpy(rs).requires_grad_(True)
    p = F.softmax(l.view(B, 2, C, h, w), dim=2)
    _, lf = ohead(torch.zeros(B, 2, 3, 4, 4), p, gfw.cpu(), gbw.cpu(), r[:, :2 * C], r[:, 2 * C:])
    ent = -(p * F.log_softmax(p, dim=2)).sum(dim=2).mean()
    pl, crf = pl.cpu(), crf.cpu()

    def asym(t, pred, wp, wn):                          # models/rcf_model.py:380-408 (pinned by the stage-2 goldens)
        d = t - pred
        return (d.clamp(min=0) ** 2).mean() * wp + (d.clamp(max=0) ** 2).mean() * wn
    lpl = asym((pl > 0.35).float(), p[:, :, 2], 2.0, 1.0)
    lcrf = asym(crf, p[:, :, 2], 2.0, 1.0)
    total = lf["seg"] + 0.05 * ent + 3.0 * lpl + 10.0 * lcrf
    gl, gr = torch.autograd.grad(total * 0.5, [l, r])
    e = {"loss": rel(float(losses["loss"]), float(total)), "entropy": rel(float(losses["loss_entropy"]), float(ent)),
         "pl": rel(float(losses["loss_pl"]), float(lpl)), "crf": rel(float(losses["loss_crf"]), float(lcrf)),
         "dlogits": rel(ops.nhwc_to_nchw(logits.grad, C).cpu().numpy(), gl.cpu().numpy()),
         "dres": rel(ops.nhwc_to_nchw(res.grad).cpu().numpy(), gr.cpu().numpy())}
    report("flow head entropy/targets: " + " ".join(f"{k} {v:.2e}" for k, v in e.items()))
    assert max(e.values()) < 2e-4, e


@pytest.mark.parametrize("tag", ["head_free", "head_affine"])
def test_flow_head_zero_mass_segment_is_absent_not_nan(tag, golden_dir, report):
    """A segment whose softmax mass is exactly 0 in a frame (logits hundreds apart) makes the reference's
    `mask / mask.sum(...)` 0/0 and the step NaN (flow_aggregation_head_with_residual.py:242-243).  Deliberate deviation:
    the segment is treated as absent from that frame -- loss and gradients stay finite, and they equal what the head
    gives for the same frame with that channel removed from the softmax altogether (the other frames are untouched)."""
    fx = np.load(os.path.join(golden_dir, tag + ".npz"))
    B, C, h, w = (int(fx[k]) for k in ("B", "C", "h", "w"))

    def run(logits_np, nchan):
        head = rcf_amd.FlowAggregationHeadWithResidual(
            args=None, create_flownet=True, mask_layer=nchan, mask_size=(h, w), clamp_flow_t=20.,
            free_residual=not bool(fx["affine"]), free_residual_with_affine=bool(fx["affine"]),
            free_residual_with_affine_quadratic=bool(fx["quadratic"]), allow_residual_resize=True,
            outlier_robust_loss=bool(fx["robust"]))
        shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
        head.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=11).items()})
        head = head.to(DEV)
        model = types.SimpleNamespace(w_seg=1.0, w_entropy=0.0, w_pl=0, w_crf=0, compactness_head=None, w_sharpen=0,
                                      t_sharpen=0.25, object_aware_sharpening=False,
                                      args=types.SimpleNamespace(object_channel=None))
        logits = Act(to_nhwc(logits_np.reshape(B * 2, nchan, h, w)))
        hr, wr = fx["rfw"].shape[2:]                               # residual maps (half resolution): [B, (x|y) x C, hr, wr]
        rfw, rbw = fx["rfw"].reshape(B, 2, C, hr, wr)[:, :, :nchan], fx["rbw"].reshape(B, 2, C, hr, wr)[:, :, :nchan]
        res = Act(to_nhwc(np.concatenate([rfw.reshape(B, 2 * nchan, hr, wr), rbw.reshape(B, 2 * nchan, hr, wr)], axis=1)))
        gfw, gbw = torch.from_numpy(fx["gfw"]).to(DEV), torch.from_numpy(fx["gbw"]).to(DEV)
        losses, seed = head.loss_and_grads(model, logits, res, gfw, gbw, {}, B, 2)
        seed(1.0)
        return float(losses["loss_warp_seg"]), ops.nhwc_to_nchw(logits.grad, nchan).cpu().numpy().reshape(B, 2, nchan, h, w)

    lg = fx["logits"].copy()                                   # [B, 2, C, h, w]
    lg[:, :, C - 1] = -1.0e4                                   # the last segment: exp underflows on every pixel of every frame
    loss_dead, g_dead = run(lg, C)
    loss_ref, g_ref = run(np.ascontiguousarray(lg[:, :, :C - 1]), C - 1)
    e_loss = abs(loss_dead - loss_ref) / abs(loss_ref)
    e_g = rel(g_dead[:, :, :C - 1], g_ref)
    report(f"flow head {tag}, segment of zero mass: loss {loss_dead:.6f} vs the head without that segment {loss_ref:.6f} "
           f"({e_loss:.1e}); dlogits of the live segments {e_g:.1e}; dead segment's dlogits max {np.abs(g_dead[:, :, C - 1]).max():.1e}")
    assert np.isfinite(loss_dead) and np.isfinite(g_dead).all()
    assert e_loss < 1e-5 and e_g < 2e-4 and np.abs(g_dead[:, :, C - 1]).max() == 0.0
