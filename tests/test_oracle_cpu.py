"""CPU: the oracle restatement against the golden vectors captured from the reference itself
(tests/golden/*.npz, generator tests/golden/make_golden.py)."""
import copy
import json
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import rcf_torch as orc
import rcf_amd
from rcf_amd import config, synth


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _args():
    return types.SimpleNamespace(checkpoints_dir="/tmp/rcf_test", object_channel=None)


@pytest.mark.parametrize("tag,affine", [("rcf_small", False), ("rcf_small_affine", True)])
def test_oracle_model_vs_reference_golden(tag, affine, golden_dir):
    fx = np.load(os.path.join(golden_dir, tag + ".npz"))
    H, W, B = int(fx["H"]), int(fx["W"]), int(fx["B"])
    kw = config.stage1_model_kwargs(tuple(int(v) for v in fx["mask_size"]), dropout=0.0, affine=affine, norm="BN")
    m = orc.RCFModel(_args(), **copy.deepcopy(kw))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=int(fx["weight_seed"])).items()})
    nb = synth.make_batch(B, H, W, config_id=int(fx["config_id"]))
    batch = {k: [torch.from_numpy(np.ascontiguousarray(x)) for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
    m.train()
    losses = m(batch)
    losses["loss"].backward()
    for k in ("loss", "loss_warp_seg", "loss_entropy"):
        assert rel(losses[k].item(), float(fx[k])) < 1e-5, k
    assert rel(m.last["masks"].detach().numpy(), fx["masks"]) < 1e-5
    assert rel(m.last["res_fw"].detach().numpy(), fx["res_fw"]) < 1e-5
    assert np.array_equal(m.last["logits"].detach().argmax(1).numpy().astype(np.uint8)[fx["margin"] > 1e-4],
                          fx["argmax"][fx["margin"] > 1e-4])
    named = dict(m.named_parameters())
    for i, n in enumerate(fx["sampled"]):
        assert rel(named[str(n)].grad.numpy().ravel()[:256], fx[f"grad_{i}"]) < 1e-4, n


@pytest.mark.parametrize("tag", ["head_free", "head_affine", "head_affine_quad", "head_free_robust"])
def test_oracle_flow_head_vs_reference_golden(tag, golden_dir):
    fx = np.load(os.path.join(golden_dir, tag + ".npz"))
    B, C, h, w = (int(fx[k]) for k in ("B", "C", "h", "w"))
    head = orc.FlowAggregationHeadWithResidual(
        args=None, create_flownet=True, mask_layer=C, mask_size=(h, w), clamp_flow_t=20.,
        free_residual=not bool(fx["affine"]), free_residual_with_affine=bool(fx["affine"]),
        free_residual_with_affine_quadratic=bool(fx["quadratic"]), allow_residual_resize=True,
        outlier_robust_loss=bool(fx["robust"]))
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    head.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=11).items()})
    lg = torch.from_numpy(fx["logits"]).requires_grad_(True)
    a, b = torch.from_numpy(fx["rfw"]).requires_grad_(True), torch.from_numpy(fx["rbw"]).requires_grad_(True)
    flows, loss = head(torch.zeros(B, 2, 3, 4, 4), F.softmax(lg, dim=2), torch.from_numpy(fx["gfw"]),
                       torch.from_numpy(fx["gbw"]), a, b)
    loss["seg"].backward()
    tol = 5e-4 if bool(fx["robust"]) else 5e-5
    assert rel(loss["seg"].item(), float(fx["seg"])) < 1e-5
    assert rel(lg.grad.numpy(), fx["dlogits"]) < tol and rel(a.grad.numpy(), fx["dres_fw"]) < tol
    for k in ("pred_flow", "agg_flow", "residual_adj", "affine_flow"):
        if "flow_" + k in fx:
            assert rel(flows[k][0].detach().numpy(), fx["flow_" + k]) < 5e-5, k


def test_oracle_warp_family_vs_reference_golden(golden_dir):
    fx = np.load(os.path.join(golden_dir, "warp.npz"))
    x, y = torch.from_numpy(fx["x"]), torch.from_numpy(fx["y"])
    for name in ("random", "integer", "outofrange"):
        f12, f21 = torch.from_numpy(fx[f"{name}_f12"]), torch.from_numpy(fx[f"{name}_f21"])
        assert rel(orc.flow_warp(x, f12, "border").numpy(), fx[f"{name}_warp_border"]) < 1e-6
        assert rel(orc.flow_warp(x, f12, "zeros").numpy(), fx[f"{name}_warp_zeros"]) < 1e-6
        ob = orc.occu_mask_backward(f21)
        assert np.array_equal(ob.numpy().astype(np.uint8), fx[f"{name}_occ_back"])
        assert np.array_equal(orc.occu_mask_bidirection(f12, f21).numpy().astype(np.uint8), fx[f"{name}_occ_bidir"])
        ph = orc.photometric_loss(y, torch.from_numpy(fx[f"{name}_warp_border"]), 1 - ob)
        assert rel(ph.item(), float(fx[f"{name}_photo"])) < 1e-6


def test_oracle_crf_head_prefix_lr_table_ema(golden_dir):
    fx = np.load(os.path.join(golden_dir, "crf_pre.npz"))
    H, W, seed = int(fx["H"]), int(fx["W"]), int(fx["seed"])
    head = orc.CRFHead(None, crf_soft=None)
    img = torch.from_numpy(synth.normalize_rgb(synth.smooth_rgb(H, W, seed)))[None]
    assert np.array_equal(head.to_uint8_image(img)[0].numpy(), fx["img_u8"])
    q, UU = head.unary(torch.from_numpy(synth.soft_blob_mask(H, W, seed)))
    assert np.array_equal(q.numpy(), fx["mask_q"]) and rel(UU.numpy(), fx["unary"]) < 1e-7
    tab = json.load(open(os.path.join(golden_dir, "lr_table.json")))
    for e, ref in enumerate(tab["factor"]):
        assert abs(orc.poly_lr_factor(e, tab["epochs"], tab["power"], tab["base_lr"], tab["min_lr"]) - ref) < 1e-15
        assert abs(rcf_amd.poly_lr_factor(e, tab["epochs"], tab["power"], tab["base_lr"], tab["min_lr"]) - ref) < 1e-15
    ema = np.load(os.path.join(golden_dir, "ema.npz"))
    src, dst = torch.nn.BatchNorm2d(4), torch.nn.BatchNorm2d(4)
    with torch.no_grad():
        src.weight.copy_(torch.tensor([1., 2., 3., 4.]))
        src.running_mean.copy_(torch.tensor([.1, .2, .3, .4]))
        src.num_batches_tracked.fill_(1000)
        dst.num_batches_tracked.fill_(3)
    orc.momentum_update_param_and_buffer(src, dst, 0.999)
    for k, v in dst.state_dict().items():
        assert np.array_equal(v.numpy(), ema[k.replace(".", "_")]), k    # incl. the int64 truncation quirk


@pytest.mark.parametrize("variant", list(config.STAGE2_VARIANTS))
def test_oracle_stage2_vs_reference_golden(variant, golden_dir):
    """stage 2.1 (EMA teacher -> CRF self-labels) and 2.2 (pseudo labels) against the numbers the REFERENCE produced
    (tests/golden/make_golden_stage2.py; torchcrf_cpp.crf_soft bound to oracle/crf_ref.c on both sides): losses,
    the FFI operands, the CRF / pl targets fed to the loss, gradient norms, the EMA copies after the update.
    Note the teacher runs in TRAINING mode after model.train(), as in the reference (no train() override there)."""
    import crf_oracle
    fx = json.load(open(os.path.join(golden_dir, "stage2.json")))[variant]
    arr = np.load(os.path.join(golden_dir, "stage2.npz"))
    H, W, B = fx["H"], fx["W"], fx["B"]
    kw, oc = config.variant_model_kwargs(variant, H, W)
    assert oc == fx["object_channel"]
    calls = []

    def crf_soft(img, UU, *rest):
        m = crf_oracle.crf_soft_torch(img, UU, *rest)
        calls.append((img.numpy().copy(), UU.numpy().copy(), m.numpy().copy()))
        return m
    if "crf_head" in kw:
        kw["crf_head"]["crf_soft"] = crf_soft
    args = _args()
    args.object_channel = oc
    m = orc.RCFModel(args, **copy.deepcopy(kw))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=fx["weight_seed"]).items()})
    nb = synth.make_batch(B, H, W, config_id=fx["config_id"])
    batch = {k: [torch.from_numpy(np.ascontiguousarray(x)) for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
    batch["pl_masks"] = [torch.from_numpy(a) for a in synth.make_pl_masks(B, H, W, config_id=fx["config_id"])]
    m.train()
    assert m.backbone2_ema.training and m.decode_head2_ema.training
    losses = m(batch)
    losses["loss"].backward()
    assert sorted(k for k in losses if "loss" in k) == sorted(fx["loss"])
    for k, v in fx["loss"].items():
        assert rel(losses[k].item(), v) < 1e-5, k
    if fx["crf_calls"]:
        assert len(calls) == fx["crf_calls"]
        assert np.array_equal(np.stack([c[0] for c in calls]), arr[variant + "_crf_img_u8"])
        assert rel(np.stack([c[1] for c in calls]), arr[variant + "_crf_unary"]) < 1e-6
        assert np.array_equal(np.stack([c[2] for c in calls]).astype(np.uint8), arr[variant + "_crf_map"])
        assert rel(losses["_crf_masks"].numpy(), arr[variant + "_crf_target"]) < 1e-6
    gn = {}
    for n, p in m.named_parameters():
        if p.grad is not None:
            gn[n.split(".")[0]] = gn.get(n.split(".")[0], 0.0) + float(p.grad.double().pow(2).sum())
    for k, v in fx["gradnorm"].items():
        assert rel(np.sqrt(gn[k]), v) < 1e-4, k
    sd = m.state_dict()
    for k in fx["ema_keys"]:
        assert np.array_equal(sd[k].numpy(), arr[variant + "_ema_" + k.replace(".", "_")]), k
