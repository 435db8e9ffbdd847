"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference).  The reference's third-party
imports that are absent here (mmcv, mmseg, torchvision, flow_vis, pytorch_lightning,
torchcrf_cpp, pydensecrf) are replaced by minimal stand-in modules whose arithmetic is
torch's own: resize == F.interpolate, build_conv_layer == nn.Conv2d,
build_norm_layer == BatchNorm2d(eps 1e-5), ConvModule == conv -> norm -> ReLU with
kaiming-normal(fan_out) init.  Nothing from the reference is copied into the repo: only
inputs (regenerable from seeds) and OUTPUT tensors are stored.

Each fixture is also checked against the oracle restatement (oracle/rcf_torch.py) right
here, so a stale oracle cannot silently ship:  python tests/golden/make_golden.py
"""
import argparse
import copy
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


# ----------------------------------------------------------------------------- stand-ins
def install_standins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def resize(input, size=None, scale_factor=None, mode="nearest", align_corners=None, warning=True):
        if size is not None:
            size = tuple(int(s) for s in size)
        return F.interpolate(input, size, scale_factor, mode, align_corners)

    def build_conv_layer(cfg, *a, **k):
        assert cfg is None
        return nn.Conv2d(*a, **k)

    def build_norm_layer(cfg, num_features, postfix=""):
        assert cfg["type"] in ("BN", "SyncBN")
        bn = nn.BatchNorm2d(num_features, eps=1e-5)
        for p in bn.parameters():
            p.requires_grad = cfg.get("requires_grad", True)
        return "bn" + str(postfix), bn

    def kaiming_init(m, a=0, mode="fan_out", nonlinearity="relu", bias=0, distribution="normal"):
        nn.init.kaiming_normal_(m.weight, a=a, mode=mode, nonlinearity=nonlinearity)
        if getattr(m, "bias", None) is not None:
            nn.init.constant_(m.bias, bias)

    def constant_init(m, val, bias=0):
        nn.init.constant_(m.weight, val)
        if getattr(m, "bias", None) is not None:
            nn.init.constant_(m.bias, bias)

    def normal_init(m, mean=0, std=1, bias=0):
        nn.init.normal_(m.weight, mean, std)
        if getattr(m, "bias", None) is not None:
            nn.init.constant_(m.bias, bias)

    class ConvModule(nn.Module):
        def __init__(self, cin, cout, kernel_size, stride=1, padding=0, dilation=1, conv_cfg=None,
                     norm_cfg=None, act_cfg=dict(type="ReLU")):
            super().__init__()
            self.conv = nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=padding,
                                  dilation=dilation, bias=norm_cfg is None)
            self.with_norm = norm_cfg is not None
            if self.with_norm:
                self.bn = build_norm_layer(norm_cfg, cout)[1]
            self.activate = nn.ReLU(inplace=True)
            kaiming_init(self.conv)

        def forward(self, x):
            x = self.conv(x)
            if self.with_norm:
                x = self.bn(x)
            return self.activate(x)

    ident_deco = lambda *a, **k: (lambda f: f)
    mod("mmcv")
    mod("mmcv.cnn", build_conv_layer=build_conv_layer, build_norm_layer=build_norm_layer,
        build_plugin_layer=None, constant_init=constant_init, kaiming_init=kaiming_init,
        normal_init=normal_init, ConvModule=ConvModule)
    mod("mmcv.runner", load_checkpoint=None, auto_fp16=ident_deco, force_fp32=ident_deco)
    mod("mmcv.utils")
    mod("mmcv.utils.parrots_wrapper", _BatchNorm=nn.modules.batchnorm._BatchNorm)
    mod("mmseg")
    mod("mmseg.ops", resize=resize)
    mod("mmseg.core", build_pixel_sampler=None)
    mod("torchvision")
    mod("flow_vis")
    mod("pytorch_lightning")
    mod("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
    mod("torchcrf_cpp", crf_soft=None, crf_hard=None)
    mod("pydensecrf")
    mod("pydensecrf.densecrf")
    # the reference pins tensors to "cuda" by literal in three places; run them on CPU
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    _tensor, _zeros = torch.tensor, torch.zeros
    torch.tensor = lambda *a, device=None, **k: _tensor(*a, **k)
    torch.zeros = lambda *a, device=None, **k: _zeros(*a, **k)


def stage1_model_kwargs(mask_size, affine=False, dropout=0.0, mask_layer=4):
    """configs/rcf/rcf_stage1.yaml:63-148 with SyncBN->BN, dropout 0 and the given mask size."""
    norm = dict(type="BN", requires_grad=True)
    return dict(
        w_seg=1.0, w_sharpen=0, w_entropy=0.05, separate_residual=True, mask_layer=mask_layer,
        align_corners=False, mask_size=list(mask_size), log_interval=10 ** 9, train_iter=1,
        backbone2=dict(type="ResNet", depth=50, num_stages=4, out_indices=[0, 1, 2, 3],
                       dilations=[1, 1, 2, 4], strides=[1, 2, 1, 1], norm_cfg=norm, norm_eval=False,
                       style="pytorch", contract_dilation=True),
        decode_head=dict(type="FlowAggregationHeadWithResidual", ssim_sz=1, create_flownet=True,
                         mask_layer=mask_layer, flow_feat_before_agg_kernel_size=3,
                         num_flow_feat_channels=64, mask_size=list(mask_size), norm_flow=False,
                         clamp_flow_t=20., free_residual=not affine, free_residual_with_affine=affine,
                         free_scale=False, outlier_robust_loss=False, eps=0.01, q=0.4,
                         allow_residual_resize=True, residual_adjustment_scale=10., pred_div_coeff=10.),
        decode_head2=dict(type="FCNHead", input_transform="resize_concat", concat_input=False, dilation=6,
                          channels=256, in_channels=[256, 2048], in_index=[0, 3], num_convs=2,
                          dropout_ratio=dropout, num_classes=mask_layer, norm_cfg=norm, align_corners=False,
                          loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0)),
        decode_head3=dict(type="FCNHead", concat_input=False, dilation=6, channels=256, in_channels=4096,
                          in_index=-1, num_convs=2, dropout_ratio=dropout, num_classes=4 * mask_layer,
                          norm_cfg=norm, align_corners=False,
                          loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0)))


def torch_batch(np_batch):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    return {"imgs": [t(a) for a in np_batch["imgs"]], "gt_fw_flows": [t(a) for a in np_batch["gt_fw_flows"]],
            "gt_bw_flows": [t(a) for a in np_batch["gt_bw_flows"]], "seq_ids": t(np_batch["seq_ids"]),
            "seq_names": np_batch["seq_names"], "paths": np_batch["paths"]}


def grad_norms(model):
    out = {}
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        top = name.split(".")[0]
        out[top] = out.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    return {k: float(np.sqrt(v)) for k, v in out.items()}


SAMPLED = ["backbone2.layer2.1.conv2.weight", "decode_head2.convs.0.conv.weight",
           "decode_head.flow_feat_after_agg.2.weight", "backbone2.bn1.bias", "decode_head3.conv_seg.bias"]


def run_step(model, batch, lr=1e-4, wd=1e-4):
    model.train()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=lr, weight_decay=wd)
    losses = model(batch)
    opt.zero_grad()
    losses["loss"].backward()
    gn = grad_norms(model)
    grads = {n: model.get_parameter(n).grad.detach().clone().numpy().ravel()[:256] for n in SAMPLED}
    opt.step()
    after = {n: model.get_parameter(n).detach().clone().numpy().ravel()[:256] for n in SAMPLED}
    return losses, gn, grads, after


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-large", action="store_true")
    opts = ap.parse_args()
    install_standins()
    sys.path.insert(0, REF)
    import models as ref_models                      # noqa: the reference itself
    import utils as ref_utils                        # noqa
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rcf_torch as orc
    import rcf_amd                                   # noqa: registers the package alias
    from rcf_amd import synth
    torch.manual_seed(0)
    torch.set_num_threads(8)
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_golden", object_channel=None,
                                 eval_save=False, eval_export=False)
    report = {}

    # ---------------------------------------------------------------- full model, small
    def model_case(tag, H, W, B, affine, store_full):
        mask = ((H + 3) // 4 if H % 4 else H // 4, None)
        mh = (H // 2 + 1) // 2 if True else None
        # conv1 s2 p3 k7 -> floor((H-1)/2)+1 ; maxpool s2 p1 k3 -> floor((h-1)/2)+1
        h1 = (H - 1) // 2 + 1
        w1 = (W - 1) // 2 + 1
        mask_size = ((h1 - 1) // 2 + 1, (w1 - 1) // 2 + 1)
        kw = stage1_model_kwargs(mask_size, affine=affine)
        ref = ref_models.RCFModel(args, **copy.deepcopy(kw))
        ora = orc.RCFModel(args, **copy.deepcopy(kw))
        shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        assert list(shapes) == list(ora.state_dict().keys()), "state-dict schema differs"
        sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()}
        ref.load_state_dict(sd)
        ora.load_state_dict(sd)
        nb = synth.make_batch(B, H, W, config_id=1)
        l_ref, gn_ref, g_ref, a_ref = run_step(ref, torch_batch(nb))
        # second forward (after one Adam step) in train mode pins the update end-to-end
        ref.train_iter = 1
        l2_ref = ref(torch_batch(nb))
        # capture reference intermediates through a fresh forward with the ORIGINAL weights
        ref0 = ref_models.RCFModel(args, **copy.deepcopy(kw))
        ref0.load_state_dict(sd)
        ref0.train()
        imgs = torch.stack(torch_batch(nb)["imgs"], dim=1)
        with torch.no_grad():
            feats = ref0.backbone2(imgs.view(2 * B, 3, H, W))
            logits = ref0.decode_head2(feats)
            rfw, rbw = ref0.pred_separate_residual(feats, B, 2)
        masks = F.softmax(logits.view(B, 2, 4, *mask_size), dim=2)
        l_o, gn_o, g_o, a_o = run_step(ora, torch_batch(nb))
        l2_o = ora(torch_batch(nb))
        chk = {"loss": rel(l_o["loss"].item(), l_ref["loss"].item()),
               "loss_warp_seg": rel(l_o["loss_warp_seg"].item(), l_ref["loss_warp_seg"].item()),
               "loss_entropy": rel(l_o["loss_entropy"].item(), l_ref["loss_entropy"].item()),
               "loss_after_step": rel(l2_o["loss"].item(), l2_ref["loss"].item())}
        for k in gn_ref:
            chk["gradnorm." + k] = rel(gn_o[k], gn_ref[k])
        for k in SAMPLED:
            chk["grad." + k] = rel(g_o[k], g_ref[k])
            chk["adam." + k] = rel(a_o[k], a_ref[k])
        report[tag] = chk
        print(tag, json.dumps(chk, indent=1))
        assert max(chk.values()) < 2e-4, f"oracle disagrees with the reference on {tag}"
        # float64 ground truth (the oracle, pinned to the reference above, run in double) and the
        # reference's OWN fp32 deviation from it: the yardstick for every other fp32 implementation
        o64 = orc.RCFModel(args, **copy.deepcopy(kw))
        o64.load_state_dict(sd)
        o64 = o64.double().train()
        b64 = torch_batch(nb)
        b64 = {k: ([t.double() for t in v] if k in ("imgs", "gt_fw_flows", "gt_bw_flows") else v) for k, v in b64.items()}
        l64 = o64(b64)
        l64["loss"].backward()
        gn64 = grad_norms(o64)
        g64 = {n: o64.get_parameter(n).grad.detach().numpy().ravel()[:256].copy() for n in SAMPLED}
        logits64 = o64.last["logits"].detach()
        masks64 = F.softmax(logits64.view(B, 2, 4, *mask_size), dim=2)
        # fp32 is not one number: the SAME reference model evaluated with a different (equally valid) reduction
        # order -- one thread, or channels_last convolutions -- moves its results by more than its default run is
        # away from the float64 truth.  The yardstick stored as ref32_err_* is the worst of these reference runs.
        ref32 = dict(loss=[rel(l_ref["loss"].item(), l64["loss"].item())], logits=[rel(logits.numpy(), logits64.numpy())],
                     masks=[float((masks.double() - masks64).abs().max())],
                     gradnorm=[[rel(gn_ref[k], gn64[k]) for k in sorted(gn_ref)]], grad=[[rel(g_ref[k], g64[k]) for k in SAMPLED]])
        # ... and so does rounding the INPUTS differently: the reference evaluated at parameters moved by one unit in
        # the last place (each weight times 1 +- 2^-23, random signs) is what any backward-stable fp32
        # implementation is allowed to return -- the conditioning of each compared quantity, measured
        for nthreads, cl, ulp_seed in ((1, False, 0), (8, True, 0), (8, False, 1), (8, False, 2), (8, False, 3), (8, False, 4)):
            torch.set_num_threads(nthreads)
            rv = ref_models.RCFModel(args, **copy.deepcopy(kw))
            if ulp_seed:
                gp = torch.Generator().manual_seed(1000 + ulp_seed)
                sdp = {k: (v * (1 + (torch.randint(0, 2, v.shape, generator=gp).float() * 2 - 1) * 2.0 ** -23)
                           if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")) else v)
                       for k, v in sd.items()}
                rv.load_state_dict(sdp)
            else:
                rv.load_state_dict(sd)
            if cl:
                rv = rv.to(memory_format=torch.channels_last)
            rv.train()
            with torch.no_grad():
                sv = copy.deepcopy(rv.state_dict())
                lg = rv.decode_head2(rv.backbone2(imgs.view(2 * B, 3, H, W)))
                rv.load_state_dict(sv)
            lv = rv(torch_batch(nb))
            lv["loss"].backward()
            gnv = grad_norms(rv)
            ref32["loss"].append(rel(lv["loss"].item(), l64["loss"].item()))
            ref32["logits"].append(rel(lg.numpy(), logits64.numpy()))
            ref32["masks"].append(float((F.softmax(lg.view(B, 2, 4, *mask_size), dim=2).double() - masks64).abs().max()))
            ref32["gradnorm"].append([rel(gnv[k], gn64[k]) for k in sorted(gn_ref)])
            ref32["grad"].append([rel(rv.get_parameter(n).grad.detach().numpy().ravel()[:256], g64[n]) for n in SAMPLED])
            torch.set_num_threads(8)
        print(tag, "reference fp32 variants vs float64:", json.dumps(ref32))
        top2 = torch.topk(logits, 2, dim=1).values
        fx = dict(H=H, W=W, B=B, affine=int(affine), weight_seed=7, config_id=1,
                  mask_size=np.array(mask_size),
                  loss=np.float64(l_ref["loss"].item()), loss_warp_seg=np.float64(l_ref["loss_warp_seg"].item()),
                  loss_entropy=np.float64(l_ref["loss_entropy"].item()),
                  loss_after_step=np.float64(l2_ref["loss"].item()),
                  argmax=logits.argmax(1).numpy().astype(np.uint8),
                  margin=(top2[:, 0] - top2[:, 1]).numpy().astype(np.float16),
                  feat_absmean=np.array([float(f.abs().mean()) for f in feats]),
                  gradnorm_keys=np.array(sorted(gn_ref)), gradnorm=np.array([gn_ref[k] for k in sorted(gn_ref)]),
                  sampled=np.array(SAMPLED),
                  truth_loss=np.float64(l64["loss"].item()),
                  truth_gradnorm=np.array([gn64[k] for k in sorted(gn_ref)]),
                  ref32_err_loss=max(ref32["loss"]), ref32_err_logits=max(ref32["logits"]),
                  ref32_err_masks=max(ref32["masks"]),
                  ref32_err_gradnorm=np.array(ref32["gradnorm"]).max(axis=0),
                  ref32_err_grad=np.array(ref32["grad"]).max(axis=0),
                  logit_absmax=np.float64(logits.abs().max().item()),
                  **{"grad_%d" % i: g_ref[k] for i, k in enumerate(SAMPLED)},
                  **{"truth_grad_%d" % i: g64[k] for i, k in enumerate(SAMPLED)},
                  **{"adam_%d" % i: a_ref[k] for i, k in enumerate(SAMPLED)})
        print(tag, "reference fp32 vs float64 truth: loss %.2e logits %.2e masks %.2e gradnorm %s grad %s" % (
            fx["ref32_err_loss"], fx["ref32_err_logits"], fx["ref32_err_masks"], fx["ref32_err_gradnorm"], fx["ref32_err_grad"]))
        if store_full:
            fx.update(masks=masks.numpy(), logits=logits.numpy(), res_fw=rfw.numpy(), res_bw=rbw.numpy(),
                      truth_logits=logits64.numpy(), truth_masks=masks64.numpy())
        else:
            fx.update(masks0=masks[0].numpy(), mask_mean=masks.mean(dim=(3, 4)).numpy(),
                      truth_logits0=logits64[:2].numpy(), truth_masks0=masks64[0].numpy())
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **fx)

    model_case("rcf_small", 96, 160, 2, affine=False, store_full=True)
    model_case("rcf_small_affine", 64, 96, 2, affine=True, store_full=True)

    # ---------------------------------------------------------------- flow head alone
    def head_case(tag, B, C, h, w, affine, quadratic=False, robust=False):
        g = np.random.Generator(np.random.PCG64(4242 + C + h))
        kw = dict(args=args, create_flownet=True, mask_layer=C, mask_size=(h, w), clamp_flow_t=20.,
                  free_residual=not affine, free_residual_with_affine=affine,
                  free_residual_with_affine_quadratic=quadratic, allow_residual_resize=True,
                  outlier_robust_loss=robust)
        ref = ref_models.rcf_model.FlowAggregationHeadWithResidual(**kw)
        ora = orc.FlowAggregationHeadWithResidual(**kw)
        shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=11).items()}
        ref.load_state_dict(sd)
        ora.load_state_dict(sd)
        logits = torch.from_numpy(g.normal(0, 2.0, size=(B, 2, C, h, w)).astype(np.float32))
        gfw = torch.from_numpy(g.normal(0, 9.0, size=(B, 1, 2, h, w)).astype(np.float32))
        gbw = torch.from_numpy(g.normal(0, 9.0, size=(B, 1, 2, h, w)).astype(np.float32))
        h2, w2 = (h + 1) // 2, (w + 1) // 2
        rfw = torch.from_numpy(g.normal(0, 8.0, size=(B, 2 * C, h2, w2)).astype(np.float32))
        rbw = torch.from_numpy(g.normal(0, 8.0, size=(B, 2 * C, h2, w2)).astype(np.float32))
        imgs = torch.zeros(B, 2, 3, 4 * h, 4 * w)
        outs = []
        for m in (ref, ora):
            lg, a, b = (t.clone().requires_grad_(True) for t in (logits, rfw, rbw))
            flows, loss = m(imgs, F.softmax(lg, dim=2), gfw.clone(), gbw.clone(), a, b)
            for p in m.parameters():
                p.grad = None
            loss["seg"].backward()
            outs.append((flows, loss, lg.grad, a.grad, b.grad, {n: p.grad.clone() for n, p in m.named_parameters()}))
        (fr, lr_, dlr, dar, dbr, pr), (fo, lo, dlo, dao, dbo, po) = outs
        chk = {"seg": rel(lo["seg"].item(), lr_["seg"].item()), "dlogits": rel(dlo, dlr), "dres_fw": rel(dao, dar),
               "dres_bw": rel(dbo, dbr)}
        for k in ("pred_flow", "agg_flow", "residual_adj", "affine_flow"):
            if fr[k]:
                chk[k] = rel(fo[k][0].detach(), fr[k][0].detach())
        for n in pr:
            chk["dparam." + n] = rel(po[n], pr[n])
        report[tag] = chk
        print(tag, json.dumps(chk, indent=1))
        assert max(chk.values()) < 5e-4, f"oracle flow head disagrees with the reference on {tag}"
        fx = dict(B=B, C=C, h=h, w=w, affine=int(affine), quadratic=int(quadratic), robust=int(robust),
                  logits=logits.numpy(), gfw=gfw.numpy(), gbw=gbw.numpy(), rfw=rfw.numpy(), rbw=rbw.numpy(),
                  seg=np.float64(lr_["seg"].item()), seg_fw=np.float64(lr_["seg_fw"].item()),
                  seg_bw=np.float64(lr_["seg_bw"].item()), dlogits=dlr.numpy(), dres_fw=dar.numpy(),
                  dres_bw=dbr.numpy(), **{"flow_" + k: fr[k][0].detach().numpy() for k in fr if fr[k]},
                  **{"dparam_" + n.replace(".", "_"): pr[n].numpy() for n in pr})
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **fx)

    head_case("head_free", 2, 4, 24, 40, affine=False)
    head_case("head_affine", 2, 3, 24, 24, affine=True)
    head_case("head_affine_quad", 1, 3, 16, 20, affine=True, quadratic=True)
    head_case("head_free_robust", 1, 4, 12, 16, affine=False, robust=True)

    # ---------------------------------------------------------------- warp / occlusion / photometric
    g = np.random.Generator(np.random.PCG64(99))
    B, H, W = 2, 64, 96
    x = torch.from_numpy(g.uniform(0, 1, size=(B, 3, H, W)).astype(np.float32))
    y = torch.from_numpy(g.uniform(0, 1, size=(B, 3, H, W)).astype(np.float32))
    flows = {"random": g.normal(0, 4.0, size=(B, 2, H, W)), "integer": g.integers(-5, 6, size=(B, 2, H, W)),
             "outofrange": g.normal(0, 60.0, size=(B, 2, H, W))}
    fx = dict(x=x.numpy(), y=y.numpy())
    from models.amd.flow_loss import unFlowLoss
    cfgobj = types.SimpleNamespace(ssim_sz=1, w_l1=0.15, w_ssim=0.85, w_ternary=0.0)
    ufl = unFlowLoss(cfgobj)
    chk = {}
    for name, fl in flows.items():
        f12 = torch.from_numpy(fl.astype(np.float32))
        f21 = torch.from_numpy((-fl[::-1].copy() * 0.7).astype(np.float32))
        wb = ref_utils.flow_warp(x, f12, pad="border")
        wz = ref_utils.flow_warp(x, f12, pad="zeros")
        ob = ref_utils.get_occu_mask_backward(f21, th=0.2)
        obi = ref_utils.get_occu_mask_bidirection(f12, f21)
        ph = ufl.loss_photomatric(y, wb, 1 - ob)
        fx.update({f"{name}_f12": f12.numpy(), f"{name}_f21": f21.numpy(), f"{name}_warp_border": wb.numpy(),
                   f"{name}_warp_zeros": wz.numpy(), f"{name}_occ_back": ob.numpy().astype(np.uint8),
                   f"{name}_occ_bidir": obi.numpy().astype(np.uint8), f"{name}_photo": np.float64(ph.item())})
        chk[name + ".border"] = rel(orc.flow_warp(x, f12, "border"), wb)
        chk[name + ".zeros"] = rel(orc.flow_warp(x, f12, "zeros"), wz)
        chk[name + ".occ_back"] = float((orc.occu_mask_backward(f21) != ob).float().mean())
        chk[name + ".occ_bidir"] = float((orc.occu_mask_bidirection(f12, f21) != obi).float().mean())
        chk[name + ".photo"] = rel(orc.photometric_loss(y, wb, 1 - ob).item(), ph.item())
    report["warp"] = chk
    print("warp", json.dumps(chk, indent=1))
    assert max(chk.values()) < 1e-5
    np.savez_compressed(os.path.join(HERE, "warp.npz"), **fx)

    # ---------------------------------------------------------------- CRF head, pre-FFI products
    captured = {}

    def fake_crf_soft(img, UU, W_, H_, *rest):
        captured["img"], captured["UU"], captured["rest"] = img.clone(), UU.clone(), rest
        return torch.zeros(H_, W_, dtype=torch.int16)

    sys.modules["torchcrf_cpp"].crf_soft = fake_crf_soft
    from models.crf_head import CRFHead as RefCRFHead
    Hc, Wc = 64, 96
    rgb = synth.smooth_rgb(Hc, Wc, 4001)
    img_n = torch.from_numpy(synth.normalize_rgb(rgb))[None]
    msk = torch.from_numpy(synth.soft_blob_mask(Hc, Wc, 4001))[None]
    RefCRFHead(args)(img_n, msk)
    o = orc.CRFHead(args, crf_soft=lambda *a: torch.zeros(a[3], a[2], dtype=torch.int16))
    q, UU = o.unary(msk[0])
    chk = {"img_u8": float((o.to_uint8_image(img_n)[0] != captured["img"]).float().mean()),
           "unary": rel(UU, captured["UU"])}
    report["crf_pre"] = chk
    print("crf_pre", chk, "u8 roundtrip mismatches vs source rgb:",
          int((captured["img"].numpy() != rgb).sum()))
    assert max(chk.values()) < 1e-6
    np.savez_compressed(os.path.join(HERE, "crf_pre.npz"), H=Hc, W=Wc, seed=4001, img_u8=captured["img"].numpy(),
                        mask_q=q.numpy(), unary=captured["UU"].numpy(),
                        params=np.array([float(v) for v in captured["rest"]]))

    # ---------------------------------------------------------------- EMA + LR table
    # main.py itself needs lightning/wandb/dataset imports; run just its get_lr (main.py:294-297)
    import ast
    tree = ast.parse(open(os.path.join(REF, "main.py")).read())
    fn = [f for c in tree.body if isinstance(c, ast.ClassDef) and c.name == "Model"
          for f in c.body if isinstance(f, ast.FunctionDef) and f.name == "get_lr"][0]
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "main.py", "exec"), ns)
    holder = types.SimpleNamespace(args=types.SimpleNamespace(epochs=200))
    table = [ns["get_lr"](holder, e, power=0.9, base_lr=1e-4, min_lr=1e-6) for e in range(201)]
    mine = [orc.poly_lr_factor(e, 200, 0.9, 1e-4, 1e-6) for e in range(201)]
    assert max(abs(a - b) for a, b in zip(table, mine)) < 1e-15
    json.dump({"epochs": 200, "power": 0.9, "base_lr": 1e-4, "min_lr": 1e-6, "factor": table},
              open(os.path.join(HERE, "lr_table.json"), "w"))
    src, dst = nn.BatchNorm2d(4), nn.BatchNorm2d(4)
    with torch.no_grad():
        src.weight.copy_(torch.tensor([1., 2., 3., 4.])); src.running_mean.copy_(torch.tensor([.1, .2, .3, .4]))
        src.num_batches_tracked.fill_(1000); dst.num_batches_tracked.fill_(3)
    dst2 = copy.deepcopy(dst)
    ref_utils.momentum_update_param_and_buffer(src, dst, 0.999)
    orc.momentum_update_param_and_buffer(src, dst2, 0.999)
    for k in dst.state_dict():
        assert torch.equal(dst.state_dict()[k], dst2.state_dict()[k]), k
    np.savez_compressed(os.path.join(HERE, "ema.npz"), **{k.replace(".", "_"): v.numpy() for k, v in dst.state_dict().items()})

    # ---------------------------------------------------------------- 480x854, B=1 (config-1 geometry)
    if not opts.skip_large:
        model_case("rcf_480x854_b1", 480, 854, 1, affine=False, store_full=False)

    json.dump(report, open(os.path.join(HERE, "oracle_vs_reference.json"), "w"), indent=1)
    print("all fixtures written")


if __name__ == "__main__":
    main()
