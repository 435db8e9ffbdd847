"""Flow-aggregation head (`decode_head`) and the loss tail of RCFModel.forward_train.

Reference: models/flow_aggregation_head_with_residual.py:50-148 (ctor), :150-162 (clamp), :164-233
(per-segment affine least squares), :235-310 (aggregate), :312-399 (forward); losses
models/rcf_model.py:350-408,433-434,464-523; compactness models/compactness_head.py:14-57.

Same constructor keywords / state-dict keys as the reference head.  The training path
(`loss_and_grads`) runs on the hand-written kernels of csrc/flowhead.hip (softmax, segment pooling,
MLP, fp64 affine least squares, reconstruction + loss, analytic backward) plus the implicit-GEMM conv
kernels for the two 3x3 flow-feature convs; the sharpen and compactness losses are part of the same
kernels.  The stand-alone `forward()` surface of the reference head (already-softmaxed NCHW masks in,
visualisation flows out) runs on the same kernels; there is no second, plain-torch implementation.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F          # softmax of the evaluation masks only

from . import _lib, ops
from ._lib import FlowHeadCfg
from .layers import Conv2d, _param_grad
from .ops import _p, _stream


class CompactnessHead(nn.Module):
    def __init__(self, args, compact_channel):
        super().__init__()
        self.args, self.compact_channel = args, compact_channel

    def get_compactness_loss(self, all_pred_mask):
        p = all_pred_mask.flatten(0, 1)
        ch = self.compact_channel
        if ch == -1:
            if self.args.object_channel is None:
                return None
            ch = self.args.object_channel
        m = p[:, ch]
        H, W = m.shape[-2:]
        cnt = m.sum(dim=(1, 2), keepdim=True)
        y = torch.arange(H, dtype=torch.float, device=m.device)[None, :, None] / H
        x = torch.arange(W, dtype=torch.float, device=m.device)[None, None, :] / W
        yc = (y * m).sum(dim=(1, 2), keepdim=True) / cnt
        xc = (x * m).sum(dim=(1, 2), keepdim=True) / cnt
        return (((y - yc) ** 2 + (x - xc) ** 2) * m).mean()


class _Pointwise(nn.Module):
    """Parameter holder with nn.Conv1d(k=1)'s state-dict shape [out, in, 1] and default init."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 1))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        nn.init.uniform_(self.bias, -1 / math.sqrt(cin), 1 / math.sqrt(cin))


class _Indexed(nn.Module):
    """children named like the slots of the reference's nn.Sequential (activations hold no state)."""

    def __init__(self, **mods):
        super().__init__()
        for k, m in mods.items():
            self.add_module(k, m)

    def __getitem__(self, i):
        return getattr(self, str(i))


def _torch_default_conv_init(conv):
    nn.init.kaiming_uniform_(conv.weight, a=math.sqrt(5))          # nn.Conv2d.reset_parameters
    fan_in = conv.weight.shape[1] * conv.weight.shape[2] * conv.weight.shape[3]
    nn.init.uniform_(conv.bias, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))


class FlowAggregationHeadWithResidual(nn.Module):
    def __init__(self, args=None, ssim_sz=1, mask_layer=5, create_flownet=False, flow_feat_before_agg_kernel_size=3,
                 num_flow_feat_channels=64, outlier_robust_loss=False, eps=0.01, q=0.4, mask_size=(48, 48),
                 residual_adjustment_scale=10., norm_flow=False, clamp_flow_t=None, filter_flow_t=None,
                 free_residual=False, free_residual_with_affine=False, free_residual_with_affine_quadratic=False,
                 object_free_residual=False, free_scale=False, affine_residual=False, allow_residual_resize=False,
                 pred_div_coeff=10.):
        super().__init__()
        assert create_flownet
        if free_residual_with_affine_quadratic:
            assert free_residual_with_affine
        assert int(free_residual) + int(free_residual_with_affine) + int(object_free_residual) + int(free_scale) \
            + int(affine_residual) <= 1
        if object_free_residual or free_scale or affine_residual or not (free_residual or free_residual_with_affine):
            raise NotImplementedError("residual mode without a code path in the reference "
                                      "(models/flow_aggregation_head_with_residual.py:305-310)")
        if norm_flow or filter_flow_t is not None:
            raise NotImplementedError("norm_flow / filter_flow_t are not used by any RCF config")
        k, nf = flow_feat_before_agg_kernel_size, num_flow_feat_channels
        if nf != 64:
            raise NotImplementedError("num_flow_feat_channels must be 64 (the wavefront width the kernels map it to)")
        c1 = Conv2d(2, nf, k, padding=(k - 1) // 2, bias=True, act=1, slope=0.1)
        c2 = Conv2d(nf, nf, k, padding=(k - 1) // 2, bias=True, act=1, slope=0.1)
        _torch_default_conv_init(c1)
        _torch_default_conv_init(c2)
        self.flow_feat_before_agg = _Indexed(**{"0": c1, "2": c2})
        self.flow_feat_after_agg = _Indexed(**{"0": _Pointwise(nf, nf), "2": _Pointwise(nf, 2)})
        self.args = args
        self.mask_layer, self.mask_size, self.nf = mask_layer, tuple(mask_size), nf
        self.outlier_robust_loss, self.eps, self.q = outlier_robust_loss, eps, q
        self.residual_adjustment_scale, self.pred_div_coeff = residual_adjustment_scale, pred_div_coeff
        self.clamp_flow_t = clamp_flow_t
        self.free_residual, self.free_residual_with_affine = free_residual, free_residual_with_affine
        self.quadratic = free_residual_with_affine_quadratic
        self.allow_residual_resize = allow_residual_resize
        self._ws = None

    # ================================================================ HIP training path
    def softmax_masks(self, logits_nhwc, B, I):
        """NHWC logits [B*I,h,w,C] -> softmax masks [B,I,C,h,w] (models/rcf_model.py:430-433)."""
        C = self.mask_layer
        l = ops.nhwc_to_nchw(logits_nhwc, C)
        return F.softmax(l.view(B, I, C, *l.shape[-2:]), dim=2)

    def _cfg(self, model, B, logits_pitch, targets):
        c = FlowHeadCfg()
        c.B, c.C, (c.h, c.w) = B, self.mask_layer, self.mask_size
        c.logits_pitch, c.nf = logits_pitch, self.nf
        c.D = 0 if not self.free_residual_with_affine else (5 if self.quadratic else 2)
        c.robust = int(self.outlier_robust_loss)
        c.tanh_residual = int(self.free_residual_with_affine or self.residual_adjustment_scale != -1.)
        c.eps, c.q = self.eps, self.q
        c.clamp_t = -1.0 if self.clamp_flow_t is None else float(self.clamp_flow_t)
        c.res_scale, c.div_coeff = float(self.residual_adjustment_scale), float(self.pred_div_coeff)
        c.w_seg, c.w_entropy = float(model.w_seg), float(model.w_entropy)
        c.n_targets = len(targets)
        c.target_channel = int(model.args.object_channel) if targets else 0
        for i, (_, _, wpos, wneg, weight, th) in enumerate(targets):
            c.t_wpos[i], c.t_wneg[i], c.t_weight[i], c.t_thresh[i] = wpos, wneg, weight, th
        oc = getattr(model.args, "object_channel", None)
        c.w_compact, c.compact_channel = 0.0, 0
        if model.compactness_head is not None:               # models/compactness_head.py:19-27
            ch = model.compactness_head.compact_channel
            ch = oc if ch == -1 else ch
            if ch is not None:
                c.w_compact, c.compact_channel = float(model.w_compactness), int(ch)
        c.w_sharpen, c.t_sharpen, c.sharpen_mode = 0.0, float(model.t_sharpen), 0
        if model.w_sharpen > 0 and (oc is not None or not model.object_aware_sharpening):   # rcf_model.py:471
            c.w_sharpen = float(model.w_sharpen)
            c.sharpen_mode = 2 if model.object_aware_sharpening else 1
            if c.sharpen_mode == 2:
                c.target_channel = int(oc)
            c.w_entropy = 0.0                                 # `elif self.w_entropy > 0`, rcf_model.py:478
        return c

    def _workspace(self, cfg, device):
        need = _lib.load().rcf_flowhead_workspace_bytes(cfg)
        if need == 0:
            raise _lib.RcfHipError("unsupported flow-head configuration for the HIP kernels")
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)    # lives across fwd -> bwd
        return self._ws, need

    def loss_and_grads(self, model, logits, res, gfw, gbw, extra, B, I, want_flows=False, act_dtype=torch.float32):
        """Returns (losses, seed): seed(scale) writes d(scale*loss)/d logits and /d res into the Acts and
        accumulates this head's parameter gradients.  want_flows: also keep the reconstructed flows (overall,
        aggregated, residual adjustment, affine; [2B,2,h,w] per direction image) and the softmax masks in
        `self.last_flows` (the reference head returns them for its visualisations, :370-399)."""
        assert I == 2, "Other im_num not implemented"
        C, (h, w) = self.mask_layer, self.mask_size
        dev = logits.t.device
        assert tuple(logits.t.shape) == (B * I, h, w, logits.t.shape[3]) and logits.t.shape[3] >= C
        targets = []
        if model.w_pl > 0:
            targets.append(("loss_pl", extra["pl_masks"].contiguous(), model.pl_pos_weight, model.pl_neg_weight,
                            model.w_pl, float(model.pl_mask_pos_th)))
        if model.w_crf > 0:
            targets.append(("loss_crf", extra["crf_masks"].contiguous(), model.crf_pos_weight, model.crf_neg_weight,
                            model.w_crf, float(model.crf_mask_pos_th)))
        cfg = self._cfg(model, B, ops.pitch_of(logits.t), targets)
        ws, need = self._workspace(cfg, dev)
        st = _stream()
        gfw, gbw = gfw.contiguous(), gbw.contiguous()                           # [B,1,2,h,w]
        flow4 = torch.empty((B * I, h, w, 4), dtype=torch.float32, device=dev)
        _lib.call("rcf_flowhead_prepare_f32", cfg, _p(gfw), _p(gbw), _p(flow4), _p(ws), need, st)
        c1, c2 = self.flow_feat_before_agg[0], self.flow_feat_before_agg[2]
        l1, l2 = self.flow_feat_after_agg[0], self.flow_feat_after_agg[2]
        w1p = c1._packed_weight()
        a1 = ops.conv2d_fwd(flow4, w1p, c1.bias, 1, c1.padding, 1, act=1, slope=0.1)
        # bf16 step (act_dtype: the forward pass's activation type): the 64 -> 64 feature conv on the bf16-operand kernels, as
        # torch autocast runs it (models/flow_aggregation_head_with_residual.py:84-93 under configs/rcf_stv2's AMP); its
        # output stays fp32 for the fp32 loss tail.  a1 (Cin = 2 conv, fp32-MFMA) is rounded to bf16 once for all three directions.
        bf = act_dtype in ops.H16 and c2.cin % 8 == 0 and c2.cout % 8 == 0
        if bf:
            a1b = ops.cast(a1, act_dtype)
            feat = ops.conv2d_fwd_bf16(a1b, c2.weight, None, c2.bias, 1, c2.padding, 1, act=1, slope=0.1, out_dtype=torch.float32)
        else:
            feat = ops.conv2d_fwd(a1, c2.weight, c2.bias, 1, c2.padding, 1, act=1, slope=0.1)
        if tuple(res.t.shape[1:3]) != (h, w):
            if not self.allow_residual_resize:
                raise RuntimeError("residual size differs from mask_size and allow_residual_resize is off")
            R = ops.resize_nhwc_fwd(res.t, (h, w), False)                       # F.interpolate(bilinear), :272-273
        else:
            R = res.t
        assert R.is_contiguous() and R.shape[3] == 4 * C
        t0 = targets[0][1] if len(targets) > 0 else None
        t1 = targets[1][1] if len(targets) > 1 else None
        l5 = torch.empty(8, dtype=torch.float32, device=dev)
        fl = {}
        if want_flows:
            fl = {k: torch.zeros((B * I, 2, h, w), dtype=torch.float32, device=dev) for k in ("pred", "agg", "adj", "aff")}
            fl["masks"] = torch.empty((B * I, C, h, w), dtype=torch.float32, device=dev)
        _lib.call("rcf_flowhead_fwd_f32", cfg, _p(logits.t), _p(feat), _p(R), _p(l1.weight), _p(l1.bias), _p(l2.weight),
                  _p(l2.bias), _p(t0), _p(t1), _p(l5), _p(fl.get("masks")), _p(fl.get("pred")), _p(fl.get("agg")),
                  _p(fl.get("adj")), _p(fl.get("aff")), _p(ws), need, st)
        if want_flows:
            # the clamped ground-truth flows as the kernels saw them: channels 0..1 of the padded conv input
            fl["gt"] = ops.nhwc_to_nchw(flow4, 2)
            fl["l5"] = l5
            self.last_flows = fl
        seg = l5[0] + l5[1]
        losses = {"loss_warp_seg": seg}
        loss = seg * model.w_seg
        if cfg.sharpen_mode:
            losses["loss_sharpen"] = l5[6]
            loss = loss + l5[6] * model.w_sharpen
        elif model.w_entropy > 0:
            losses["loss_entropy"] = l5[2]
            loss = loss + l5[2] * model.w_entropy
        if cfg.w_compact != 0.0:
            losses["loss_compactness"] = l5[5]
            loss = loss + l5[5] * model.w_compactness
        for i, (name, _, _, _, weight, _) in enumerate(targets):
            losses[name] = l5[3 + i]
            loss = loss + l5[3 + i] * weight
        losses["loss"] = loss

        def seed(scale):
            dlogits = torch.empty_like(logits.t)
            dR = torch.empty_like(R)
            dfeat = torch.empty_like(feat)
            _lib.call("rcf_flowhead_bwd_f32", cfg, _p(feat), _p(R), _p(l1.weight), _p(l2.weight), _p(t0), _p(t1),
                      float(scale), _p(dlogits), _p(dR), _p(dfeat), _p(_param_grad(l1.weight)), _p(_param_grad(l1.bias)),
                      _p(_param_grad(l2.weight)), _p(_param_grad(l2.bias)), _p(ws), need, _stream())
            logits.grad = dlogits
            res.grad = dR if R is res.t else ops.resize_nhwc_bwd(dR, res.t.shape[1:3], False)
            # second 3x3 conv (64 -> 64): dfeat already carries the LeakyReLU derivative
            ops.colsum(dfeat, _param_grad(c2.bias), beta=1)
            if bf:
                dfb = ops.cast(dfeat, act_dtype)
                ops.conv2d_wgrad_bf16(a1b, dfb, c2.weight, _param_grad(c2.weight), 1, c2.padding, 1, beta=1)
                da1 = ops.cast(ops.conv2d_dgrad_bf16(dfb, c2.weight, a1.shape, 1, c2.padding, 1), torch.float32)
            else:
                ops.conv2d_wgrad(a1, dfeat, c2.weight, _param_grad(c2.weight), 1, c2.padding, 1, beta=1)
                da1 = ops.conv2d_dgrad(dfeat, c2.weight, a1.shape, 1, c2.padding, 1)
            _lib.call("rcf_lrelu_bwd_f32", _p(da1), _p(a1), _p(da1), da1.numel(), 0.1, _stream())
            # first 3x3 conv (2 -> 64, input zero-padded to 4 channels): no data gradient (input = RAFT flow)
            dwp = torch.empty_like(w1p)
            ops.conv2d_wgrad(flow4, da1, w1p, dwp, 1, c1.padding, 1, beta=0)
            dw = ops.nhwc_to_nchw(dwp.permute(0, 2, 3, 1), c1.cin)
            g = _param_grad(c1.weight)
            ops.copy2d(dw, dw.numel(), g, g.numel(), 1, dw.numel(), beta=1)
            ops.colsum(da1, _param_grad(c1.bias), beta=1)
        return losses, seed

    # ================================================================ stand-alone nn.Module surface
    def forward(self, imgs, masks, gt_fw_flows, gt_bw_flows, res_fw, res_bw):
        """nn.Module surface of the reference head (models/flow_aggregation_head_with_residual.py:312-399; NCHW
        tensors, masks ALREADY softmaxed [B,2,C,h,w]): (flows dict of normalised [B,4,h,w] visualisation flows,
        loss dict seg / seg_fw / seg_bw).  Runs on the same HIP kernels as the training step (softmax(log p) = p);
        inference only -- the training step differentiates through loss_and_grads, not through this."""
        import types
        from .layers import Act
        assert imgs.shape[1] == 2, "Other im_num not implemented"
        B, I, C, h, w = masks.shape
        lg = torch.log(masks.detach().float().clamp_min(1e-30)).reshape(B * I, C, h, w).contiguous()
        logits = Act(ops.nchw_to_nhwc(lg), needs_grad=False)
        res = Act(ops.nchw_to_nhwc(torch.cat([res_fw, res_bw], dim=1).detach().float().contiguous()), needs_grad=False)
        model = types.SimpleNamespace(w_seg=1.0, w_entropy=0.0, w_pl=0, w_crf=0, compactness_head=None, w_sharpen=0,
                                      t_sharpen=0.25, object_aware_sharpening=False,
                                      args=types.SimpleNamespace(object_channel=None))
        losses, _ = self.loss_and_grads(model, logits, res, gt_fw_flows.float(), gt_bw_flows.float(), {}, B, I,
                                        want_flows=True)
        f = self.last_flows
        s = torch.tensor([h / 2.0, w / 2.0], dtype=torch.float32, device=masks.device).view(1, 1, 2, 1, 1)

        def vis(t):                                   # get_norm_flow :18-30; direction images n = 2b + d -> [B,4,h,w]
            return (t.view(B, 2, 2, h, w) / s).reshape(B, 4, h, w)
        flows = {"gt_flow": [vis(f["gt"])], "pred_flow": [vis(f["pred"])], "agg_flow": [vis(f["agg"])],
                 "residual_adj": [vis(f["adj"])],
                 "affine_flow": [vis(f["aff"])] if self.free_residual_with_affine else []}
        return flows, {"seg_fw": f["l5"][0], "seg_bw": f["l5"][1], "seg": f["l5"][0] + f["l5"][1]}
