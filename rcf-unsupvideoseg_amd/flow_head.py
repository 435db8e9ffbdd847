"""Flow-aggregation head (`decode_head`) and the loss assembly of RCFModel.forward_train.

Reference: models/flow_aggregation_head_with_residual.py:50-148 (ctor), :150-162 (clamp), :164-233
(per-segment affine least squares), :235-310 (aggregate), :312-399 (forward); losses
models/rcf_model.py:350-408,433-434,464-523; compactness models/compactness_head.py:14-57.

Same constructor keywords / state-dict keys as the reference head.  `loss_and_grads` evaluates the
whole loss tail (softmax -> flow reconstruction -> L1 / entropy / pl / crf / compactness) and returns
a closure that seeds the tape with d loss / d logits and d loss / d residual.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .layers import Act, _param_grad


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
        self.flow_feat_before_agg = nn.Sequential(
            nn.Conv2d(2, nf, k, padding=(k - 1) // 2), nn.LeakyReLU(0.1),
            nn.Conv2d(nf, nf, k, padding=(k - 1) // 2), nn.LeakyReLU(0.1))
        self.flow_feat_after_agg = nn.Sequential(nn.Conv1d(nf, nf, 1), nn.LeakyReLU(0.1), nn.Conv1d(nf, 2, 1))
        self.args = args
        self.mask_layer, self.mask_size, self.nf = mask_layer, tuple(mask_size), nf
        self.outlier_robust_loss, self.eps, self.q = outlier_robust_loss, eps, q
        self.residual_adjustment_scale, self.pred_div_coeff = residual_adjustment_scale, pred_div_coeff
        self.clamp_flow_t = clamp_flow_t
        self.free_residual, self.free_residual_with_affine = free_residual, free_residual_with_affine
        self.quadratic = free_residual_with_affine_quadratic
        self.allow_residual_resize = allow_residual_resize

    # ---------------------------------------------------------------- pieces
    def softmax_masks(self, logits_nhwc, B, I):
        """NHWC logits [B*I,h,w,C] -> softmax masks [B,I,C,h,w] (models/rcf_model.py:430-433)."""
        C = self.mask_layer
        l = ops.nhwc_to_nchw(logits_nhwc, C)
        return F.softmax(l.view(B, I, C, *l.shape[-2:]), dim=2)

    def _coord_map(self, device):
        H, W = self.mask_size
        yy, xx = torch.meshgrid(torch.arange(H, device=device), torch.arange(W, device=device), indexing="ij")
        cols = [yy, xx] + ([yy * yy, xx * xx, yy * xx] if self.quadratic else [])
        return torch.stack(cols, dim=2).view(H * W, -1).float()

    def _affine(self, mask, flow):
        B, C, H, W = mask.shape
        w = (mask / mask.sum(dim=(2, 3), keepdim=True)).flatten(2)
        Fu = flow.flatten(2).permute(0, 2, 1)
        om = self._coord_map(mask.device)
        mu_F, mu_o = torch.bmm(w, Fu), w @ om
        Fd, od = Fu[:, None] - mu_F[:, :, None], om[None, None] - mu_o[:, :, None]
        S_Fo = torch.einsum("bcp,bcpk,bcpl->bckl", w, Fd, od)
        S_oo = torch.einsum("bcp,bcpk,bcpl->bckl", w, od, od)
        A = torch.linalg.solve(S_oo.float(), S_Fo.permute(0, 1, 3, 2).float()).permute(0, 1, 3, 2)
        pred = torch.einsum("bcjk,bclk->bclj", A, od).view(B, C, H, W, 2)
        return torch.einsum("bchw,bchwl->blhw", mask, pred)

    def _aggregate(self, mask, flow, residual):
        B, C, H, W = mask.shape
        mhat = mask / mask.flatten(2).sum(dim=2).view(B, C, 1, 1)
        feat = self.flow_feat_before_agg(flow)
        assert feat.shape[2:] == mask.shape[2:], f"{feat.shape[2:]} != {mask.shape[2:]}"
        u = self.flow_feat_after_agg(torch.einsum("bkhw,bchw->bkc", feat, mhat))
        agg = torch.einsum("bdc,bchw->bdhw", u, mask)
        affine = self._affine(mask, flow) if self.free_residual_with_affine else None
        if self.allow_residual_resize and tuple(residual.shape[-2:]) != self.mask_size:
            residual = F.interpolate(residual, self.mask_size, mode="bilinear")
        r = residual.unflatten(1, (2, self.mask_layer))
        if self.free_residual_with_affine or self.residual_adjustment_scale != -1.:
            adj = (torch.tanh(r / self.pred_div_coeff) * mask[:, None]).sum(dim=2) * self.residual_adjustment_scale
        else:
            adj = (r * mask[:, None]).sum(dim=2)
        overall = agg + adj if affine is None else agg + affine + adj
        return overall, agg, adj, affine

    def _flow_loss(self, gt, pred):
        d = (gt - pred).abs().view(-1)
        return ((d + self.eps) ** self.q).mean() if self.outlier_robust_loss else d.mean()

    def flow_losses(self, masks, gfw, gbw, res_fw, res_bw):
        clamp = (lambda f: f.clamp(min=-self.clamp_flow_t, max=self.clamp_flow_t)) if self.clamp_flow_t is not None \
            else (lambda f: f)
        gt_fw, gt_bw = clamp(gfw[:, 0]), clamp(gbw[:, 0])
        fw = self._aggregate(masks[:, 0], gt_fw, res_fw)
        bw = self._aggregate(masks[:, 1], gt_bw, res_bw)
        l_fw, l_bw = self._flow_loss(gt_fw, fw[0]), self._flow_loss(gt_bw, bw[0])
        return {"seg_fw": l_fw, "seg_bw": l_bw, "seg": l_fw + l_bw}, (fw, bw, gt_fw, gt_bw)

    def forward(self, imgs, masks, gt_fw_flows, gt_bw_flows, res_fw, res_bw):
        """nn.Module surface of the reference head (NCHW tensors): (flows dict, loss dict)."""
        assert imgs.shape[1] == 2, "Other im_num not implemented"
        loss, (fw, bw, gt_fw, gt_bw) = self.flow_losses(masks, gt_fw_flows, gt_bw_flows, res_fw, res_bw)

        def vis(a, b):                               # get_norm_flow :18-30
            h, w = a.shape[-2:]
            s = torch.tensor([h / 2.0, w / 2.0], dtype=a.dtype, device=a.device).view(1, 2, 1, 1)
            return torch.cat([a / s, b / s], dim=1)
        flows = {"gt_flow": [vis(gt_fw, gt_bw)], "pred_flow": [vis(fw[0], bw[0])], "agg_flow": [vis(fw[1], bw[1])],
                 "residual_adj": [vis(fw[2], bw[2])],
                 "affine_flow": [vis(fw[3], bw[3])] if fw[3] is not None else []}
        return flows, loss

    # ---------------------------------------------------------------- loss tail of forward_train
    def loss_and_grads(self, model, logits, res, gfw, gbw, extra, B, I):
        """Returns (losses, seed) -- seed(scale) puts d(scale*loss)/d logits, /d res into the Acts and
        accumulates this head's parameter gradients."""
        C = self.mask_layer
        with torch.enable_grad():
            l_nchw = ops.nhwc_to_nchw(logits.t, C).requires_grad_(True)          # [B*I,C,h,w]
            r_nchw = ops.nhwc_to_nchw(res.t, 4 * C).requires_grad_(True)          # [B,4C,h2,w2]
            p = F.softmax(l_nchw.view(B, I, C, *l_nchw.shape[-2:]), dim=2)
            logp = F.log_softmax(p, dim=2)                                        # double softmax (rcf_model.py:434)
            lf, _ = self.flow_losses(p, gfw, gbw, r_nchw[:, :2 * C], r_nchw[:, 2 * C:])
            losses = {"loss_warp_seg": lf["seg"]}
            loss = lf["seg"] * model.w_seg
            if model.w_entropy > 0:
                le = -(p * logp).sum(dim=2).mean()
                loss = loss + le * model.w_entropy
                losses["loss_entropy"] = le
            if model.compactness_head is not None:
                lc = model.compactness_head.get_compactness_loss(p)
                if lc is not None:
                    losses["loss_compactness"] = lc
                    loss = loss + lc * model.w_compactness
            oc = getattr(model.args, "object_channel", None)

            def asym_mse(target, pred, wpos, wneg):
                d = target - pred
                return (d.clamp(min=0) ** 2).mean() * wpos + (d.clamp(max=0) ** 2).mean() * wneg
            if model.w_pl > 0:
                t = extra["pl_masks"]
                if model.pl_mask_pos_th != -1:
                    t = (t > model.pl_mask_pos_th).float()
                lp = asym_mse(t, p[:, :, oc], model.pl_pos_weight, model.pl_neg_weight)
                losses["loss_pl"] = lp
                loss = loss + lp * model.w_pl
            if model.w_crf > 0:
                t = extra["crf_masks"]
                if model.crf_mask_pos_th != -1.:
                    t = (t > model.crf_mask_pos_th).float()
                lcrf = asym_mse(t, p[:, :, oc], model.crf_pos_weight, model.crf_neg_weight)
                losses["loss_crf"] = lcrf
                loss = loss + lcrf * model.w_crf
            params = [q for q in self.parameters() if q.requires_grad]
        losses = {k: v.detach() for k, v in losses.items()}
        losses["loss"] = loss.detach()

        def seed(scale):
            # called from inside autograd's backward (grad mode off): scale through grad_outputs
            grads = torch.autograd.grad(loss, [l_nchw, r_nchw] + params, allow_unused=True,
                                        grad_outputs=torch.as_tensor(scale, dtype=loss.dtype, device=loss.device))
            logits.grad = ops.nchw_to_nhwc(grads[0].contiguous(), logits.t.shape[3])
            res.grad = ops.nchw_to_nhwc(grads[1].contiguous(), res.t.shape[3])
            for q, g in zip(params, grads[2:]):
                if g is not None:
                    _param_grad(q).add_(g)
        return losses, seed
