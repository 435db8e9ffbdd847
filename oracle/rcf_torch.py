"""ORACLE (test infrastructure, never shipped): CPU restatement of RCF's training hot path.

Plain PyTorch fp32 on CPU, no mmcv / mmseg / lightning.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file; the
product package (`rcf-unsupvideoseg_amd/`) never does.

Parity status: PINNED for everything except the CRF call — `tests/golden/make_golden.py`
imports the reference itself (with stand-in modules for its absent third-party imports)
in the build container, feeds both the same weights and inputs and asserts agreement;
the captured vectors live in `tests/golden/*.npz`.  The CRF (reference = CUDA-only
`tools/torchCRF`) is restated in `oracle/crf_ref.c`: parity unpinned (no reference run
possible here), see DESIGN.md.

Every class/function cites the reference file:line (under /root/reference) it follows.
State-dict keys equal the reference's so one weight file loads
into the reference, this oracle and the HIP product.
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- helpers

def resize_bilinear(x, size, align_corners=False):
    """mmseg.ops.resize as used by models/rcf_model.py:213-220, models/decode_head.py:157-163."""
    return F.interpolate(x, size=tuple(int(s) for s in size), mode="bilinear",
                         align_corners=align_corners)


def make_norm(norm_cfg, num_features):
    """mmcv build_norm_layer for the two types the RCF configs use
    (configs/rcf/rcf_stage1.yaml:83-85).  SyncBN == BN over the global batch; on one
    process they coincide, so the oracle always uses BatchNorm2d (eps 1e-5, momentum 0.1)."""
    t = (norm_cfg or {}).get("type", "BN")
    if t not in ("BN", "SyncBN"):
        raise NotImplementedError(f"norm type {t}")
    bn = nn.BatchNorm2d(num_features, eps=1e-5, momentum=0.1)
    for p in bn.parameters():
        p.requires_grad = bool((norm_cfg or {}).get("requires_grad", True))
    return bn


def kaiming_normal_fan_out_(conv):
    """mmcv kaiming_init default (mode fan_out, relu, normal) — models/resnet.py:609-611."""
    nn.init.kaiming_normal_(conv.weight, a=0, mode="fan_out", nonlinearity="relu")
    if conv.bias is not None:
        nn.init.constant_(conv.bias, 0)


# ----------------------------------------------------------------------------- backbone

class Bottleneck(nn.Module):
    """models/resnet.py:95-302, style 'pytorch' (stride on the 3x3)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride, dilation, downsample, norm_cfg):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = make_norm(norm_cfg, planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation,
                               dilation=dilation, bias=False)
        self.bn2 = make_norm(norm_cfg, planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = make_norm(norm_cfg, planes * 4)
        self.downsample = downsample

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        identity = x if self.downsample is None else self.downsample(x)
        return F.relu(out + identity)          # models/resnet.py:291,300


class ResNet(nn.Module):
    """models/resnet.py:371-466 (constructor), :630-645 (forward), :598-628 (init);
    stage assembly models/res_layer.py:26-94 incl. contract_dilation (:66-70)."""
    blocks_per_depth = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth=50, in_channels=3, stem_channels=64, base_channels=64,
                 num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(0, 1, 2, 3), style="pytorch", norm_cfg=None, norm_eval=False,
                 contract_dilation=False, zero_init_residual=True, **unsupported):
        super().__init__()
        for k, v in unsupported.items():
            if v not in (None, False, -1, (False, False, False, False)):
                raise NotImplementedError(f"ResNet option {k}={v!r} is off the RCF path")
        assert style == "pytorch" and depth in self.blocks_per_depth
        norm_cfg = norm_cfg or dict(type="BN", requires_grad=True)
        self.out_indices = tuple(out_indices)
        self.norm_eval = norm_eval
        self.zero_init_residual = zero_init_residual
        self.conv1 = nn.Conv2d(in_channels, stem_channels, 7, stride=2, padding=3, bias=False)
        self.bn1 = make_norm(norm_cfg, stem_channels)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inplanes = stem_channels
        self.num_stages = num_stages
        for i, nblocks in enumerate(self.blocks_per_depth[depth][:num_stages]):
            planes = base_channels * 2 ** i
            stride, dil = strides[i], dilations[i]
            down = None
            if stride != 1 or inplanes != planes * 4:
                down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                     make_norm(norm_cfg, planes * 4))
            first_dil = dil // 2 if (dil > 1 and contract_dilation) else dil
            blocks = [Bottleneck(inplanes, planes, stride, first_dil, down, norm_cfg)]
            inplanes = planes * 4
            blocks += [Bottleneck(inplanes, planes, 1, dil, None, norm_cfg) for _ in range(1, nblocks)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                kaiming_normal_fan_out_(m)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        outs = []
        for i in range(self.num_stages):
            x = getattr(self, f"layer{i + 1}")(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        if mode and self.norm_eval:               # models/resnet.py:652-656
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self


# ----------------------------------------------------------------------------- FCN head

class ConvModule(nn.Module):
    """mmcv ConvModule(conv -> norm -> ReLU) as instantiated by models/fcn_head.py:107-130:
    bias-free conv when a norm follows, kaiming-normal(fan_out) conv init, BN weight 1."""

    def __init__(self, cin, cout, kernel_size, padding, dilation, stride, norm_cfg):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=norm_cfg is None)
        self.with_norm = norm_cfg is not None
        if self.with_norm:
            self.bn = make_norm(norm_cfg, cout)
        kaiming_normal_fan_out_(self.conv)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = self.bn(x)
        return F.relu(x)


class FixedDropout2d(nn.Module):
    """nn.Dropout2d with its random draw replaced by a given one: `scale` [N, C] holds 0 for a dropped (sample, channel) plane
    and 1 / (1 - p) for a kept one.  Test infrastructure: assigned to `head.dropout` of the reference's / the oracle's FCNHead
    (models/fcn_head.py:144-145) so that both and the HIP path (FCNHead.keep_mask) see the same planes dropped."""

    def __init__(self, scale):
        super().__init__()
        self.scale = torch.as_tensor(scale)

    def forward(self, x):
        assert tuple(x.shape[:2]) == tuple(self.scale.shape), (tuple(x.shape), tuple(self.scale.shape))
        return x * self.scale.to(x.dtype)[:, :, None, None]


class FCNHead(nn.Module):
    """models/fcn_head.py:50-140,142-147,211-218 + models/decode_head.py:45-90,141-170.
    The create_flownet=True branch (PWC-Lite, AMD baseline only) is off the RCF path."""

    def __init__(self, in_channels, channels, *, num_classes, num_convs=2, kernel_size=3,
                 concat_input=True, dilation=1, input_stride=1, input_dilation=None,
                 dropout_ratio=0.1, norm_cfg=None, in_index=-1, input_transform=None,
                 align_corners=False, transform_scale=None, create_flownet=False,
                 conv_cfg=None, act_cfg=None, loss_decode=None, ignore_index=255, sampler=None,
                 mask_layer=1, ssim_sz=1, load_flownet=False, freeze_flownet=False,
                 flow_model_path=""):
        super().__init__()
        if create_flownet:
            raise NotImplementedError("FCNHead(create_flownet=True) belongs to the AMD baseline")
        assert num_convs > 0 and conv_cfg is None and sampler is None
        self.input_transform, self.in_index = input_transform, in_index
        if input_transform == "resize_concat":
            in_channels = sum(in_channels)
        elif input_transform is not None:
            raise NotImplementedError(input_transform)
        self.in_channels, self.channels, self.num_classes = in_channels, channels, num_classes
        self.align_corners, self.transform_scale = align_corners, transform_scale
        self.concat_input = concat_input
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None
        self.conv_seg = nn.Conv2d(channels, num_classes, 1)
        if input_dilation is None:
            input_dilation = dilation
        convs = [ConvModule(in_channels, channels, kernel_size, input_dilation, input_dilation,
                            input_stride, norm_cfg)]
        convs += [ConvModule(channels, channels, kernel_size, dilation, dilation, 1, norm_cfg)
                  for _ in range(num_convs - 1)]
        self.convs = nn.Sequential(*convs)
        if concat_input:
            self.conv_cat = ConvModule(in_channels + channels, channels, kernel_size, dilation,
                                       dilation, 1, norm_cfg)

    def _transform_inputs(self, inputs):
        if self.input_transform == "resize_concat":
            sel = [inputs[i] for i in self.in_index]
            size = sel[0].shape[2:]
            if self.transform_scale is not None:
                size = tuple(s * self.transform_scale for s in size)
            return torch.cat([resize_bilinear(x, size, self.align_corners) for x in sel], dim=1)
        return inputs[self.in_index]

    def forward(self, inputs):
        x = self._transform_inputs(inputs)
        out = self.convs(x)
        if self.concat_input:
            out = self.conv_cat(torch.cat([x, out], dim=1))
        if self.dropout is not None:
            out = self.dropout(out)
        return self.conv_seg(out)


# ----------------------------------------------------------------------------- flow head

class FlowAggregationHeadWithResidual(nn.Module):
    """models/flow_aggregation_head_with_residual.py:50-148 (ctor), :150-162 (clamp),
    :164-233 (affine LS), :235-310 (aggregate), :312-399 (forward)."""

    def __init__(self, args=None, ssim_sz=1, mask_layer=5, create_flownet=False,
                 flow_feat_before_agg_kernel_size=3, num_flow_feat_channels=64,
                 outlier_robust_loss=False, eps=0.01, q=0.4, mask_size=(48, 48),
                 residual_adjustment_scale=10., norm_flow=False, clamp_flow_t=None,
                 filter_flow_t=None, free_residual=False, free_residual_with_affine=False,
                 free_residual_with_affine_quadratic=False, object_free_residual=False,
                 free_scale=False, affine_residual=False, allow_residual_resize=False,
                 pred_div_coeff=10.):
        super().__init__()
        assert create_flownet
        if free_residual_with_affine_quadratic:
            assert free_residual_with_affine
        assert int(free_residual) + int(free_residual_with_affine) + int(object_free_residual) \
            + int(free_scale) + int(affine_residual) <= 1
        if object_free_residual or free_scale or affine_residual:
            raise NotImplementedError("flag accepted by the reference but with no code path "
                                      "(models/flow_aggregation_head_with_residual.py:305-310)")
        k, nf = flow_feat_before_agg_kernel_size, num_flow_feat_channels
        self.flow_feat_before_agg = nn.Sequential(
            nn.Conv2d(2, nf, k, padding=(k - 1) // 2), nn.LeakyReLU(0.1),
            nn.Conv2d(nf, nf, k, padding=(k - 1) // 2), nn.LeakyReLU(0.1))
        self.flow_feat_after_agg = nn.Sequential(
            nn.Conv1d(nf, nf, 1), nn.LeakyReLU(0.1), nn.Conv1d(nf, 2, 1))
        self.mask_layer, self.mask_size = mask_layer, tuple(mask_size)
        self.outlier_robust_loss, self.eps, self.q = outlier_robust_loss, eps, q
        self.residual_adjustment_scale, self.pred_div_coeff = residual_adjustment_scale, pred_div_coeff
        self.norm_flow, self.clamp_flow_t, self.filter_flow_t = norm_flow, clamp_flow_t, filter_flow_t
        self.free_residual = free_residual
        self.free_residual_with_affine = free_residual_with_affine
        self.quadratic = free_residual_with_affine_quadratic
        self.allow_residual_resize = allow_residual_resize

    def coord_map(self, device, dtype=torch.float32):
        """:135-148 — (row, col[, row^2, col^2, row*col]) per pixel; float32 in the reference (a
        float64 run of this oracle serves as numerical ground truth in the tests)."""
        H, W = self.mask_size
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        cols = [yy, xx] + ([yy * yy, xx * xx, yy * xx] if self.quadratic else [])
        return torch.stack(cols, dim=2).view(H * W, -1).to(dtype).to(device)

    def norm_and_clamp_flow(self, flow):
        if self.norm_flow:
            flow = flow / flow.abs().max()
        if self.clamp_flow_t is not None:
            flow = flow.clamp(min=-self.clamp_flow_t, max=self.clamp_flow_t)
        if self.filter_flow_t is not None:
            flow = torch.where(flow.abs() < self.filter_flow_t, torch.zeros_like(flow), flow)
        return flow

    def demean_affine_flow(self, mask, flow):
        """:164-233 — per (b, c) weighted least squares A* = S_Fw S_ww^-1, returned as
        sum_c mask_c * A*_c (w - mu_w,c)."""
        B, C, H, W = mask.shape
        w = (mask / mask.sum(dim=(2, 3), keepdim=True)).flatten(2)            # [B,C,HW]
        Fu = flow.flatten(2).permute(0, 2, 1)                                 # [B,HW,2]
        om = self.coord_map(mask.device, mask.dtype)                          # [HW,D]
        mu_F = torch.bmm(w, Fu)                                               # [B,C,2]
        mu_o = w @ om                                                         # [B,C,D]
        Fd = Fu[:, None] - mu_F[:, :, None]                                   # [B,C,HW,2]
        od = om[None, None] - mu_o[:, :, None]                                # [B,C,HW,D]
        S_Fo = torch.einsum("bcp,bcpk,bcpl->bckl", w, Fd, od)                 # [B,C,2,D]
        S_oo = torch.einsum("bcp,bcpk,bcpl->bckl", w, od, od)                 # [B,C,D,D]
        cast = (lambda t: t) if S_oo.dtype == torch.float64 else (lambda t: t.float())   # :216-217 forces fp32
        A = torch.linalg.solve(cast(S_oo), cast(S_Fo.permute(0, 1, 3, 2))).permute(0, 1, 3, 2)
        pred = torch.einsum("bcjk,bclk->bclj", A, od).view(B, C, H, W, 2)
        return torch.einsum("bchw,bchwl->blhw", mask, pred)

    def aggregate_flow_with_residual(self, mask, flow, residual):
        B, C, H, W = mask.shape
        mhat = mask / mask.view(B, C, H * W, 1).sum(dim=2, keepdim=True).view(B, C, 1, 1)
        feat = self.flow_feat_before_agg(flow)                                # [B,64,H,W]
        assert feat.shape[2:] == mask.shape[2:], f"{feat.shape[2:]} != {mask.shape[2:]}"
        pooled = torch.einsum("bkhw,bchw->bkc", feat, mhat)                   # [B,64,C]
        u = self.flow_feat_after_agg(pooled)                                  # [B,2,C]
        agg = torch.einsum("bdc,bchw->bdhw", u, mask)
        affine, adj = None, None
        if self.free_residual or self.free_residual_with_affine:
            if self.free_residual_with_affine:
                affine = self.demean_affine_flow(mask, flow)
            if self.allow_residual_resize and residual.shape[-2:] != self.mask_size:
                residual = F.interpolate(residual, self.mask_size, mode="bilinear")
            r = residual.unflatten(1, (2, self.mask_layer))
            if self.free_residual_with_affine or self.residual_adjustment_scale != -1.:
                adj = (torch.tanh(r / self.pred_div_coeff) * mask[:, None]).sum(dim=2) \
                    * self.residual_adjustment_scale
            else:
                adj = (r * mask[:, None]).sum(dim=2)
            overall = agg + adj if affine is None else agg + affine + adj
        else:
            raise NotImplementedError("no residual mode: the reference returns an unbound "
                                      "residual_adjustment here (:305-310)")
        return overall, agg, adj, affine

    @staticmethod
    def _vis_norm(a, b):
        """get_norm_flow :18-30 — ch0 / (h/2), ch1 / (w/2), fw and bw concatenated."""
        h, w = a.shape[-2:]
        s = torch.tensor([h / 2.0, w / 2.0], dtype=a.dtype, device=a.device).view(1, 2, 1, 1)
        return torch.cat([a / s, b / s], dim=1)

    def _loss(self, gt, pred):
        d = (gt - pred).abs().view(-1)
        return ((d + self.eps) ** self.q).mean() if self.outlier_robust_loss else d.mean()

    def forward(self, imgs, masks, gt_fw_flows, gt_bw_flows, res_fw, res_bw):
        assert imgs.shape[1] == 2, "Other im_num not implemented"
        gt_fw = self.norm_and_clamp_flow(gt_fw_flows[:, 0])
        gt_bw = self.norm_and_clamp_flow(gt_bw_flows[:, 0])
        fw = self.aggregate_flow_with_residual(masks[:, 0], gt_fw, res_fw)
        bw = self.aggregate_flow_with_residual(masks[:, 1], gt_bw, res_bw)
        loss = {"seg_fw": self._loss(gt_fw, fw[0]), "seg_bw": self._loss(gt_bw, bw[0])}
        loss["seg"] = loss["seg_fw"] + loss["seg_bw"]
        flows = {"gt_flow": [self._vis_norm(gt_fw, gt_bw)],
                 "pred_flow": [self._vis_norm(fw[0], bw[0])],
                 "agg_flow": [self._vis_norm(fw[1], bw[1])],
                 "residual_adj": [self._vis_norm(fw[2], bw[2])],
                 "affine_flow": [self._vis_norm(fw[3], bw[3])] if fw[3] is not None else []}
        return flows, loss


# ----------------------------------------------------------------------------- small heads

class CompactnessHead(nn.Module):
    """models/compactness_head.py:14-57."""

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


class CRFHead(nn.Module):
    """models/crf_head.py:12-31 (ctor), :33-37, :39-60, :93-109.  `crf_soft` is injected:
    tests bind it to the C restatement (oracle/crf_ref.c) through ctypes."""

    def __init__(self, args=None, srgb=5., scomp=5., sxy=60., scomp_smooth=0., sxy_smooth=0.,
                 refine_iters=50, crf_scale=0.7, mean=(0.485, 0.456, 0.406),
                 std=(0.229, 0.224, 0.225), crf_soft=None):
        super().__init__()
        self.srgb, self.scomp, self.sxy = srgb, scomp, sxy
        self.scomp_smooth, self.sxy_smooth = scomp_smooth, sxy_smooth
        self.refine_iters, self.crf_scale = refine_iters, crf_scale
        self.mean = torch.tensor(mean, dtype=torch.float)[None, :, None, None]
        self.std = torch.tensor(std, dtype=torch.float)[None, :, None, None]
        self.crf_soft = crf_soft

    def to_uint8_image(self, imgs, unstandardize=True):
        if unstandardize:
            imgs = imgs * self.std + self.mean
        imgs = imgs.permute(0, 2, 3, 1) * 255.
        return imgs.clamp(min=0., max=255.).type(torch.uint8)

    def unary(self, mask):
        """:43-55 — quantise to u8, divide by max, clamp, -log([1-U, U]) as [HW,2]."""
        q = (mask * 255. / self.crf_scale).clip(min=0, max=255).type(torch.uint8)
        U = q / (torch.max(q) + 1e-8)
        U = torch.clamp(U, 1e-6, 1.0 - 1e-6)
        UU = -torch.log(torch.stack([1.0 - U, U], dim=0))
        return q, UU.view(2, -1).T.contiguous()

    def forward(self, imgs, masks, unstandardize=True):
        imgs = self.to_uint8_image(imgs, unstandardize)
        out = []
        for img, mask in zip(imgs, masks):
            H, W, _ = img.shape
            _, UU = self.unary(mask)
            out.append(self.crf_soft(img.contiguous(), UU, W, H, self.scomp_smooth, self.sxy_smooth,
                                     self.scomp, self.sxy, self.srgb, self.refine_iters).float())
        return torch.stack(out, dim=0)


# ----------------------------------------------------------------------------- EMA helpers

@torch.no_grad()
def copy_param_and_buffer(src, dest):
    """utils/model_utils.py:12-19."""
    s, d = src.state_dict(), dest.state_dict()
    assert list(s.keys()) == list(d.keys())
    for k in s:
        d[k].data.copy_(s[k])


@torch.no_grad()
def momentum_update_param_and_buffer(src, dest, m):
    """utils/model_utils.py:33-38 — every state-dict entry, incl. num_batches_tracked
    (int64: the float result is truncated by copy_)."""
    s, d = src.state_dict(), dest.state_dict()
    for k in s:
        d[k].data.copy_(d[k].data * m + s[k].data * (1.0 - m))


def sharpen(p, T, dim=1):
    """utils/loss_utils.py:105-108."""
    sp = p ** (1. / T)
    return sp / torch.sum(sp, dim=dim, keepdim=True)


# ----------------------------------------------------------------------------- model

_REGISTRY = {}


class RCFModel(nn.Module):
    """models/rcf_model.py:28-153 (ctor), :410-611 (forward_train), :275-320 (forward_eval),
    :350-408 (losses), :613-626 (forward).  JPEG visualisation / PNG export (:241-273,
    :456-462,:562-608) is not part of the arithmetic and is left out."""

    def __init__(self, args, backbone2, decode_head, decode_head2, decode_head3,
                 compactness_head=None, crf_head=None, crf_use_ema=False, ema_m=0.999, w_seg=2.0,
                 w_sharpen=0, t_sharpen=0.25, w_entropy=0, w_compactness=0, w_pl=0,
                 pl_pos_weight=1., pl_neg_weight=1., pl_mask_pos_th=0.35, w_crf=0,
                 crf_pos_weight=1., crf_neg_weight=1., crf_mask_pos_th=-1., mask_layer=1,
                 train_iter=0, train_cfg=None, test_cfg=None, align_corners=False,
                 mask_size=(48, 48), log_interval=50, freeze_backbone=False,
                 object_aware_sharpening=False, separate_residual=False, allow_mask_resize=False):
        super().__init__()
        from copy import deepcopy
        self.args = args
        reg = _REGISTRY

        def build(cfg, **extra):
            cfg = dict(cfg)
            ema = cfg.pop("create_ema", False)
            mod = reg[cfg.pop("type")](**extra, **cfg)
            mod_ema = None
            if ema:
                mod_ema = deepcopy(mod)
                for p in mod_ema.parameters():
                    p.requires_grad = False
                mod_ema.eval()
            return mod, mod_ema

        self.backbone2, self.backbone2_ema = build(backbone2)
        self.align_corners, self.mask_layer = align_corners, mask_layer
        self.decode_head, _ = build(decode_head, args=args)
        self.decode_head2, self.decode_head2_ema = build(decode_head2)
        self.num_classes = self.decode_head2.num_classes
        self.decode_head3, _ = build(decode_head3)
        self.w_compactness = w_compactness
        self.compactness_head = None
        if compactness_head:
            self.compactness_head, _ = build(compactness_head, args=args)
            assert w_compactness != 0
        self.backbone2.init_weights()
        if freeze_backbone:
            for p in self.backbone2.parameters():
                p.requires_grad_(False)
        self.train_iter = train_iter
        self.w_seg, self.w_sharpen, self.t_sharpen, self.w_entropy = w_seg, w_sharpen, t_sharpen, w_entropy
        assert not (w_sharpen != 0 and w_entropy != 0)
        self.w_pl = w_pl
        if w_pl > 0:
            assert args.object_channel is not None
        self.pl_pos_weight, self.pl_neg_weight, self.pl_mask_pos_th = pl_pos_weight, pl_neg_weight, pl_mask_pos_th
        self.w_crf, self.crf_head = w_crf, None
        if crf_head:
            self.crf_head, _ = build(crf_head, args=args)
            assert w_crf != 0
        self.crf_pos_weight, self.crf_neg_weight, self.crf_mask_pos_th = crf_pos_weight, crf_neg_weight, crf_mask_pos_th
        self.crf_use_ema, self.ema_m = crf_use_ema, ema_m
        self.log_interval = log_interval
        self.mask_size, self.allow_mask_resize = tuple(mask_size), allow_mask_resize
        self.object_aware_sharpening, self.separate_residual = object_aware_sharpening, separate_residual
        self.eval_on_ema = getattr(args, "eval_on_ema", False)
        if self.backbone2_ema is not None:
            copy_param_and_buffer(self.backbone2, self.backbone2_ema)
        if self.decode_head2_ema is not None:
            copy_param_and_buffer(self.decode_head2, self.decode_head2_ema)

    # NOTE: like the reference, there is NO train() override: the EMA copies are put in eval mode once, at
    # construction (models/rcf_model.py:171,187), and a later `model.train()` -- which Lightning issues on the whole
    # module tree before fitting and after every validation run -- switches them to training mode with everything
    # else: under main.py the teacher of stage 2.1 normalises with BATCH statistics (and updates its own running
    # statistics, and draws its own Dropout2d mask).  Pinned by tests/golden/make_golden_stage2.py.

    def resize(self, x, shape):
        return resize_bilinear(x, shape, self.align_corners)

    # -- losses (models/rcf_model.py:350-408)
    def get_entropy_loss(self, p, logp):
        return -(p * logp).sum(dim=2).mean()

    def get_sharpen_loss(self, p, logp, object_channel=None):
        if self.object_aware_sharpening:
            obj = p[:, :, object_channel]
            rest = p.detach().clone()
            rest[:, :, object_channel] = 0.
            diff = (obj - rest.max(dim=2).values).abs()
            return (self.t_sharpen - diff).clamp(min=0).mean()
        target = sharpen(p.detach(), self.t_sharpen, dim=2)
        return F.kl_div(logp, target, reduction="none").mean()

    @staticmethod
    def _asym_mse(target, pred, wpos, wneg):
        d = target - pred
        return (d.clamp(min=0) ** 2).mean() * wpos + (d.clamp(max=0) ** 2).mean() * wneg

    def get_pl_loss(self, p, pl):
        if self.pl_mask_pos_th != -1:
            pl = (pl > self.pl_mask_pos_th).float()
        return self._asym_mse(pl, p[:, :, self.args.object_channel], self.pl_pos_weight, self.pl_neg_weight)

    def get_crf_loss(self, p, crf):
        if self.crf_mask_pos_th != -1.:
            crf = (crf > self.crf_mask_pos_th).float()
        return self._asym_mse(crf, p[:, :, self.args.object_channel], self.crf_pos_weight, self.crf_neg_weight)

    def pred_residual(self, feats, B, I):
        if self.separate_residual:                 # :322-335
            feats = [f.unflatten(0, (B, I)).flatten(1, 2) for f in feats]
            r = self.decode_head3(feats)
            return r[:, :2 * self.num_classes], r[:, 2 * self.num_classes:]
        f = feats[-1].unflatten(0, (B, I))         # :337-348
        return self.decode_head3([f.flatten(1, 2)]), self.decode_head3([f[:, [1, 0]].flatten(1, 2)])

    def forward_train(self, imgs, gt_fw_flows, gt_bw_flows, pl_masks=None):
        B, I, C3, H, W = imgs.shape
        nflow = gt_fw_flows.shape[1]
        img_3 = imgs.view(B * I, C3, H, W)
        feats = self.backbone2(img_3)
        logits = self.decode_head2(feats)
        if self.allow_mask_resize and tuple(logits.shape[-2:]) != self.mask_size:
            logits = self.resize(logits, self.mask_size)
        res_fw, res_bw = self.pred_residual(feats, B, I)
        fh, fw_ = logits.shape[-2:]
        p = F.softmax(logits.view(B, I, self.mask_layer, fh, fw_), dim=2)
        logp = F.log_softmax(p, dim=2)              # double softmax, :433-434
        gfw = self.resize(gt_fw_flows.view(B * nflow, *gt_fw_flows.shape[2:]), self.mask_size)
        gbw = self.resize(gt_bw_flows.view(B * nflow, *gt_bw_flows.shape[2:]), self.mask_size)
        gfw = gfw.view(B, nflow, 2, *self.mask_size)
        gbw = gbw.view(B, nflow, 2, *self.mask_size)
        flows, loss_flow = self.decode_head(imgs, p, gfw, gbw, res_fw, res_bw)
        losses = {"loss_warp_seg": loss_flow["seg"]}
        loss = loss_flow["seg"] * self.w_seg
        oc = getattr(self.args, "object_channel", None)
        if self.w_sharpen > 0 and (oc is not None or not self.object_aware_sharpening):
            ls = self.get_sharpen_loss(p, logp, oc if self.object_aware_sharpening else None)
            loss = loss + ls * self.w_sharpen
            losses["loss_sharpen"] = ls
        elif self.w_entropy > 0:
            le = self.get_entropy_loss(p, logp)
            loss = loss + le * self.w_entropy
            losses["loss_entropy"] = le
        if self.compactness_head:
            lc = self.compactness_head.get_compactness_loss(p)
            if lc is not None:
                losses["loss_compactness"] = lc
                loss = loss + lc * self.w_compactness
        if self.w_pl > 0:
            lp = self.get_pl_loss(p, self.resize(pl_masks, self.mask_size))
            losses["loss_pl"] = lp
            loss = loss + lp * self.w_pl
        if self.w_crf > 0:
            if self.crf_use_ema:
                with torch.no_grad():
                    pe = self.decode_head2_ema(self.backbone2_ema(img_3))
                    p_crf = F.softmax(pe.view(B, I, self.mask_layer, fh, fw_), dim=2)
            else:
                p_crf = p
            up = self.resize(p_crf.detach().flatten(0, 1)[:, oc:oc + 1], img_3.shape[-2:])
            crf = self.crf_head(img_3, up[:, 0]).unflatten(0, (B, I))
            crf = self.resize(crf, self.mask_size)
            lcrf = self.get_crf_loss(p, crf)
            losses["loss_crf"] = lcrf
            loss = loss + lcrf * self.w_crf
            losses["_crf_masks"] = crf
        if self.backbone2_ema is not None:
            momentum_update_param_and_buffer(self.backbone2, self.backbone2_ema, self.ema_m)
        if self.decode_head2_ema is not None:
            momentum_update_param_and_buffer(self.decode_head2, self.decode_head2_ema, self.ema_m)
        losses["loss"] = loss
        self.train_iter += 1
        self.last = {"masks": p, "res_fw": res_fw, "res_bw": res_bw, "flows": flows,
                     "loss_flow": loss_flow, "feats": feats, "logits": logits}
        return losses

    def forward_eval(self, imgs):
        B, I, C3, H, W = imgs.shape
        img_3 = imgs.view(B * I, C3, H, W)
        if self.eval_on_ema:
            logits = self.decode_head2_ema(self.backbone2_ema(img_3))
        else:
            logits = self.decode_head2(self.backbone2(img_3))
        p = F.softmax(logits.view(B, I, self.mask_layer, *logits.shape[-2:]), dim=2)
        return p[:, 0]

    def forward(self, x):
        imgs = torch.stack(x["imgs"], dim=1)
        if self.training:
            pl = torch.stack(x["pl_masks"], dim=1) if self.w_pl > 0 else None
            return self.forward_train(imgs, torch.stack(x["gt_fw_flows"], dim=1),
                                      torch.stack(x["gt_bw_flows"], dim=1), pl)
        return self.forward_eval(imgs)


_REGISTRY.update(ResNet=ResNet, FCNHead=FCNHead, CompactnessHead=CompactnessHead, CRFHead=CRFHead,
                 FlowAggregationHeadWithResidual=FlowAggregationHeadWithResidual)


# ----------------------------------------------------------------------------- trainer bits

def poly_lr_factor(epoch, epochs, power, base_lr, min_lr):
    """main.py:294-297."""
    return ((base_lr - min_lr) * (1 - epoch / epochs) ** power + min_lr) / base_lr


def make_optimizer(model, lr, weight_decay):
    """main.py:299-307 — torch Adam with coupled weight decay over requires_grad params."""
    return torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=lr,
                            weight_decay=weight_decay)


# ----------------------------------------------------------------------------- warp (AMD callers)

def flow_warp(x, flow12, pad="border"):
    """utils/warp_utils.py:84-94 (+ mesh_grid :8-14, norm_grid :17-24)."""
    B, _, H, W = x.shape
    xs = torch.arange(W, dtype=x.dtype).view(1, 1, W).expand(B, H, W)
    ys = torch.arange(H, dtype=x.dtype).view(1, H, 1).expand(B, H, W)
    g = torch.stack([xs, ys], 1) + flow12
    gn = torch.stack([2.0 * g[:, 0] / (W - 1) - 1.0, 2.0 * g[:, 1] / (H - 1) - 1.0], dim=-1)
    return F.grid_sample(x, gn, mode="bilinear", padding_mode=pad, align_corners=True)


def corresponding_map(data):
    """utils/warp_utils.py:27-81 — forward-splat of bilinear weights, out-of-range corners dropped."""
    B, _, H, W = data.shape
    x, y = data[:, 0].reshape(B, -1), data[:, 1].reshape(B, -1)
    x1, y1 = torch.floor(x), torch.floor(y)
    xf, yf = x1.clamp(0, W - 1), y1.clamp(0, H - 1)
    x0, y0 = x1 + 1, y1 + 1
    xc, yc = x0.clamp(0, W - 1), y0.clamp(0, H - 1)
    xco, yco, xfo, yfo = x0 != xc, y0 != yc, x1 != xf, y1 != yf
    invalid = torch.cat([xco | yco, xco | yfo, xfo | yco, xfo | yfo], dim=1)
    idx = torch.cat([xc + yc * W, xc + yf * W, xf + yc * W, xf + yf * W], 1).long()
    val = torch.cat([(1 - (x - xc).abs()) * (1 - (y - yc).abs()), (1 - (x - xc).abs()) * (1 - (y - yf).abs()),
                     (1 - (x - xf).abs()) * (1 - (y - yc).abs()), (1 - (x - xf).abs()) * (1 - (y - yf).abs())], 1)
    val = torch.where(invalid, torch.zeros_like(val), val)
    out = torch.zeros(B, H * W, dtype=data.dtype).scatter_add_(1, idx, val)
    return out.view(B, 1, H, W)


def occu_mask_backward(flow21, th=0.2):
    """utils/warp_utils.py:107-113."""
    B, _, H, W = flow21.shape
    xs = torch.arange(W, dtype=flow21.dtype).view(1, 1, W).expand(B, H, W)
    ys = torch.arange(H, dtype=flow21.dtype).view(1, H, 1).expand(B, H, W)
    corr = corresponding_map(torch.stack([xs, ys], 1) + flow21)
    return (corr.clamp(min=0., max=1.) < th).float()


def occu_mask_bidirection(flow12, flow21, scale=0.01, bias=0.5):
    """utils/warp_utils.py:97-104."""
    f21w = flow_warp(flow21, flow12, pad="zeros")
    diff = flow12 + f21w
    mag = (flow12 * flow12).sum(1, keepdim=True) + (f21w * f21w).sum(1, keepdim=True)
    return ((diff * diff).sum(1, keepdim=True) > scale * mag + bias).float()


def ssim_dist(x, y, md=1):
    """models/amd/loss_blocks.py:46-65."""
    k = 2 * md + 1
    ap = lambda t: F.avg_pool2d(t, k, 1, 0)
    mx, my = ap(x), ap(y)
    sx, sy, sxy = ap(x * x) - mx * mx, ap(y * y) - my * my, ap(x * y) - mx * my
    n = (2 * mx * my + 0.01 ** 2) * (2 * sxy + 0.03 ** 2)
    d = (mx * mx + my * my + 0.01 ** 2) * (sx + sy + 0.03 ** 2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def photometric_loss(im, recon, occ, w_l1=0.15, w_ssim=0.85, ssim_sz=1):
    """models/amd/flow_loss.py:15-29 with the weights of models/fcn_head.py:73-85."""
    terms = []
    if w_l1 > 0:
        terms.append(w_l1 * (im - recon).abs() * occ)
    if w_ssim > 0:
        terms.append(w_ssim * ssim_dist(recon * occ, im * occ, ssim_sz))
    return sum(t.mean() for t in terms) / occ.mean()
