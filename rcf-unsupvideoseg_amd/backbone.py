"""ResNet backbone (`backbone2`) and FCN decode heads (`decode_head2/3`) on the HIP tape.

Same constructor keywords and state-dict keys as the reference components they replace:
  ResNet      models/resnet.py:371-466 (ctor), :630-645 (forward), :598-628 (init); stage assembly
              models/res_layer.py:26-94 (contract_dilation :66-70); Bottleneck :95-302
  FCNHead     models/fcn_head.py:50-140,142-147,211-218 + models/decode_head.py:45-90,141-170
Only the options the RCF configs use are implemented; anything else raises.
"""
import torch
import torch.nn as nn

from . import layers, ops
from .layers import (Act, BatchNorm2d, Conv2d, commuted_concat_conv, commuted_concat_conv_ok, concat_channels,
                     maxpool3x3s2)


def make_norm(norm_cfg, n):
    t = (norm_cfg or {}).get("type", "BN")
    if t not in ("BN", "SyncBN"):
        raise NotImplementedError(f"norm type {t}")
    # 'BN' keeps per-rank statistics under data parallelism (torch BatchNorm2d inside DDP), 'SyncBN' exchanges them
    return BatchNorm2d(n, requires_grad=bool((norm_cfg or {}).get("requires_grad", True)), sync=(t == "SyncBN"))


class Downsample(nn.Module):
    """nn.Sequential(conv, norm) of models/res_layer.py:53-63 -- keys `0.weight`, `1.*`."""

    def __init__(self, conv, norm):
        super().__init__()
        self.add_module("0", conv)
        self.add_module("1", norm)

    def fwd(self, x, tape, dist, lazy=False):
        conv, norm = getattr(self, "0"), getattr(self, "1")
        if layers.fold_ok(conv, norm, x):
            return layers.conv_bn_fold(conv, norm, x, tape, relu=False, dist=dist)
        return norm.fwd(conv.fwd(x, tape, stats=norm.stats_request(dist)), tape, relu=False, dist=dist, lazy=lazy)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, dilation, downsample, norm_cfg):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1)
        self.bn1 = make_norm(norm_cfg, planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation)
        self.bn2 = make_norm(norm_cfg, planes)
        self.conv3 = Conv2d(planes, planes * 4, 1)
        self.bn3 = make_norm(norm_cfg, planes * 4)
        self.downsample = downsample

    def takes_planes(self, act_dtype=torch.float32):
        """do this block's first convs read their input as fp16 pair planes when the producer supplies them?"""
        ok = self.conv1.planes_ok(act_dtype) and (self.downsample is None or getattr(self.downsample, "0").planes_ok(act_dtype))
        return ok and (layers.SCHED.join_planes == "all" or self.downsample is not None)

    def fwd(self, x, tape, dist, emit_planes=False):
        """emit_planes: the block's output is ALSO written as fp16 pair planes (the next block's conv1 / downsample read them).
        Inside the block the outputs of bn1 / bn2 -- whose only readers are conv2 / conv3 and their weight gradients -- exist as
        pair planes ONLY, and every conv that got its input so receives its output gradient so (layers.BatchNorm2d.fwd)."""
        o1 = self.conv1.fwd(x, tape, stats=self.bn1.stats_request(dist))
        ods = None
        if self.downsample is not None and dist is not None and dist.on and self.bn1.sync and getattr(self.downsample, "1").sync:
            # data parallel: conv1 and the downsample conv read the same input, so their statistics exist together -- ONE
            # all-reduce for both batch norms instead of two latency-bound ones (the single-rank order is left as it was)
            ods = getattr(self.downsample, "0").fwd(x, tape, stats=getattr(self.downsample, "1").stats_request(dist))
            if o1.stats is not None and ods.stats is not None:
                dist.allreduce_sum_many([o1.stats, ods.stats])
                o1.stats_global = ods.stats_global = True
        o = self.bn1.fwd(o1, tape, relu=True, dist=dist, planes="only" if self.conv2.planes_ok(tape.act_dtype) else None)
        o = self.bn2.fwd(self.conv2.fwd(o, tape, stats=self.bn2.stats_request(dist)), tape, relu=True, dist=dist,
                         planes="only" if self.conv3.planes_ok(tape.act_dtype) else None)
        # the downsample norm's apply pass is left to the join (BatchNorm2d.fwd lazy) unless conv3 folds (its tile reads a tensor)
        lazy = not layers.fold_ok(self.conv3, self.bn3, o)
        if ods is not None:
            idt = getattr(self.downsample, "1").fwd(ods, tape, relu=False, dist=dist, lazy=lazy)
        else:
            idt = x if self.downsample is None else self.downsample.fwd(x, tape, dist, lazy=lazy)
        if layers.fold_ok(self.conv3, self.bn3, o, idt):
            # bf16 step: conv3 -> bn3 -> + identity -> ReLU as one tile, conv3's output never written (layers.conv_bn_fold)
            return layers.conv_bn_fold(self.conv3, self.bn3, o, tape, relu=True, residual=idt, dist=dist)
        o = self.conv3.fwd(o, tape, stats=self.bn3.stats_request(dist))
        return self.bn3.fwd(o, tape, relu=True, residual=idt, dist=dist,          # relu(bn3 + identity)
                            planes="both" if emit_planes else None)


class Stage(nn.Module):
    """children named 0..n-1 like the reference's ResLayer(nn.Sequential)."""

    def __init__(self, blocks):
        super().__init__()
        for i, b in enumerate(blocks):
            self.add_module(str(i), b)

    def fwd(self, x, tape, dist, next_takes_planes=False):
        blocks = list(self.children())
        for i, b in enumerate(blocks):
            nxt = blocks[i + 1].takes_planes(tape.act_dtype) if i + 1 < len(blocks) else next_takes_planes
            x = b.fwd(x, tape, dist, emit_planes=nxt)
        return x


class ResNet(nn.Module):
    blocks_per_depth = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth=50, in_channels=3, stem_channels=64, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style="pytorch",
                 norm_cfg=None, norm_eval=False, contract_dilation=False, zero_init_residual=True, **unsupported):
        super().__init__()
        for k, v in unsupported.items():
            if v not in (None, False, -1, (False, False, False, False)):
                raise NotImplementedError(f"ResNet option {k}={v!r} is off the RCF path")
        if style != "pytorch" or depth not in self.blocks_per_depth or norm_eval:
            raise NotImplementedError("only pytorch-style bottleneck ResNets with train-mode norm")
        norm_cfg = norm_cfg or dict(type="BN", requires_grad=True)
        self.out_indices, self.num_stages, self.zero_init_residual = tuple(out_indices), num_stages, zero_init_residual
        self.conv1 = Conv2d(in_channels, stem_channels, 7, stride=2, padding=3)
        self.bn1 = make_norm(norm_cfg, stem_channels)
        inplanes = stem_channels
        for i, nblocks in enumerate(self.blocks_per_depth[depth][:num_stages]):
            planes, stride, dil = base_channels * 2 ** i, strides[i], dilations[i]
            down = None
            if stride != 1 or inplanes != planes * 4:
                down = Downsample(Conv2d(inplanes, planes * 4, 1, stride=stride), make_norm(norm_cfg, planes * 4))
            first = dil // 2 if (dil > 1 and contract_dilation) else dil
            blocks = [Bottleneck(inplanes, planes, stride, first, down, norm_cfg)]
            inplanes = planes * 4
            blocks += [Bottleneck(inplanes, planes, 1, dil, None, norm_cfg) for _ in range(1, nblocks)]
            setattr(self, f"layer{i + 1}", Stage(blocks))
        self.feat_dim = inplanes
        self.heads_take_planes = True       # RCFModel: the last stage's output also as fp16 pair planes (layers.SCHED.planes gates it)

    def init_weights(self, pretrained=None):
        if pretrained is not None:
            raise NotImplementedError("load checkpoints through load_state_dict (main.py:76-144 does)")
        for m in self.modules():
            if isinstance(m, Conv2d):
                nn.init.kaiming_normal_(m.weight, a=0, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def fwd(self, img, tape, dist=None):
        """img: Act of the NHWC image zero-padded to 4 channels; returns the list of stage outputs."""
        tape.mark("stem")
        x = self.bn1.fwd(self.conv1.fwd(img, tape, stats=self.bn1.stats_request(dist)), tape, relu=True, dist=dist)
        x = maxpool3x3s2(x, tape)
        outs = []
        for i in range(self.num_stages):
            tape.mark(f"layer{i + 1}")
            # the last stage's output is read by the decode heads: decode_head3's first conv takes it as planes (through pair_concat)
            nxt = getattr(getattr(self, f"layer{i + 2}"), "0").takes_planes(tape.act_dtype) if i + 1 < self.num_stages else self.heads_take_planes
            x = getattr(self, f"layer{i + 1}").fwd(x, tape, dist, next_takes_planes=nxt)
            if i in self.out_indices:
                outs.append(x)
        return outs

    def forward(self, x):
        """nn.Module surface (`backbone2(img [N,3,H,W]) -> tuple of 4 NCHW maps`), inference only."""
        from .layers import Tape
        img = Act(ops.nchw_to_nhwc(x.contiguous().float(), 4), needs_grad=False)
        outs = self.fwd(img, Tape(enabled=False))
        return tuple(ops.nhwc_to_nchw(o.t) for o in outs)


class ConvModule(nn.Module):
    """mmcv ConvModule as built by models/fcn_head.py:107-130: conv (bias only without norm) -> BN -> ReLU."""

    def __init__(self, cin, cout, k, padding, dilation, stride, norm_cfg):
        super().__init__()
        if norm_cfg is None:
            raise NotImplementedError("ConvModule without a norm layer is off the RCF path")
        self.conv = Conv2d(cin, cout, k, stride=stride, padding=padding, dilation=dilation)
        self.bn = make_norm(norm_cfg, cout)

    def fwd(self, x, tape, dist, chan_scale=None, planes=None):
        """planes: "only" when the next module is a conv that reads fp16 pair planes (layers.BatchNorm2d.fwd)"""
        return self.bn.fwd(self.conv.fwd(x, tape, stats=self.bn.stats_request(dist)), tape, relu=True, chan_scale=chan_scale, dist=dist,
                           planes=planes)


class FCNHead(nn.Module):
    def __init__(self, in_channels, channels, *, num_classes, num_convs=2, kernel_size=3, concat_input=True,
                 dilation=1, input_stride=1, input_dilation=None, dropout_ratio=0.1, norm_cfg=None, in_index=-1,
                 input_transform=None, align_corners=False, transform_scale=None, create_flownet=False,
                 conv_cfg=None, act_cfg=None, loss_decode=None, ignore_index=255, sampler=None, mask_layer=1,
                 ssim_sz=1, load_flownet=False, freeze_flownet=False, flow_model_path=""):
        super().__init__()
        if create_flownet:
            raise NotImplementedError("FCNHead(create_flownet=True) is the AMD baseline (PWC-Lite), not RCF")
        if concat_input or conv_cfg is not None or sampler is not None or num_convs < 1:
            raise NotImplementedError("FCNHead option off the RCF path (concat_input / conv_cfg / sampler)")
        self.input_transform, self.in_index = input_transform, in_index
        if input_transform == "resize_concat":
            in_channels = sum(in_channels)
        elif input_transform is not None:
            raise NotImplementedError(input_transform)
        self.in_channels, self.channels, self.num_classes = in_channels, channels, num_classes
        self.align_corners, self.transform_scale, self.dropout_ratio = align_corners, transform_scale, dropout_ratio
        self.conv_seg = Conv2d(channels, num_classes, 1, bias=True)
        self.conv_seg.out_fp32 = True       # the logits feed the fp32 loss tail also in the bf16 step
        nn.init.kaiming_uniform_(self.conv_seg.weight, a=5 ** 0.5)      # torch Conv2d default: never re-initialised
        nn.init.uniform_(self.conv_seg.bias, -1 / channels ** 0.5, 1 / channels ** 0.5)   # 
        if input_dilation is None:
            input_dilation = dilation
        convs = [ConvModule(in_channels, channels, kernel_size, input_dilation, input_dilation, input_stride, norm_cfg)]
        convs += [ConvModule(channels, channels, kernel_size, dilation, dilation, 1, norm_cfg)
                  for _ in range(num_convs - 1)]
        self.convs = Stage(convs)
        self.keep_mask = None           # tests may inject a Dropout2d keep-mask [N, channels]
        self.commute_upsample = True    # layers.commuted_concat_conv when its conditions hold

    def transform_inputs(self, feats, tape):
        if self.input_transform == "resize_concat":
            sel = [feats[i] for i in self.in_index]
            size = tuple(sel[0].t.shape[1:3])
            if self.transform_scale is not None:
                size = tuple(s * self.transform_scale for s in size)
            return concat_channels(sel, tape, size, self.align_corners)
        return feats[self.in_index]

    def fwd(self, feats, tape, dist=None):
        mods = list(self.convs.children())
        first = None
        if self.commute_upsample and self.input_transform == "resize_concat" and len(self.in_index) == 2 \
                and self.transform_scale is None:
            fa, fb = feats[self.in_index[0]], feats[self.in_index[1]]
            if commuted_concat_conv_ok(fa, fb, mods[0].conv, self.align_corners):
                first = commuted_concat_conv(fa, fb, mods[0].conv, tape)      # conv(concat(a, up2(b))) restructured
        x = self.transform_inputs(feats, tape) if first is None else None
        scale = None
        if self.training and self.dropout_ratio > 0:
            # nn.Dropout2d: one Bernoulli(1-p) per (n, c) plane, kept planes scaled by 1/(1-p)
            if self.keep_mask is not None:
                scale = self.keep_mask
            else:
                # the draw: one launch of the library's own counter-based generator, seeded from torch's HOST generator (so that
                # torch.manual_seed governs it, as it governs the reference's nn.Dropout2d) -- no torch kernels in the step
                src = (first if first is not None else x).t
                seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
                scale = ops.dropout2d_scale(src.shape[0], self.channels, self.dropout_ratio, seed, src.device)
        for i, m in enumerate(mods):
            cs = scale if i == len(mods) - 1 else None
            # a conv module whose successor is another conv module hands its output on as fp16 pair planes only
            pl = "only" if (i + 1 < len(mods) and cs is None and mods[i + 1].conv.planes_ok(tape.act_dtype)) else None
            if i == 0 and first is not None:
                x = m.bn.fwd(first, tape, relu=True, chan_scale=cs, dist=dist, planes=pl)
            else:
                x = m.fwd(x, tape, dist, chan_scale=cs, planes=pl)
        return self.conv_seg.fwd(x, tape)

    def forward(self, inputs):
        """nn.Module surface (list of NCHW maps -> NCHW logits), inference only."""
        from .layers import Tape
        feats = [Act(ops.nchw_to_nhwc(f.contiguous().float()), needs_grad=False) for f in inputs]
        return ops.nhwc_to_nchw(self.fwd(feats, Tape(enabled=False)).t)
