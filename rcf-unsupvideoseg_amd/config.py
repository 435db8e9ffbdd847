"""YAML config loading with `base_config` inheritance and dotted `--opts` overrides -- the part of
utils/utils.py:8-16,36-65,83-148 the hot path needs (duplicate keys rejected, lists replaced,
override type coerced from the existing value)."""
import argparse
import os

import yaml


class _UniqueKeyLoader(yaml.SafeLoader):
    def construct_mapping(self, node, deep=False):
        seen = set()
        for key_node, _ in node.value:
            key = self.construct_object(key_node, deep=deep)
            if key in seen:
                raise ValueError(f"Duplicate {key!r} key found in YAML.")
            seen.add(key)
        return super().construct_mapping(node, deep)


def _merge(base, new):
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            _merge(base[k], v)
        else:
            base[k] = v
    return base


def load_yaml(path):
    with open(path) as f:
        cfg = yaml.load(f, Loader=_UniqueKeyLoader) or {}
    parent = cfg.pop("base_config", None)
    if parent:
        if not os.path.isabs(parent) and not os.path.exists(parent):
            parent = os.path.join(os.path.dirname(path), parent)
        cfg = _merge(load_yaml(parent), cfg)
    return cfg


def apply_opts(cfg, opts):
    assert len(opts) % 2 == 0, f"{len(opts)} should be even"
    for key, value in zip(opts[::2], opts[1::2]):
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node[p]
        old = node[parts[-1]]
        if isinstance(old, bool):
            if value in ("True", "true", "1"):
                value = True
            elif value in ("False", "false", "0"):
                value = False
            else:
                raise ValueError(f"Unknown bool value for {key}: {value}")
        elif isinstance(old, int):
            value = int(value)
        elif isinstance(old, float):
            value = float(value)
        assert type(old) == type(value), f"{type(old)} != {type(value)}"
        node[parts[-1]] = value
    return cfg


def load_args(path, cli_opts=()):
    return argparse.Namespace(**apply_opts(load_yaml(path), list(cli_opts)))


def stage1_model_kwargs(mask_size=(120, 214), mask_layer=4, dropout=0.1, affine=False, norm="SyncBN"):
    """`model_kwargs` of configs/rcf/rcf_stage1.yaml:63-148 with the mask size of the 480x854 runs
    (SURVEY.md F1).  affine=True gives the STv2/FBMS variant (free_residual_with_affine)."""
    ncfg = dict(type=norm, requires_grad=True)
    return dict(
        w_seg=1.0, w_sharpen=0, w_entropy=0.05, separate_residual=True, mask_layer=mask_layer, align_corners=False,
        mask_size=list(mask_size),
        backbone2=dict(type="ResNet", depth=50, num_stages=4, out_indices=[0, 1, 2, 3], dilations=[1, 1, 2, 4],
                       strides=[1, 2, 1, 1], norm_cfg=dict(ncfg), norm_eval=False, style="pytorch",
                       contract_dilation=True),
        decode_head=dict(type="FlowAggregationHeadWithResidual", ssim_sz=1, create_flownet=True,
                         mask_layer=mask_layer, flow_feat_before_agg_kernel_size=3, num_flow_feat_channels=64,
                         mask_size=list(mask_size), norm_flow=False, clamp_flow_t=20., free_residual=not affine,
                         free_residual_with_affine=affine, free_scale=False, outlier_robust_loss=False, eps=0.01,
                         q=0.4, allow_residual_resize=True, residual_adjustment_scale=10., pred_div_coeff=10.),
        decode_head2=dict(type="FCNHead", input_transform="resize_concat", concat_input=False, dilation=6,
                          channels=256, in_channels=[256, 2048], in_index=[0, 3], num_convs=2, dropout_ratio=dropout,
                          num_classes=mask_layer, norm_cfg=dict(ncfg), align_corners=False,
                          loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0)),
        decode_head3=dict(type="FCNHead", concat_input=False, dilation=6, channels=256, in_channels=4096, in_index=-1,
                          num_convs=2, dropout_ratio=dropout, num_classes=4 * mask_layer, norm_cfg=dict(ncfg),
                          align_corners=False,
                          loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0)))


def stage21_model_kwargs(mask_size=(120, 214), dropout=0.1, norm="SyncBN", refine_iters=5):
    """configs/rcf/rcf_stage2.1.yaml on top of stage 1: CRF self-labels from the EMA teacher (w_crf 10, pos/neg weights
    2/1, ema 0.999); `refine_iters` 5 is BASELINE configs[3] (the reference default is 50)"""
    kw = stage1_model_kwargs(mask_size, dropout=dropout, norm=norm)
    kw.update(w_entropy=0, w_crf=10.0, crf_use_ema=True, ema_m=0.999, crf_pos_weight=2.0, crf_neg_weight=1.0,
              crf_head=dict(type="CRFHead", refine_iters=refine_iters))
    kw["backbone2"]["create_ema"] = True
    kw["decode_head2"]["create_ema"] = True
    return kw


def stage22_model_kwargs(mask_size=(120, 214), dropout=0.1, norm="SyncBN"):
    """configs/rcf/rcf_stage2.2.yaml:63-78 on top of stage 1: pseudo-label loss (w_seg 0.1, w_pl 2, pos/neg weights 2/1)
    against the exported stage-2.1 masks; the EMA copies are still created (and updated) but feed no loss"""
    kw = stage1_model_kwargs(mask_size, dropout=dropout, norm=norm)
    kw.update(w_seg=0.1, w_entropy=0, w_pl=2.0, pl_pos_weight=2.0, pl_neg_weight=1.0)
    kw["backbone2"]["create_ema"] = True
    kw["decode_head2"]["create_ema"] = True
    return kw


def mask_size_for(H, W):
    """spatial size after the 7x7/2 stem and the 3x3/2 max-pool."""
    h1, w1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    return ((h1 - 1) // 2 + 1, (w1 - 1) // 2 + 1)


VARIANTS = ("fbms", "stv2", "sharpen_kl", "sharpen_obj", "joint", "compact_obj", "mask_resize", "freeze")
# stage 2.1 / 2.2 (models/rcf_model.py:380-408,490-529): these also need pl_masks in the batch / a CRF binding
STAGE2_VARIANTS = ("stage21", "stage22", "stage22_soft")


def variant_model_kwargs(name, H, W, norm="BN"):
    """Small-geometry stage-1 variants used by the parity fixtures/tests; returns (model_kwargs, object_channel).
    fbms: configs/rcf_fbms59/rcf_stage1.yaml (3 segments + affine);  stv2: configs/rcf_stv2/rcf_stage1.yaml
    (single-map head at 1/8 resolution, affine, compactness);  sharpen_kl / sharpen_obj: the two branches of
    get_sharpen_loss (models/rcf_model.py:350-374);  joint: pred_joint_residual (:337-348);  compact_obj:
    compactness on the object channel (compact_channel -1, models/compactness_head.py:19-24)."""
    ms = mask_size_for(H, W)
    oc = None
    if name == "fbms":
        kw = stage1_model_kwargs(ms, mask_layer=3, dropout=0.0, affine=True, norm=norm)
    elif name == "stv2":
        h8 = ((ms[0] - 1) // 2 + 1, (ms[1] - 1) // 2 + 1)
        kw = stage1_model_kwargs(h8, dropout=0.0, affine=True, norm=norm)
        kw["decode_head"]["allow_residual_resize"] = False
        kw["decode_head2"].update(in_channels=2048, in_index=3)
        kw["decode_head2"].pop("input_transform")
        kw.update(compactness_head=dict(type="CompactnessHead", compact_channel=0), w_compactness=1.0)
    elif name in ("sharpen_kl", "sharpen_obj"):
        kw = stage1_model_kwargs(ms, dropout=0.0, norm=norm)
        kw.update(w_sharpen=0.1, w_entropy=0, t_sharpen=0.25, object_aware_sharpening=(name == "sharpen_obj"))
        oc = 1 if name == "sharpen_obj" else None
    elif name == "joint":
        kw = stage1_model_kwargs(ms, dropout=0.0, norm=norm)
        kw["separate_residual"] = False
        kw["decode_head3"]["num_classes"] = 2 * kw["mask_layer"]
    elif name == "compact_obj":
        kw = stage1_model_kwargs(ms, dropout=0.0, norm=norm)
        kw.update(compactness_head=dict(type="CompactnessHead", compact_channel=-1), w_compactness=0.5)
        oc = 2
    elif name == "mask_resize":
        # allow_mask_resize (models/rcf_model.py:421-422): the flow head works at a mask size the decode head does not
        # produce (here 3/4 of it), the logits are bilinearly resized first
        small = (ms[0] * 3 // 4, ms[1] * 3 // 4)
        kw = stage1_model_kwargs(small, dropout=0.0, norm=norm)
        kw["allow_mask_resize"] = True
    elif name == "stage21":
        # configs/rcf/rcf_stage2.1.yaml: CRF self-labels from the EMA teacher; 5 mean-field iterations (BASELINE configs[3])
        kw = stage21_model_kwargs(ms, dropout=0.0, norm=norm, refine_iters=5)
        oc = 1
    elif name in ("stage22", "stage22_soft"):
        # configs/rcf/rcf_stage2.2.yaml: pseudo-label loss, thresholded at pl_mask_pos_th (default 0.35) or soft (-1)
        kw = stage22_model_kwargs(ms, dropout=0.0, norm=norm)
        if name == "stage22_soft":
            kw["pl_mask_pos_th"] = -1
        oc = 2
    elif name == "freeze":
        kw = stage1_model_kwargs(ms, dropout=0.0, norm=norm)                 # freeze_backbone (models/rcf_model.py:107-110)
        kw["freeze_backbone"] = True
    else:
        raise KeyError(name)
    kw.update(log_interval=10 ** 9, train_iter=1)
    return kw, oc


# ------------------------------------------------------------------------------------------------------------------- schedule
class Schedule:
    """Every choice of HOW the training step runs (never of WHAT it computes: all settings give the same results up to the
    re-association of fp32 sums, most of them bit-identical -- tests/test_pipeline_gpu.py, test_planes_gpu.py, test_fold_gpu.py).
    One object, `rcf_amd.config.SCHED`, read at call time by layers / model / trainer; tools and tests set its attributes
    explicitly (`config.SCHED.overlap_wgrad = False`, `tools/step_prof.py bf16 4 8 fold_bn=0`).  No environment variable feeds it:
    the only environment overrides the package reads are RCF_CONV_FLAGS (per-launch kernel A/B bits, ops.py) and
    RCF_DEBUG_WEIGHT_CACHE.

    streams
      overlap_wgrad      weight gradients on a second HIP stream (joined before anything reads a parameter gradient)
      late_wgrad         ... started AFTER the layer's data gradient, so that a weight gradient (MFMA-bound) runs beside the NEXT
                         layer's batch-norm backward (HBM-bound); the overlapped launches take the 128 x 256 tile
      side_priority      HIP priority of the second stream (0 default, -1 high, 1 low)
      overlap_teacher    stage 2.1: the EMA teacher's forward + CRF on the second stream beside the student's forward
    batch norm
      fuse_bn_stats      conv -> training-mode norm: per-channel sums from the conv epilogue instead of a pass over its output
      fuse_bn_finalize   the reduction of those sums also finalizes the norm (one launch, not four)
      fuse_bn_bwd        fp32 step: the norm's backward sums from the epilogue of the data gradient that writes its output
                         gradient last (57 -> 6 reduction passes; built, measured, does not pay: off)
      relu_bitmask       norm + ReLU keeps the sign bits of its output (1/16 of its bytes) for the backward pass
      defer_residual     fp32 step: a join does not write its identity branch's gradient; conv1's data gradient adds it
      lazy_downsample_norm  a stage's downsample norm has no apply pass: the join that adds it normalises the raw conv output on
                         the way in (bit-identical; one read and one write of the 4C-wide tensor less per stage)
      merge_downsample_bwd  ... and in the backward pass that norm and the join share ONE reduction and ONE apply pass over the join's
                         output gradient and sign bits (bit-identical to the separate passes)
      fold_bn            bf16 step: 1x1 conv -> norm (-> + residual -> ReLU) as ONE tile, statistics from the Gram matrix of the
                         conv's input, algebraic backward: neither the conv output nor its gradient exists (layers.conv_bn_fold)
      fold_max_k         ... for convs with at most this many input channels (the Gram matrix costs 2 rows K^2 FLOPs)
      fold_masked_dgrad  ... the join's ReLU mask + column sums from the epilogue of the LAST writer of its gradient
    operands
      fp16_pairs         fp32 step: convs as 3 fp16 partial products (operand ranges known) instead of 6 bf16 ones
      h2_kinds           which conv directions may do so: "f" forward, "d" data gradient, "w" weight gradient (debug)
      planes             fp32 step: the norms between the bottlenecks' convs write fp16 pair planes, those convs take both operands
                         by LDS-DMA
      join_planes        which joins are ALSO written as planes: "stage" (in front of a stage's first block) or "all"
      nt_stores          forward / data-gradient tile kernels with at least this many output columns write their output with
                         streaming (non-temporal) stores, RCF_CONV_NT_STORES: the tiles' output does not push the weights and the
                         neighbouring column tiles' activation rows out of the XCD's L2 (0 = never; results bit-identical)
      bf16_stem          bf16 step: the stem conv on the bf16 kernels too (image padded to 8 channels)
      autocast_fp16_as_bf16  a model called under torch.autocast(float16) with no explicit precision stores bf16 (rounds 2-5: no loss
                         scaling needed) instead of fp16 (Lightning `precision: 16` as the reference runs it, with its GradScaler)
      cache_weight_operands  derived weight operands once per weight update, not per launch
      bulk_weight_prep   ... for all weights of the model in two / three launches right after the optimizer step (trainer.WeightPrep)
    data parallel
      grad_group         the gradient chunks' all-reduces on their OWN communicator (beside the default one that carries SyncBN):
                         "auto" (default) = try it -- new_group + one probe all-reduce in Trainer.__init__, every rank agreeing on
                         the outcome through the default group -- and share the default communicator if any rank failed;
                         True = own communicator or raise; False = always share.  On a shared communicator the 10-60 MB
                         asynchronous chunks and the SyncBN exchanges queue on ONE RCCL stream in issue order, so every
                         statistics exchange issued after a chunk waits for it (Trainer.grad_group_mode says what ran)
      teacher_group      stage 2.1: the EMA teacher's SyncBN exchanges on their own communicator (default False: a third
                         communicator only pays in stage 2.1 and no multi-GPU box has run one yet)."""

    __slots__ = ("overlap_wgrad", "late_wgrad", "side_priority", "overlap_teacher", "fuse_bn_stats", "fuse_bn_finalize",
                 "fuse_bn_bwd", "relu_bitmask", "defer_residual", "lazy_downsample_norm", "merge_downsample_bwd", "fold_bn", "fold_max_k", "fold_masked_dgrad", "fp16_pairs",
                 "h2_kinds", "planes", "join_planes", "nt_stores", "bf16_stem", "autocast_fp16_as_bf16", "cache_weight_operands", "bulk_weight_prep", "grad_group",
                 "teacher_group")

    def __init__(self):
        self.overlap_wgrad = self.late_wgrad = self.overlap_teacher = True
        self.side_priority = 0
        self.fuse_bn_stats = self.fuse_bn_finalize = self.relu_bitmask = self.defer_residual = self.lazy_downsample_norm = True
        self.merge_downsample_bwd = True
        self.fuse_bn_bwd = False
        self.fold_bn, self.fold_max_k, self.fold_masked_dgrad = True, 512, True
        self.fp16_pairs, self.h2_kinds, self.planes, self.join_planes = True, "fdw", True, "stage"
        self.bf16_stem = self.cache_weight_operands = self.bulk_weight_prep = True
        self.grad_group, self.teacher_group = "auto", False
        self.nt_stores = 0
        self.autocast_fp16_as_bf16 = False

    def set(self, **kw):
        """set several fields; returns the previous values (for a `finally: SCHED.set(**old)`)"""
        old = {k: getattr(self, k) for k in kw}
        for k, v in kw.items():
            setattr(self, k, v)
        return old

    def parse(self, items):
        """`name=value` strings (command-line tails of the tools): booleans as 0 / 1, integers, or bare strings"""
        for it in items:
            k, v = it.split("=", 1)
            cur = getattr(self, k)
            setattr(self, k, (v not in ("0", "false", "False")) if isinstance(cur, bool) else (int(v) if isinstance(cur, int) else v))

    def as_dict(self):
        return {k: getattr(self, k) for k in self.__slots__}


SCHED = Schedule()
