"""RCFModel on the HIP tape -- drop-in for the reference's `models.RCFModel`
(models/rcf_model.py:28-153 ctor, :410-611 forward_train, :275-320 forward_eval, :350-408 losses,
:613-626 forward): same constructor keywords (the YAML under `model_kwargs`), same component
registry by `type` name, same state-dict keys, same return contract (dict of 0-dim loss tensors in
training mode whose 'loss' entry supports `.backward()`, softmax masks [B,C,h,w] in eval mode).

The evaluation side effects are reproduced (eval JPEG, `pred_seg_*.png` export, :241-273,:296-315, through
export.py's stand-in for torchvision.utils.save_image).  Not reproduced: the flow-colour training
visualisations every `log_interval` steps (:456-462,:562-608; they need flow_vis and are off the arithmetic path).
"""
import os
from copy import deepcopy

import torch
import torch.nn as nn

from . import ops
from .export import save_image
from .backbone import FCNHead, ResNet
from .crf import CRFHead
from .flow_head import CompactnessHead, FlowAggregationHeadWithResidual
from . import layers
from .layers import Act, DistCtx, Tape, concat_channels, pair_concat, resize_act

REGISTRY = dict(ResNet=ResNet, FCNHead=FCNHead, FlowAggregationHeadWithResidual=FlowAggregationHeadWithResidual,
                CompactnessHead=CompactnessHead, CRFHead=CRFHead)


def _rank():
    import torch.distributed as dist                            # rank_zero_only of the reference
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _item(v):
    return v.item() if torch.is_tensor(v) else v


class _TapeBackward(torch.autograd.Function):
    """Bridges `losses['loss'].backward()` (what main.py / Lightning call) to the HIP tape.

    The trainable parameters are INPUTS of this node and their gradients are its outputs, so autograd's own
    AccumulateGrad nodes run for every parameter -- which is where torch's DistributedDataParallel hangs its
    bucket hooks (Lightning strategy `ddp_find_unused_parameters_false`, main.py:453-455): under the unchanged
    main.py the gradients are all-reduced by DDP, every parameter receives one every step, and
    torch.optim.Adam (main.py:299-307) sees ordinary `.grad` tensors."""

    @staticmethod
    def forward(ctx, loss_value, model, *params):
        ctx.model, ctx.n = model, len(params)
        return loss_value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        model = ctx.model
        params = [p for p in model.parameters() if p.requires_grad]
        assert len(params) == ctx.n, "the set of trainable parameters changed between forward and backward"
        # the tape accumulates into p.grad: point every p.grad at a view of one fresh zero-filled flat buffer for the
        # duration of the tape, hand those views to autograd, and put the user's .grad tensors back
        sizes = [(p.numel() + 63) // 64 * 64 for p in params]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=params[0].device)
        saved, views, off = [], [], 0
        for p, n in zip(params, sizes):
            chunk = flat[off:off + p.numel()]
            if p.dim() == 4 and p.permute(0, 2, 3, 1).is_contiguous():      # channels_last conv weight
                co, ci, r, s = p.shape
                v = chunk.view(co, r, s, ci).permute(0, 3, 1, 2)
            else:
                v = chunk.view(p.shape)
            saved.append(p.grad)
            p.grad = v
            views.append(v)
            off += n
        try:
            model.run_backward(grad_out)
        finally:
            for p, g in zip(params, saved):
                p.grad = g
        return (None, None) + tuple(views)


@torch.no_grad()
def copy_param_and_buffer(src, dest):
    """utils/model_utils.py:12-19"""
    s, d = src.state_dict(), dest.state_dict()
    assert list(s.keys()) == list(d.keys())
    for k in s:
        d[k].data.copy_(s[k])
    ops.weights_changed()            # `.data` writes do not bump `_version`: drop every cached weight operand


EMA_CHUNK = 1 << 16          # elements per workgroup of rcf_ema_update_multi


class _EmaPlan:
    """the chunk table of one (source, EMA copy) pair for rcf_ema_update_multi: built once, valid while no tensor of either module
    is re-allocated (`key`: every tensor's address)"""

    def __init__(self, pairs, key, device):
        rows, self.others = [], []
        for d, s in pairs:
            kind = 0 if d.dtype == torch.float32 else (1 if d.dtype == torch.int64 else -1)
            same_order = (d.is_contiguous() and s.is_contiguous()) or (
                d.dim() == 4 and d.stride() == s.stride() and d.permute(0, 2, 3, 1).is_contiguous())   # channels_last conv weights
            if kind < 0 or s.dtype != d.dtype or not d.is_cuda or not same_order or d.numel() == 0:
                self.others.append((d, s))
                continue
            n, size = d.numel(), d.element_size()
            for o in range(0, n, EMA_CHUNK):
                rows.append((d.data_ptr() + o * size, s.data_ptr() + o * size, min(EMA_CHUNK, n - o), kind))
        self.key, self.count = key, len(rows)
        self.table = torch.tensor(rows, dtype=torch.int64).to(device) if rows else None


def _state_tensors(mod):
    return list(mod.parameters()) + list(mod.buffers())


@torch.no_grad()
def momentum_update_param_and_buffer(src, dest, m):
    """utils/model_utils.py:33-38: dest = dest*m + src*(1-m) over every state-dict entry -- in ONE launch (rcf_ema_update_multi: the
    float entries and the int64 num_batches_tracked counters with the reference's truncation); until round 6 one launch per float
    entry and four torch kernels per counter: ~500 launches of 2-4 us that the host enqueues more slowly than the GPU runs them."""
    st, dt = _state_tensors(src), _state_tensors(dest)
    key = tuple((t.data_ptr(), t.numel()) for t in st) + tuple((t.data_ptr(), t.numel()) for t in dt)
    plan = getattr(dest, "_ema_plan", None)
    if plan is None or plan.key != key:
        s, d = src.state_dict(), dest.state_dict()
        plan = _EmaPlan([(d[k], s[k]) for k in s], key, dt[0].device if dt else torch.device("cpu"))
        object.__setattr__(dest, "_ema_plan", plan)              # (not a submodule / buffer: stays out of the state dict)
    if plan.table is not None:
        ops.ema_update_multi(plan.table, plan.count, m)
    for d, s in plan.others:
        d.data.copy_(d.data * m + s.data * (1.0 - m))
    ops.weights_changed(dest)            # only `dest` was written: the source's cached operands stay valid


class RCFModel(nn.Module):
    def __init__(self, args, backbone2, decode_head, decode_head2, decode_head3, compactness_head=None,
                 crf_head=None, crf_use_ema=False, ema_m=0.999, w_seg=2.0, w_sharpen=0, t_sharpen=0.25, w_entropy=0,
                 w_compactness=0, w_pl=0, pl_pos_weight=1., pl_neg_weight=1., pl_mask_pos_th=0.35, w_crf=0,
                 crf_pos_weight=1., crf_neg_weight=1., crf_mask_pos_th=-1., mask_layer=1, train_iter=0, train_cfg=None,
                 test_cfg=None, align_corners=False, mask_size=(48, 48), log_interval=50, freeze_backbone=False,
                 object_aware_sharpening=False, separate_residual=False, allow_mask_resize=False):
        super().__init__()
        self.args = args
        ckpt = getattr(args, "checkpoints_dir", ".")
        self.save_dir = os.path.join(ckpt, "saved")
        self.save_dir_eval = os.path.join(ckpt, getattr(args, "saved_eval_dir_name", "saved_eval"))
        self.save_dir_eval_export = os.path.join(ckpt, getattr(args, "saved_eval_export_dir_name", "saved_eval_export"))

        def build(cfg, **extra):
            # like the reference (rcf_model.py:161-197) this pops 'type'/'create_ema' from the caller's dict
            ema = cfg.pop("create_ema", False)
            mod = REGISTRY[cfg.pop("type")](**extra, **cfg)
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
        self.w_compactness, self.compactness_head = w_compactness, None
        if compactness_head:
            self.compactness_head, _ = build(compactness_head, args=args)
            assert w_compactness != 0, "Compactness head is used but weight is 0"
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.backbone2.init_weights()
        if freeze_backbone:
            for p in self.backbone2.parameters():
                p.requires_grad_(False)
        self.train_iter = train_iter
        self.w_seg, self.w_sharpen, self.t_sharpen, self.w_entropy = w_seg, w_sharpen, t_sharpen, w_entropy
        assert not (w_sharpen != 0 and w_entropy != 0), "Only one of w_entropy and w_sharpen could be nonzero"
        self.w_pl = w_pl
        if w_pl > 0:
            assert args.object_channel is not None, "Pseudo label loss requires an object channel"
        self.pl_pos_weight, self.pl_neg_weight, self.pl_mask_pos_th = pl_pos_weight, pl_neg_weight, pl_mask_pos_th
        self.w_crf, self.crf_head = w_crf, None
        if crf_head:
            self.crf_head, _ = build(crf_head, args=args)
            assert w_crf != 0, "CRF head is used but weight is 0"
        self.crf_pos_weight, self.crf_neg_weight, self.crf_mask_pos_th = crf_pos_weight, crf_neg_weight, crf_mask_pos_th
        self.crf_use_ema, self.ema_m, self.log_interval = crf_use_ema, ema_m, log_interval
        self.mask_size, self.allow_mask_resize = tuple(mask_size), allow_mask_resize
        self.object_aware_sharpening, self.separate_residual = object_aware_sharpening, separate_residual
        self.eval_on_ema = getattr(args, "eval_on_ema", False)
        if self.eval_on_ema:
            assert self.backbone2_ema is not None and self.decode_head2_ema is not None
        if self.backbone2_ema is not None:
            copy_param_and_buffer(self.backbone2, self.backbone2_ema)
        if self.decode_head2_ema is not None:
            copy_param_and_buffer(self.decode_head2, self.decode_head2_ema)
        self._tape = None
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: ops.weights_changed())
        # "fp32" | "bf16" | None.  None = follow torch autocast: the reference's AMP configs (configs/rcf_stv2/rcf_stage1.yaml:57-60,
        # Lightning `precision: 16`) call the model inside torch.autocast -- that selects the bf16 step here.
        self.precision = None
        self.dist = None
        self.grad_ready_hook = None      # callable(group) set by the trainer: "heads", "layer4" ... "layer1", "stem"

    # ------------------------------------------------------------------ plumbing
    # train() keeps the reference's semantics (it has no override): the EMA copies are set to eval mode once, at
    # construction (models/rcf_model.py:171,187); `model.train()` -- Lightning calls it on the whole module tree --
    # switches them to training mode with everything else, so under main.py the stage-2.1 teacher runs with batch
    # statistics.  The override below only invalidates the cached weight operands (fp16 planes, bf16 copies, ranges):
    # they are keyed by (data_ptr, _version, epoch) and an in-place write through `.data` bumps none of those, so every
    # mode SWITCH (not the train() call a trainer repeats every step) and every load_state_dict starts from fresh operands.  Anything else that writes parameters through
    # `.data` / raw pointers after a forward must call `rcf_amd.ops.weights_changed()` itself.

    def train(self, mode=True):
        if bool(mode) != self.training:                     # a transition (trainers call train() at the top of every step)
            ops.weights_changed()
        return super().train(mode)

    def _dist(self):
        if self.dist is None:
            self.dist = DistCtx()
        return self.dist

    def _teacher_dist(self):
        """the EMA teacher's SyncBN exchanges on their OWN communicator: its forward is enqueued (on the side stream)
        ahead of the student's, and on one communicator every statistics all-reduce of the student would queue behind
        all of the teacher's -- the two forwards would serialise across ranks.  The group is created EAGERLY by the trainer
        (`Trainer.__init__`, next to its gradient group: `new_group` is a collective of its own, not something to hide inside
        a forward) and only on request, SCHED.teacher_group: no multi-GPU run of this repo has exercised three concurrent RCCL
        communicators yet, so the default shares the student's communicator (correct, slower in stage 2.1)."""
        d = self._dist()
        t = getattr(self, "_tdist", None)
        return t if (d.on and t is not None) else d

    def make_teacher_group(self):
        """collective: every rank calls it at the same point (Trainer.__init__)"""
        if self._dist().on and self.w_crf > 0 and self.crf_use_ema and layers.SCHED.teacher_group:
            import torch.distributed as tdist
            self._tdist = DistCtx(group=tdist.new_group())

    def _select_precision(self):
        """"fp32" | "bf16" | "fp16": the storage type of the activations between layers (weights, gradients and accumulation stay
        fp32).  Unset, torch autocast decides as Lightning's `precision:` does: autocast(float16) -- `precision: 16`,
        configs/rcf_stv2/rcf_stage1.yaml:57-60, to be used with a GradScaler like there -- stores fp16, autocast(bfloat16) bf16
        (SCHED.autocast_fp16_as_bf16 = True restores rounds 2-5: any autocast as bf16 storage, no loss scaling needed)."""
        p = self.precision
        if p is None:
            if torch.is_autocast_enabled():
                fp16 = torch.get_autocast_dtype("cuda") == torch.float16 and not layers.SCHED.autocast_fp16_as_bf16
                p = "fp16" if fp16 else "bf16"
            else:
                p = "fp32"
        if p not in ("fp32", "bf16", "fp16"):
            raise ValueError(f"precision must be 'fp32', 'bf16' or 'fp16', got {p!r}")
        # of THIS forward pass: every Tape it makes carries it
        self._act_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[p]
        return p

    def _images_nhwc(self, imgs):
        B, I, C3, H, W = imgs.shape
        x = imgs.reshape(B * I, C3, H, W).contiguous().float()
        if not x.is_cuda:
            raise RuntimeError("RCFModel (HIP) needs the batch on the GPU: there is no CPU fallback")
        if self._act_dtype in ops.H16 and layers.SCHED.bf16_stem:
            # 16-bit step: the stem conv takes 16-bit operands like every other conv (torch autocast casts conv1's input too)
            return Act(ops.cast(ops.nchw_to_nhwc(x, 8), self._act_dtype), needs_grad=False)
        return Act(ops.nchw_to_nhwc(x, 4), needs_grad=False)

    def run_backward(self, grad_out=None):
        """Backward of the last forward_train: fills param.grad (accumulating)."""
        if self._tape is None:
            raise RuntimeError("backward called twice or before forward")
        tape, self._tape = self._tape, None
        scale = 1.0 if grad_out is None else float(grad_out)
        with ops.half_storage(tape.act_dtype):               # the loss tail's backward launches kernels too
            self._seed_backward(scale)
            tape.backward()

    # ------------------------------------------------------------------ training forward
    def forward_train(self, imgs, gt_fw_flows, gt_bw_flows, pl_masks=None):
        B, I = imgs.shape[:2]
        dist = self._dist()
        self._select_precision()
        tape = Tape(on_mark=self.grad_ready_hook, act_dtype=self._act_dtype)
        img = self._images_nhwc(imgs)
        crf_side = None
        if self.w_crf > 0 and self.crf_use_ema and layers.SCHED.overlap_teacher:
            # the EMA teacher's forward + CRF need only the images: run them on the second stream beside the student's
            # forward (its HBM-bound BN passes and the teacher's MFMA-bound convs fill each other's gaps)
            crf_side = layers._side_stream(img.t.device)
            if layers.SCHED.fp16_pairs and img.t.dtype == torch.float32:
                img.range()      # on THIS stream, before the fork: teacher and student stems share the cached range
            ops.reserve_amax(img.t.device, 512)     # the teacher's range slots: zero-filled before the fork as well
            crf_side.wait_stream(torch.cuda.current_stream(img.t.device))
            with torch.cuda.stream(crf_side):
                crf_early = self._crf_targets(img, imgs, None, B, I)
        if any(q.requires_grad for q in self.backbone2.parameters()):
            feats = self.backbone2.fwd(img, tape, dist)
        else:                                            # freeze_backbone: nothing behind the features needs a gradient
            feats = self.backbone2.fwd(img, Tape(enabled=False, act_dtype=self._act_dtype), dist)
            for f in feats:
                f.needs_grad = False
        tape.mark("heads")                                                       # fires once all three heads are done
        logits = self.decode_head2.fwd(feats, tape, dist)                       # Act [B*I,h,w,C]
        if self.allow_mask_resize and tuple(logits.t.shape[1:3]) != self.mask_size:
            logits = resize_act(logits, tape, self.mask_size, self.align_corners)          # rcf_model.py:421-422
        if self.separate_residual:
            res = self.decode_head3.fwd([pair_concat(feats[-1], tape, B, I)], tape, dist)   # [B,h2,w2,4C]
        else:
            # pred_joint_residual (rcf_model.py:337-348): the same head on (f0,f1) and on (f1,f0); each run
            # gives 2C channels (fw resp. bw); laid side by side they form the [B,h2,w2,4C] residual
            nc2 = self.decode_head3.num_classes
            r_fw = self.decode_head3.fwd([pair_concat(feats[-1], tape, B, I)], tape, dist)
            r_bw = self.decode_head3.fwd([pair_concat(feats[-1], tape, B, I, order=(1, 0))], tape, dist)
            res = concat_channels([r_fw, r_bw], tape, widths=[nc2, nc2])
        # ground-truth flows to mask resolution (values not rescaled, rcf_model.py:438-442)
        nf = gt_fw_flows.shape[1]
        gfw = ops.resize_nchw(gt_fw_flows.reshape(B * nf, *gt_fw_flows.shape[2:]).contiguous().float(),
                              self.mask_size, self.align_corners)
        gbw = ops.resize_nchw(gt_bw_flows.reshape(B * nf, *gt_bw_flows.shape[2:]).contiguous().float(),
                              self.mask_size, self.align_corners)
        extra = {}
        if self.w_pl > 0:
            extra["pl_masks"] = ops.resize_nchw(pl_masks.contiguous().float(), self.mask_size, self.align_corners)
        if self.w_crf > 0:
            if crf_side is not None:
                torch.cuda.current_stream(img.t.device).wait_stream(crf_side)
                extra["crf_masks"] = crf_early
            else:
                extra["crf_masks"] = self._crf_targets(img, imgs, logits, B, I)
        losses, seed = self.decode_head.loss_and_grads(
            self, logits, res, gfw.view(B, nf, 2, *self.mask_size), gbw.view(B, nf, 2, *self.mask_size), extra, B, I,
            act_dtype=self._act_dtype)
        self._seed_backward = seed
        self._tape = tape
        self.last_targets = extra        # the pl / crf targets at mask size that entered the loss (tests look at them)
        self.last_logits = logits.t      # NHWC fp32 logits of decode_head2 (kept alive by the tape anyway)
        if self.backbone2_ema is not None:
            momentum_update_param_and_buffer(self.backbone2, self.backbone2_ema, self.ema_m)
        if self.decode_head2_ema is not None:
            momentum_update_param_and_buffer(self.decode_head2, self.decode_head2_ema, self.ema_m)
        self.train_iter += 1
        if torch.is_grad_enabled():
            losses["loss"] = _TapeBackward.apply(losses["loss"], self, *[p for p in self.parameters() if p.requires_grad])
        return losses

    @torch.no_grad()
    def _crf_targets(self, img_act, imgs, logits, B, I):
        """rcf_model.py:496-520: (EMA) masks -> object channel -> image size -> CRF -> mask size."""
        oc = self.args.object_channel
        if self.crf_use_ema:
            t = Tape(enabled=False, act_dtype=self._act_dtype)
            d = self._teacher_dist()     # in training mode (after model.train()) the teacher's SyncBN exchanges statistics too
            le = self.decode_head2_ema.fwd(self.backbone2_ema.fwd(img_act, t, d), t, d)
        else:
            le = logits
        h, w = le.t.shape[1:3]
        p = self.decode_head.softmax_masks(le.t, B, I)                           # [B,I,C,h,w]
        H, W = imgs.shape[-2:]
        obj = p.flatten(0, 1)[:, oc:oc + 1].contiguous()
        up = ops.resize_nchw(obj, (H, W), self.align_corners)[:, 0]
        crf = self.crf_head(imgs.reshape(B * I, *imgs.shape[2:]), up)            # [B*I,H,W] float 0/1
        return ops.resize_nchw(crf.view(B, I, H, W).contiguous(), self.mask_size, self.align_corners)

    # ------------------------------------------------------------------ eval forward
    def resize(self, x, size):
        """mmseg.ops.resize(bilinear, align_corners=self.align_corners) on NCHW (rcf_model.py:208-214)"""
        return ops.resize_nchw(x.contiguous().float(), tuple(size), self.align_corners)

    def save_eval_visualizations(self, tosave, paths, seq_ids, seq_names, name="eval", train_iter=0):
        """rcf_model.py:241-251: one JPEG of the whole batch grid, named after sample 0"""
        if _rank() != 0:
            return
        frame_id = paths[0][0].split("/")[-1][:-4]
        fn = f"{self.save_dir_eval}/{name}_{seq_names[0]}_{_item(seq_ids[0])}_{frame_id}_{train_iter:07}.jpg"
        try:
            save_image(tosave, fn)
        except Exception as e:                                 # the reference logs and carries on
            print(f"Error in saving: {fn} {e}")

    def export_seg(self, tosave, paths, seq_ids, seq_names, name="eval", train_iter=0, subdir=""):
        """rcf_model.py:253-267: one PNG per sample of the batch"""
        if _rank() != 0:
            return
        sub = subdir + "/" if subdir else ""
        for i, (path, seq_name, _seq_id) in enumerate(zip(paths[0], seq_names, seq_ids)):
            frame_id = path.split("/")[-1][:-4]
            fn = f"{self.save_dir_eval_export}/{sub}{name}_{seq_name}_{frame_id}_{train_iter:07}.png"
            try:
                save_image(tosave[i], fn)
            except Exception as e:
                print(f"Error in saving: {fn} {e}")

    def export_all_seg(self, tosave, paths, seq_ids, seq_names, name="eval", train_iter=0):
        for idx, item in enumerate(tosave):                    # rcf_model.py:269-273
            self.export_seg(item, paths, seq_ids, seq_names, name, train_iter, subdir=str(idx))

    @torch.no_grad()
    def forward_eval(self, imgs, seq_ids=None, seq_names=None, paths=None, return_pred_vis_list=False):
        """rcf_model.py:275-320: softmax masks of the current frame [B,C,h,w]; with args.eval_save the 2x
        visualisation grid (and with args.eval_export the object-channel / all-channel PNGs) are written."""
        B, I = imgs.shape[:2]
        self._select_precision()
        t = Tape(enabled=False, act_dtype=self._act_dtype)
        img = self._images_nhwc(imgs)
        if self.eval_on_ema:
            logits = self.decode_head2_ema.fwd(self.backbone2_ema.fwd(img, t), t)
        else:
            logits = self.decode_head2.fwd(self.backbone2.fwd(img, t), t)
        pred_masks = self.decode_head.softmax_masks(logits.t, B, I)[:, 0]
        want_save = bool(getattr(self.args, "eval_save", False))
        if not (want_save or return_pred_vis_list):
            return pred_masks
        h, w = pred_masks.shape[-2:]
        img0 = (self.resize(imgs[:, 0], (h * 2, w * 2)) + 2.0) / 4.0
        vis = [self.resize(pred_masks[:, i:i + 1].contiguous(), (h * 2, w * 2)).repeat(1, 3, 1, 1)
               for i in range(min(self.mask_layer, pred_masks.shape[1]))]
        if want_save:
            self.save_eval_visualizations(torch.cat([img0] + vis, 2), paths, seq_ids, seq_names, train_iter=self.train_iter)
            if getattr(self.args, "eval_export", False):
                if getattr(self.args, "export_all_seg", False):
                    self.export_all_seg(vis, paths, seq_ids, seq_names, name="pred_seg", train_iter=self.train_iter)
                else:
                    self.export_seg(vis[self.args.object_channel], paths, seq_ids, seq_names, name="pred_seg",
                                    train_iter=self.train_iter)
        return (pred_masks, vis) if return_pred_vis_list else pred_masks

    def forward(self, x, return_pred_vis_list=False):
        imgs = torch.stack(x["imgs"], dim=1)
        self._select_precision()
        with ops.half_storage(self._act_dtype):          # the library build that stores this pass's 16-bit type (ops.half_storage)
            if self.training:
                pl = torch.stack(x["pl_masks"], dim=1) if self.w_pl > 0 else None
                return self.forward_train(imgs, torch.stack(x["gt_fw_flows"], dim=1),
                                          torch.stack(x["gt_bw_flows"], dim=1), pl)
            return self.forward_eval(imgs, x.get("seq_ids"), x.get("seq_names"), x.get("paths"), return_pred_vis_list)
