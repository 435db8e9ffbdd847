"""Evaluation loop of the reference's LightningModule (main.py:180-292) over the HIP model: per-frame foreground IoU of
the predicted masks against the annotation, the object-channel vote, per-sequence / overall mIoU.

`Evaluator` keeps main.py's state and method names (`on_test_start`, `test_step`, `test_epoch_end`, `iou_all_sequences`,
`max_channel_freq`, `object_channel`); the arithmetic -- resize to the annotation size with align_corners=True
(utils/eval_utils.py:5-12), threshold `eval_pos_th` or hard arg-max one-hot (main.py:209-219), 255 -> 1 / 128 -> ignore
labels (:221-224), intersect_and_union (utils/eval_utils.py:14-52,120-123) -- runs in ONE device kernel per batch
(csrc/evalmetrics.hip) that returns integer counts; the IoUs are the same integer ratios the reference's numpy code forms.
"""
import numpy as np
import torch

from . import _lib
from .ops import _p, _stream


def iou_counts(masks, ann, pos_th):
    """masks [B,C,h,w] fp32 (softmax), ann [B,H,W] uint8 -> int64 [B,C,3] on the host: (intersection, prediction area,
    label area) of the foreground over the non-ignored pixels"""
    if not masks.is_cuda:
        raise _lib.RcfHipError("rcf_amd.evaluate needs CUDA (HIP) tensors: there is no CPU fallback")
    masks = masks.contiguous().float()
    ann = ann.to(masks.device)
    if ann.dtype != torch.uint8:
        ann = ann.round().clamp(0, 255).to(torch.uint8)       # annotations are 0 / 128 / 255 images
    ann = ann.contiguous()
    B, C, h, w = masks.shape
    assert ann.shape[0] == B
    H, W = ann.shape[1:3]
    counts = torch.zeros((B, C, 3), dtype=torch.int64, device=masks.device)
    _lib.call("rcf_eval_iou_counts_f32", _p(masks), _p(ann), B, C, h, w, H, W, float(pos_th), _p(counts), _stream())
    return counts.cpu().numpy()


def iou_from_counts(c):
    """foreground IoU = I / (P + L - I) in float64; 0 / 0 = nan, like numpy's `area_intersect / area_union`"""
    inter, union = c[..., 0].astype(np.float64), (c[..., 1] + c[..., 2] - c[..., 0]).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / union


class Evaluator:
    def __init__(self, args, mask_layer):
        self.args = args
        self.mask_layer = mask_layer
        self.object_channel = getattr(args, "object_channel", None)          # main.py:152
        self.on_test_start()

    def on_test_start(self):
        """main.py:193-196"""
        self.iou_all_sequences = {}
        self.max_channel_freq = [0 for _ in range(self.mask_layer)]

    def test_step(self, model, batch, always_use_max_iou_channel=False):
        """main.py:198-235 (validation_step is the same call, :184-188)"""
        model.eval()
        pred = model(batch)
        assert len(pred) == len(batch["ann"]), f"{len(pred)} != {len(batch['ann'])}"
        ious = iou_from_counts(iou_counts(pred, batch["ann"], self.args.eval_pos_th))       # [B, C]
        for frame_ious, seq_name in zip(ious, batch["seq_names"]):
            if always_use_max_iou_channel or self.object_channel is None:
                max_channel = int(np.argmax(frame_ious))
                self.max_channel_freq[max_channel] += 1
                frame_iou = frame_ious[max_channel]
            else:
                frame_iou = frame_ious[self.object_channel]
            self.iou_all_sequences.setdefault(seq_name, []).append(frame_iou)
        return ious

    def test_epoch_end(self, current_epoch=0, testing=True, sanity_checking=False):
        """main.py:237-292: fixes the object channel by majority vote once, then per-sequence mIoU (nan-mean over the
        frames), their mean, and the frame average.  Returns (mean over sequences, frame average, per-sequence dict)."""
        if self.object_channel is None and not sanity_checking and (
                current_epoch >= getattr(self.args, "set_object_channel_after_epoch", 1) - 1 or testing):
            rank = getattr(self.args, "rank", -1)
            if rank >= 0:                                         # distributed: rank 0's vote is broadcast as a sum
                import torch.distributed as dist
                oc = torch.tensor([int(np.argmax(self.max_channel_freq)) if rank == 0 else 0], device="cuda")
                dist.all_reduce(oc, op=dist.ReduceOp.SUM)
                oc = int(oc.item())
            else:
                oc = int(np.argmax(self.max_channel_freq))
            self.object_channel = oc
            self.args.object_channel = oc                         # shared with the model (models/rcf_model.py:64-66)
        miou_each_sequence, iou_sum, iou_num_frames = {}, 0.0, 0.0
        for seq_name, seq in self.iou_all_sequences.items():
            miou_each_sequence[seq_name] = np.nanmean(seq).astype(np.float32)
            iou_sum += np.sum(seq).astype(np.float32)
            iou_num_frames += len(seq)
        mean_miou = np.mean(list(miou_each_sequence.values())).astype(np.float32)
        return mean_miou, iou_sum / iou_num_frames, miou_each_sequence
