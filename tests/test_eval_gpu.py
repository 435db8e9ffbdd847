"""GPU: the evaluation metric (main.py:193-292, utils/eval_utils.py) -- device kernel + rcf_amd.Evaluator against numbers
the reference's own test_step / test_epoch_end produced (tests/golden/eval.json, generator make_golden_eval.py)."""
import json
import os
import types

import numpy as np
import pytest
import torch

import rcf_amd
from rcf_amd import evaluate, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _FixedMasks(torch.nn.Module):
    """stands in for the model: returns the fixture's masks batch by batch (what RCFModel.forward_eval returns)"""

    def __init__(self, masks):
        super().__init__()
        self.masks, self.pos = masks, 0

    def forward(self, batch):
        n = len(batch["ann"])
        m = self.masks[self.pos:self.pos + n]
        self.pos += n
        return m


@pytest.mark.parametrize("case", ["th035_vote", "argmax_vote", "th035_oc2"])
def test_evaluator_vs_reference_golden(case, golden_dir, report):
    fx = json.load(open(os.path.join(golden_dir, "eval.json")))[case]
    masks, ann, names = synth.eval_inputs(fx["seed"])
    args = types.SimpleNamespace(eval_pos_th=fx["eval_pos_th"], rank=-1, object_channel=fx["object_channel_given"],
                                 set_object_channel_after_epoch=1)
    ev = evaluate.Evaluator(args, mask_layer=masks.shape[1])
    model = _FixedMasks(torch.from_numpy(masks).to(DEV))
    for i in range(0, len(ann), 2):
        ev.test_step(model, {"ann": torch.from_numpy(ann[i:i + 2]).to(DEV), "seq_names": names[i:i + 2]})
    # per-frame IoUs are ratios of the same integers: bit-exact
    for k, want in fx["per_frame_iou"].items():
        got = [float(v) for v in ev.iou_all_sequences[k]]
        assert got == want, (k, got, want)
    assert ev.max_channel_freq == fx["max_channel_freq"]
    miou, frame_avg, per_seq = ev.test_epoch_end(current_epoch=0, testing=True)
    assert ev.object_channel == fx["object_channel_after"] and args.object_channel == fx["object_channel_after"]
    assert float(miou) == fx["test_miou"] and abs(float(frame_avg) - fx["test_miou_frame_avg"]) < 1e-7
    for k, v in fx["per_sequence"].items():
        assert float(per_seq[k]) == v
    report(f"evaluator [{case}]: per-frame IoUs, channel vote {ev.max_channel_freq} -> {ev.object_channel}, mIoU {float(miou):.6f} "
           f"identical to the reference's test_step / test_epoch_end")


def test_iou_counts_vs_numpy_fullsize(report):
    """480x854 annotations, masks at 120x214: the kernel's integer counts against a numpy restatement of the same steps"""
    import torch.nn.functional as F
    g = np.random.Generator(np.random.PCG64(3))
    B, C, h, w, H, W = 3, 4, 120, 214, 480, 854
    masks = torch.softmax(torch.from_numpy(g.normal(0, 1.5, size=(B, C, h, w)).astype(np.float32)), dim=1)
    ann = g.choice(np.array([0, 128, 255], dtype=np.uint8), size=(B, H, W), p=[0.6, 0.05, 0.35])
    rs = F.interpolate(masks, size=(H, W), mode="bilinear", align_corners=True)
    for th in (0.35, -1):
        got = evaluate.iou_counts(masks.to(DEV), torch.from_numpy(ann).to(DEV), th)
        if th >= 0:
            safe = (rs - th).abs() > 3e-7
            pred = rs > th
        else:
            top2 = torch.topk(rs, 2, dim=1).values
            safe = ((top2[:, 0] - top2[:, 1]) > 3e-7)[:, None].expand_as(rs)
            pred = F.one_hot(rs.argmax(1), C).permute(0, 3, 1, 2).bool()
        valid = torch.from_numpy(ann != 128)[:, None]
        lab = torch.from_numpy(ann == 255)[:, None]
        want = torch.stack([(pred & lab & valid).sum((2, 3)), (pred & valid).sum((2, 3)),
                            (lab & valid).expand_as(pred).sum((2, 3))], dim=-1).numpy()
        unsafe = int((~safe & valid).sum())
        diff = int(np.abs(got - want).max())
        report(f"iou counts 480x854 th={th}: max |count diff| {diff} (pixels within 3e-7 of a decision: {unsafe})")
        assert diff <= unsafe
