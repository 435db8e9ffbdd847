"""Per-layer table of the conv launches inside the training step (480x854, 8 pairs): every (kernel family, layer shape)
with launches per step, average launch time, achieved TF/s and fraction of its MFMA roofline, the number of workgroup
tiles and rounds on the chip -- once with the weight gradients on the second stream (live) and once with one stream.
usage: python tools/layer_table.py [fp32|bf16] [pairs] [ab]   (ab: one stream only, with the fp16 pair planes on and off: the
per-layer A/B of the LDS-DMA plane kernels against the register-split kernels in one process)"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import config, layers, ops, synth


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    H, W = 480, 854
    dev = torch.device("cuda", 0)
    mask = config.mask_size_for(H, W)
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_bench", object_channel=None, eval_save=False, eval_export=False)
    nb = synth.make_batch(B, H, W, config_id=2)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    batch = {"imgs": [t(x) for x in nb["imgs"]], "gt_fw_flows": [t(x) for x in nb["gt_fw_flows"]],
             "gt_bw_flows": [t(x) for x in nb["gt_bw_flows"]], "seq_ids": nb["seq_ids"], "seq_names": nb["seq_names"],
             "paths": nb["paths"]}
    model = rcf_amd.RCFModel(args, **config.stage1_model_kwargs(mask, dropout=0.1, norm="SyncBN"))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
    tr = rcf_amd.Trainer(model, lr=1e-4, weight_decay=1e-4, device=dev, precision=prec)
    for _ in range(4):
        tr.step(batch)
    ab = len(sys.argv) > 3 and sys.argv[3] == "ab"
    fams = ["conv_h2d_fwd", "conv_h2d_fwd_narrow", "conv_h2d_dgrad", "conv_h2d_dgrad_narrow", "conv_wgrad_h2d", "conv_wgrad_h2d_narrow", "conv_h2p_fwd", "conv_h2p_dgrad", "conv_x3_128x256", "conv_fwd_narrow", "conv_dgrad_wide", "conv_dgrad_other", "conv_wgrad_h2t4", "conv_wgrad_other",
            "conv_bf16_fwd", "conv_bf16_fwd_narrow", "conv_bf16_dgrad_wide", "conv_bf16_dgrad_other", "conv_bf16_wgrad4",
            "conv_bf16_wgrad_other"]
    peak = 2500.0 if prec == "bf16" else 2500.0 / 3
    for overlap, planes in (((False, True), (False, False)) if ab else ((True, layers.SCHED.planes), (False, layers.SCHED.planes))):
        layers.SCHED.overlap_wgrad, layers.SCHED.planes = overlap, planes
        tr.step(batch)
        tr.step(batch)
        nsteps = 3
        ops.PROFILE.start(fams)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(nsteps):
            tr.step(batch)
        e1.record()
        det = ops.PROFILE.stop_detail()
        print(f"==== {prec} step, {B} pairs, weight gradients on the second stream: {overlap}, fp16 pair planes: {planes}; {e0.elapsed_time(e1) / nsteps:.2f} ms/step (bracketed)")
        rows = sorted(det.items(), key=lambda kv: -kv[1]["ms"])
        tot = {}
        for (fam, tag), r in rows:
            tot[fam] = tot.get(fam, 0.0) + r["ms"] / nsteps
        print("family totals (ms/step): " + ", ".join(f"{k} {v:.2f}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1])) +
              f" | sum {sum(tot.values()):.2f}")
        for (fam, tag), r in rows:
            n = r["launches"] // nsteps
            ms = r["ms"] / r["launches"]
            tf = r["flops"] / (r["ms"] * 1e-3) / 1e12
            print(f"{fam:22s} {tag:58s} x{n:2d}  {ms:7.3f} ms  {r['ms'] / nsteps:7.3f} ms/step  {tf:7.1f} TF/s  {tf / peak:5.3f}")
    layers.SCHED.overlap_wgrad = False


if __name__ == "__main__":
    main()
