"""Same-process A/B of host-side switches on the training step (one thermal state, interleaved rounds).
usage: python tools/ab_step.py [fp32|bf16] [rounds=3] [steps=6]"""
import os, sys, time, types
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import config, synth, layers, ops

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
H, W, B = 480, 854, 8
dev = torch.device("cuda:0")
args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_bench", object_channel=None, eval_save=False, eval_export=False)
model = rcf_amd.RCFModel(args, **config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.1, norm="SyncBN"))
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
tr = rcf_amd.Trainer(model, device=dev, precision=prec)
nb = synth.make_batch(B, H, W, config_id=2)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
batch = {k: [t(x) for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
configs = {"all on": {}, "no operand cache": {"CACHE_WEIGHT_OPERANDS": False}, "no fused finalize": {"FUSE_BN_FINALIZE": False},
           "fp32 stem": {"BF16_STEM": False}, "dma pieces ahead": {"tile": 4},
           "bf16 wgrad on 1 stream": {"OVERLAP_WGRAD_BF16": False}, "main stream high priority": {"hp": True}}
HP = torch.cuda.Stream(device=dev, priority=-1)
if os.environ.get("AB_ONLY"):
    configs = {k: v for k, v in configs.items() if k == "all on" or k in os.environ["AB_ONLY"].split(",")}
FLAGS = ("CACHE_WEIGHT_OPERANDS", "FUSE_BN_FINALIZE", "BF16_STEM")
for _ in range(3):
    tr.step(batch)
res = {k: [] for k in configs}
for r in range(rounds):
    for name, flags in configs.items():
        for k in FLAGS:
            setattr(layers, k, flags.get(k, True))
        rcf_amd._lib.load().rcf_conv_bf16_set_tile(flags.get("tile", -1))
        layers.OVERLAP_WGRAD_BF16 = flags.get("OVERLAP_WGRAD_BF16", True)
        ctx = torch.cuda.stream(HP) if flags.get("hp") else __import__("contextlib").nullcontext()
        torch.cuda.synchronize()
        ctx.__enter__()
        tr.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(batch)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / steps * 1e3)
        fams = ["conv_x3_128x256", "conv_fwd_narrow", "conv_dgrad_wide", "conv_dgrad_other", "conv_wgrad_h2t4", "conv_wgrad_other"] \
            if prec == "fp32" else ["conv_bf16_fwd", "conv_bf16_dgrad_wide", "conv_bf16_wgrad4"]
        ops.PROFILE.start(fams)
        tr.step(batch)
        by = ops.PROFILE.stop()
        torch.cuda.synchronize()
        ctx.__exit__(None, None, None)
        print(f"   round {r} {name:20s} {res[name][-1]:7.2f} ms/step | " + " ".join(f"{k.replace('conv_', '')} {v['ms']:6.2f}" for k, v in by.items()),
              f"| reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
for name, v in res.items():
    print(f"{prec} {name:20s}: " + " ".join(f"{x:7.2f}" for x in v) + f"   median {np.median(v):7.2f} ms/step")
