"""Does the Infinity Cache (256 MB, memory side) keep what a kernel just WROTE, and does a streaming kernel that reads it right
afterwards run faster?  For each tensor size: (a) fill x (a write-only kernel), then bn_apply(x) -> y timed alone; (b) the same
with a 1.5 GB write-only fill between the two (evicts whatever the cache held); (c) x filled, then read in the order it was
written vs (d) after an unrelated read of 1.5 GB.  usage: python tools/mall_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import ops

dev = "cuda:0"
rows = 16 * 60 * 107
big = torch.empty(384 * 1024 * 1024, dtype=torch.float32, device=dev)       # 1.5 GB
for C in (64, 128, 256, 512, 1024, 2048):
    x = torch.empty(16, 60, 107, C, device=dev)
    y = torch.empty_like(x)
    mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    res = {}
    for mode in ("fresh", "evicted", "fresh", "evicted"):
        ts = []
        for _ in range(6):
            x.fill_(1.5)
            if mode == "evicted":
                big.fill_(0.5)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.bn_apply(x, mean, invstd, gamma, beta, True, out=y)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        res.setdefault(mode, []).append(min(ts[1:]))
    mb = rows * C * 4 / 1e6
    f, e = min(res["fresh"]), min(res["evicted"])
    print(f"x {mb:7.1f} MB (C = {C:4d}): bn_apply right after the fill {f*1e6:7.1f} us = {2*mb/f/1e6:5.2f} TB/s | after a 1.5 GB fill in between "
          f"{e*1e6:7.1f} us = {2*mb/e/1e6:5.2f} TB/s | ratio {e/f:4.2f}", flush=True)
