"""Times rcf_conv2d_wgrad_bf16 alone (one stream, nothing beside it) for the bf16 step's weight-gradient shapes: what plan_wgrad's
split-K choice costs per layer.  usage: python tools/wgrad_probe.py   (profiles/r05_wgrad_split_probe.txt holds a sweep over forced
split counts taken with a temporary override)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import ops

dev = torch.device("cuda", 0)
shapes = [(16, 60, 107, 1024, 256, 1), (16, 60, 107, 2048, 512, 1), (16, 60, 107, 256, 1024, 1), (16, 60, 107, 512, 2048, 1),
          (16, 120, 214, 64, 64, 3), (16, 120, 214, 64, 256, 1), (16, 120, 214, 64, 64, 1), (16, 60, 107, 128, 512, 1),
          (16, 120, 214, 256, 64, 1), (16, 60, 107, 512, 128, 1), (16, 60, 107, 256, 256, 3), (16, 60, 107, 512, 512, 3),
          (16, 60, 107, 256, 256, 1), (16, 60, 107, 512, 512, 1)]


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (N, H, W, cin, cout, k) in shapes:
    x = torch.randn(N, H, W, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, H, W, cout, device=dev).to(torch.bfloat16)
    dw = torch.zeros(cout, k, k, cin, device=dev).permute(0, 3, 1, 2)
    t = timeit(lambda: ops.conv2d_wgrad_bf16(x, dy, dw, dw, 1, k // 2, 1, beta=0))
    flops = 2.0 * N * H * W * cin * cout * k * k
    byt = 2.0 * N * H * W * (cin + cout)
    print(f"{cin}->{cout} k{k} {N}x{H}x{W}: {t * 1e3:.1f} us  {flops / t / 1e9:.0f} TF/s  (additive model {flops / 1.1e15 * 1e6 + byt / 6e12 * 1e6:.1f} us)", flush=True)
