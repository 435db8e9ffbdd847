"""A/B of the exact-2x bilinear resize kernels (csrc/spatial.hip resize2x_*) against the general ones on the decode heads'
tensors.  usage: python tools/bench_resize.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import ops


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for dt in (torch.float32, torch.bfloat16):
    for C in (256, 512):
        x = torch.randn(16, 60, 107, C, device="cuda:0").to(dt)
        dy = torch.randn(16, 120, 214, C, device="cuda:0").to(dt)
        out = {}
        for mode in (0, 1, 0, 1):
            fr = 0 if mode else -1                  # frame = -1: the general kernels on the whole tensor
            tf = timeit(lambda: ops.resize_nhwc_fwd(x, (120, 214), False, frame=fr))
            tb = timeit(lambda: ops.resize_nhwc_bwd(dy, (60, 107), False, frame=fr))
            out.setdefault(mode, []).append((tf, tb))
        f0, b0 = min(t[0] for t in out[0]), min(t[1] for t in out[0])
        f1, b1 = min(t[0] for t in out[1]), min(t[1] for t in out[1])
        mb = dy.numel() * dy.element_size() / 1e6
        print(f"{str(dt):15s} C={C}: forward general {f0*1e6:6.1f} us, exact-2x {f1*1e6:6.1f} us ({f0/f1:4.2f}x; {mb/f1/1e6:4.2f} TB/s of output) | "
              f"backward general {b0*1e6:6.1f} us, exact-2x {b1*1e6:6.1f} us ({b0/b1:4.2f}x)", flush=True)
