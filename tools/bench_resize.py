#!/usr/bin/env python3
"""bilinear NHWC resize at the step's large tensor ([16,60,107,256] <-> [16,120,214,256]): time and bytes moved"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import rcf_amd
from rcf_amd import ops


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(16, 60, 107, 256, device="cuda:0").to(dt)
    g = torch.randn(16, 120, 214, 256, device="cuda:0").to(dt)
    b = x.element_size()
    tf = timeit(lambda: ops.resize_nhwc_fwd(x, (120, 214), False))
    tb = timeit(lambda: ops.resize_nhwc_bwd(g, (60, 107), False))
    nb = (x.numel() + g.numel()) * b
    y = ops.resize_nhwc_fwd(x, (120, 214), False).float()
    ref = torch.nn.functional.interpolate(x.float().permute(0, 3, 1, 2), size=(120, 214), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    xg = x.float().requires_grad_(True)
    torch.nn.functional.interpolate(xg.permute(0, 3, 1, 2), size=(120, 214), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).backward(g.float())
    gb = ops.resize_nhwc_bwd(g, (60, 107), False).float()
    print(f"{dt}: fwd {tf*1e6:7.1f} us {nb/tf/1e12:.2f} TB/s | bwd {tb*1e6:7.1f} us {nb/tb/1e12:.2f} TB/s | "
          f"fwd err {float((y-ref).abs().max()):.2e} bwd err {float((gb-xg.grad).abs().max()):.2e}")

# frame-restricted forms (the commuted up-sampling of the decode head's dilated conv): only the border frame is written / read
for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(16, 60, 107, 256, device="cuda:0").to(dt)
    g = torch.randn(16, 120, 214, 256, device="cuda:0").to(dt)
    out = torch.zeros(16, 120, 214, 256, device="cuda:0").to(dt)
    gx = torch.zeros(16, 60, 107, 256, device="cuda:0").to(dt)
    for fr in (6, 12):
        tf = timeit(lambda: ops.resize_nhwc_fwd(x, (120, 214), False, out=out, frame=fr))
        tb = timeit(lambda: ops.resize_nhwc_bwd(g, (60, 107), False, out=gx, beta=1, frame=fr))
        print(f"{dt} frame {fr}: fwd {tf*1e6:7.1f} us | bwd (beta=1) {tb*1e6:7.1f} us")
