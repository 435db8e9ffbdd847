"""batch-norm element-wise kernels on the step's largest tensors: GB/s of the bytes they must move"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import rcf_amd
from rcf_amd import ops

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

for dt in (torch.bfloat16, torch.float32):
    for (N, H, W, C) in [(16, 120, 214, 256), (16, 60, 107, 1024), (16, 60, 107, 512)]:
        x = torch.randn(N, H, W, C, device="cuda:0").to(dt)
        res = torch.randn(N, H, W, C, device="cuda:0").to(dt)
        dy = torch.randn(N, H, W, C, device="cuda:0").to(dt)
        mean = torch.zeros(C, device="cuda:0"); invstd = torch.ones(C, device="cuda:0")
        g = torch.ones(C, device="cuda:0"); b = torch.zeros(C, device="cuda:0")
        es = x.element_size(); n = x.numel()
        mask = torch.empty(n // 4, dtype=torch.uint8, device="cuda:0")
        y = torch.empty_like(x)
        t1 = timeit(lambda: ops.bn_apply(x, mean, invstd, g, b, True, residual=res, out=y, relu_mask=mask, out_dtype=dt))
        t0 = timeit(lambda: ops.bn_apply(x, mean, invstd, g, b, True, out=y, relu_mask=mask, out_dtype=dt))
        s2 = None
        t2 = timeit(lambda: ops.bn_bwd_reduce(dy, x, y, mean, invstd, True, relu_mask=mask))
        s2 = ops.bn_bwd_reduce(dy, x, y, mean, invstd, True, relu_mask=mask)
        dg, db = torch.zeros(C, device="cuda:0"), torch.zeros(C, device="cuda:0")
        dx = torch.empty_like(x)
        t3 = timeit(lambda: ops.bn_bwd_apply(dy, x, y, mean, invstd, g, True, s2, N * H * W, dg, db, dx=dx, relu_mask=mask))
        print(f"{str(dt)[6:]:8s} [{N},{H},{W},{C}]: apply {(2*n*es+n//4)/t0/1e12:.2f} TB/s ({t0*1e6:.0f} us) | apply+res {(3*n*es+n//4)/t1/1e12:.2f} ({t1*1e6:.0f} us) | "
              f"bwd reduce {(2*n*es+n//4)/t2/1e12:.2f} ({t2*1e6:.0f} us) | bwd apply {(3*n*es+n//4)/t3/1e12:.2f} ({t3*1e6:.0f} us)")
