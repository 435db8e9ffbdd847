"""The step's conv kernels against the vendor GEMM on the SAME GEMM shapes, in ONE process and interleaved (boxes and thermal
states differ by more than 10 %): torch.matmul (hipBLASLt) in bf16 on [M,K] x [N,K]^T, and this repo's implicit-GEMM conv
(forward) with M = 16 x 60 x 107 pixels, K = R*S*Cin, N = Cout."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import rcf_amd
from rcf_amd import ops

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

layers = [("layer4.conv2 3x3 d4 512->512", 512, 512, 3, 4), ("layer4.conv3 1x1 512->2048", 512, 2048, 1, 1), ("layer4.conv1 1x1 2048->512", 2048, 512, 1, 1),
          ("layer3.conv2 3x3 d2 256->256", 256, 256, 3, 2), ("layer3.conv3 1x1 256->1024", 256, 1024, 1, 1), ("layer3.conv1 1x1 1024->256", 1024, 256, 1, 1),
          ("dh2 coarse 3x3 d3 2048->256", 2048, 256, 3, 3)]
N_, H, W = 16, 60, 107
M = N_ * H * W
for rep in range(2):
    for name, cin, cout, k, dil in layers:
        K = k * k * cin
        x = torch.randn(N_, H, W, cin, device="cuda:0").to(torch.bfloat16)
        w = torch.randn(cout, cin, k, k, device="cuda:0").contiguous(memory_format=torch.channels_last)
        wb = ops.weight_bf16(w)
        a = torch.randn(M, K, device="cuda:0", dtype=torch.bfloat16); b = torch.randn(cout, K, device="cuda:0", dtype=torch.bfloat16)
        pad = dil * (k // 2)
        res = []
        for _ in range(2):                                   # interleaved
            res.append((timeit(lambda: ops.conv2d_fwd_bf16(x, w, wb, None, 1, pad, dil)), timeit(lambda: torch.matmul(a, b.t()))))
        tc, tg = min(r[0] for r in res), min(r[1] for r in res)
        fl = 2.0 * M * K * cout
        print(f"pass {rep} {name:30s} M={M} K={K:5d} N={cout:4d}: conv {fl/tc/1e12:5.0f} TF/s | hipBLASLt GEMM {fl/tg/1e12:5.0f} TF/s | ratio {tg/tc:.2f}", flush=True)
a = torch.randn(8192, 8192, device="cuda:0", dtype=torch.bfloat16); b = torch.randn(8192, 8192, device="cuda:0", dtype=torch.bfloat16)
t = timeit(lambda: torch.matmul(a, b.t()))
print(f"8192^3 GEMM: hipBLASLt {2*8192**3/t/1e12:.0f} TF/s")
