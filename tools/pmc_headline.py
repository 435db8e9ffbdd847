"""The shipped headline conv kernels on three bottleneck layer shapes at 16 frames of 60x107, on RANDOM or on ALL-ZERO operands
(same launches, same code path): for tools/pmc_headline.sh, which puts MFMA-busy / clock / L2 counters of both runs side by side
(DESIGN.md section 4.1's "power-limited" statement: the all-zero run does the same instructions on data that toggles nothing).
fp32 step: conv_h2d_kernel<4,false,false> (forward), <4,false,true> (data gradient), igemm_wgrad_h2d_kernel<4,...> on fp16 pair
planes; bf16 step: conv_bf16_kernel<2,4,...> forward / data gradient, wgrad_bf16_dma_kernel<4,...>.
usage: python tools/pmc_headline.py <random|zero> [launches] [bf16|fp16]   (the 16-bit kernels from librcf_hip.so or librcf_hip_f16.so)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa: E402
from rcf_amd import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [("layer3.conv2 3x3d2 256->256", 256, 256, 3, 2, 2), ("layer3.conv3 1x1 256->1024", 256, 1024, 1, 0, 1),
          ("layer3.conv1 1x1 1024->256", 1024, 256, 1, 0, 1), ("layer4.conv2 3x3d4 512->512", 512, 512, 3, 4, 4)]
N, H, W = 16, 60, 107


def to_planes(x, bound_bits):
    b = float(bound_bits.view(torch.float32))
    k = 14 - int(np.floor(np.log2(b)))
    t = x.double() * 2.0 ** k
    h = t.to(torch.float16)
    m = (t - h.double()).to(torch.float16)
    n, hh, ww, c = x.shape
    return torch.stack([h, m], dim=3).contiguous().view(torch.float32).reshape(n, hh, ww, c)


def main():
    mode = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    half = torch.float16 if (len(sys.argv) > 3 and sys.argv[3] == "fp16") else torch.bfloat16
    g = torch.Generator().manual_seed(3)
    for name, cin, cout, k, pad, dil in SHAPES:
        x = torch.randn(N, H, W, cin, generator=g).to(DEV)
        dy = torch.randn(N, H, W, cout, generator=g).to(DEV)
        w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
        ax, aw, ag = ops.absmax(x), ops.absmax(ops.weight_rsck(w)), ops.absmax(dy)      # the ranges of the RANDOM data in both modes
        if mode == "zero":
            x, dy, w = torch.zeros_like(x), torch.zeros_like(dy), torch.zeros_like(w)
        wp, wpt = ops.weight_pairs(w, aw), ops.weight_pairs_t(w, aw)
        xp, dyp = to_planes(x, ax), to_planes(dy, ag)
        y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.zeros_like(w)
        xb, dyb = x.to(half), dy.to(half)
        with ops.half_storage(half):
            wb, wbt = ops.weight_bf16(w), ops.weight_bf16(w, transpose=True)
        yb, dxb = torch.empty_like(dyb), torch.empty_like(xb)
        for _ in range(reps):
            ops.conv2d_fwd_stats(xp, w, 1, pad, dil, amax=(ax, aw), w_pairs=wp, x_planes=True)
            ops.conv2d_dgrad(dyp, w, x.shape, 1, pad, dil, out=dx, amax=(ag, aw), w_pairs_t=wpt, dy_planes=True)
            ops.conv2d_wgrad(xp, dyp, w, dw, 1, pad, dil, beta=0, amax=(ax, ag), planes=True)
            with ops.half_storage(half):
                ops.conv2d_fwd_bf16(xb, w, wb, None, 1, pad, dil, out=yb)
                ops.conv2d_dgrad_bf16(dyb, w, xb.shape, 1, pad, dil, out=dxb, w_t_bf16=wbt)
                ops.conv2d_wgrad_bf16(xb, dyb, w, dw, 1, pad, dil, beta=0)
        torch.cuda.synchronize()
    print("done", mode)


if __name__ == "__main__":
    main()
