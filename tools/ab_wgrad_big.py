"""A/B of the fp16-pair weight gradient's tile: 128 x 256 on two workgroups per CU against 256 x 256 on one
(igemm_wgrad_h2t_kernel<.., MR = 4>).  Same process, interleaved; results compared (bit-identical when the split counts
agree).  usage: python tools/ab_wgrad_big.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import ops

# name, Cin, Cout, k, pad, dil, H, W, N
SHAPES = [
    ("layer4.conv2 3x3d4 512->512", 512, 512, 3, 4, 4, 60, 107, 16),
    ("layer3.conv2 3x3d2 256->256", 256, 256, 3, 2, 2, 60, 107, 16),
    ("dh2 coarse 3x3d3 2048->256", 2048, 256, 3, 3, 3, 60, 107, 16),
    ("dh3 coarse 3x3d6 4096->256", 4096, 256, 3, 6, 6, 60, 107, 8),
    ("dh2.convs.1 3x3d6 256->256 @120x214", 256, 256, 3, 6, 6, 120, 214, 16),
    ("layer4.conv1 1x1 2048->512", 2048, 512, 1, 0, 1, 60, 107, 16),
    ("layer4.conv3 1x1 512->2048", 512, 2048, 1, 0, 1, 60, 107, 16),
    ("layer3.conv1 1x1 1024->256", 1024, 256, 1, 0, 1, 60, 107, 16),
    ("layer3.conv3 1x1 256->1024", 256, 1024, 1, 0, 1, 60, 107, 16),
    ("layer4.ds 1x1 1024->2048", 1024, 2048, 1, 0, 1, 60, 107, 16),
    ("layer3.0.conv1 1x1 512->256", 512, 256, 1, 0, 1, 60, 107, 16),
]


def timeit(fn, iters=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = "cuda:0"
    tot = {0: 0.0, 1: 0.0}
    totb = {0: 0.0, 1: 0.0}
    for name, Cin, Cout, k, pad, dil, H, W, N in SHAPES:
        x = torch.randn(N, H, W, Cin, device=dev)
        dy = torch.randn(N, H, W, Cout, device=dev)
        w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        ax, ag = ops.absmax(x), ops.absmax(dy)
        xb, dyb = x.bfloat16(), dy.bfloat16()
        flops = 2.0 * N * H * W * Cout * Cin * k * k
        res = {}
        for mode in (0, 1, 0, 1):
            ops.conv_set_wgrad_big(3 * mode)
            dw = torch.zeros_like(w)
            t = timeit(lambda: ops.conv2d_wgrad(x, dy, w, dw, 1, pad, dil, beta=0, amax=(ax, ag)))
            d32 = dw.clone()
            tb = timeit(lambda: ops.conv2d_wgrad_bf16(xb, dyb, w, dw, 1, pad, dil, beta=0))
            res.setdefault(mode, []).append((t, d32, tb, dw.clone()))
        ops.conv_set_wgrad_big(0)
        t0, t1 = min(r[0] for r in res[0]), min(r[0] for r in res[1])
        b0, b1 = min(r[2] for r in res[0]), min(r[2] for r in res[1])
        tot[0] += t0
        tot[1] += t1
        totb[0] += b0
        totb[1] += b1
        d = float((res[0][-1][1] - res[1][-1][1]).abs().max() / res[0][-1][1].abs().max())
        db = float((res[0][-1][3] - res[1][-1][3]).abs().max() / res[0][-1][3].abs().max())
        print(f"{name:38s} N={N} fp16 pairs 128x256: {t0*1e3:7.3f} ms {flops/t0/1e12:6.1f} TF/s | 256x256: {t1*1e3:7.3f} ms {flops/t1/1e12:6.1f} TF/s "
              f"({t0/t1:4.2f}x) diff {d:.1e} || bf16: {b0*1e3:7.3f} ms {flops/b0/1e12:7.1f} | {b1*1e3:7.3f} ms {flops/b1/1e12:7.1f} ({b0/b1:4.2f}x) diff {db:.1e}", flush=True)
    print(f"sum over the shapes: fp16 pairs {tot[0]*1e3:.3f} -> {tot[1]*1e3:.3f} ms, bf16 {totb[0]*1e3:.3f} -> {totb[1]*1e3:.3f} ms")


if __name__ == "__main__":
    main()
