"""Micro-benchmark of the implicit-GEMM conv kernels on the RCF layer shapes (480x854, one GPU).
usage: python tools/bench_conv.py [frames]   -> prints TF/s per kernel (fwd / dgrad / wgrad)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import ops

# name, Cin, Cout, k, stride, pad, dil, H, W (input), frames-multiplier
SHAPES = [
    ("dh2.convs.0 3x3d6 2304->256 @120x214", 2304, 256, 3, 1, 6, 6, 120, 214),
    ("layer4.conv2 3x3d4 512->512 @60x107", 512, 512, 3, 1, 4, 4, 60, 107),
    ("layer4.conv3 1x1 512->2048 @60x107", 512, 2048, 1, 1, 0, 1, 60, 107),
    ("layer4.conv1 1x1 2048->512 @60x107", 2048, 512, 1, 1, 0, 1, 60, 107),
    ("layer3.conv2 3x3d2 256->256 @60x107", 256, 256, 3, 1, 2, 2, 60, 107),
    ("layer1.conv2 3x3 64->64 @120x214", 64, 64, 3, 1, 1, 1, 120, 214),
    ("layer1.conv3 1x1 64->256 @120x214", 64, 256, 1, 1, 0, 1, 120, 214),
    ("layer1.conv1 1x1 256->64 @120x214", 256, 64, 1, 1, 0, 1, 120, 214),
    ("layer2.conv3 1x1 128->512 @60x107", 128, 512, 1, 1, 0, 1, 60, 107),
    ("stem 7x7s2 4->64 @480x854", 4, 64, 7, 2, 3, 1, 480, 854),
]


PAIRS = os.environ.get("RCF_BENCH_PAIRS", "1") != "0"


def timeit(fn, iters=5):
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
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1]
    for v in variants:
        from rcf_amd import _lib
        _lib.load().rcf_conv_set_variant(v)
        print(f"==== conv variant {v} (bit0: BK=32, bit1: row-major LDS)")
        run(N)


def run(N):
    dev = "cuda:0"
    for name, Cin, Cout, k, stride, pad, dil, H, W in SHAPES:
        x = torch.randn(N, H, W, Cin, device=dev)
        w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        y = ops.conv2d_fwd(x, w, None, stride, pad, dil)
        dy = torch.randn_like(y)
        dw = torch.zeros_like(w)
        flops = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * Cout * Cin * k * k
        if PAIRS:                                   # fp16-pair kernels: operand ranges (+ pre-split weights forward)
            ax, aw, ag = ops.absmax(x), ops.absmax(ops.weight_rsck(w)), ops.absmax(dy)
            wp = ops.weight_pairs(w, aw)
            tf = timeit(lambda: ops.conv2d_fwd(x, w, None, stride, pad, dil, out=y, amax=(ax, aw), w_pairs=wp))
            td = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, stride, pad, dil, out=x, amax=(ag, aw)))
            ax = ops.absmax(x)
            tw = timeit(lambda: ops.conv2d_wgrad(x, dy, w, dw, stride, pad, dil, beta=0, amax=(ax, ag)))
        else:
            tf = timeit(lambda: ops.conv2d_fwd(x, w, None, stride, pad, dil, out=y))
            td = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, stride, pad, dil, out=x))
            tw = timeit(lambda: ops.conv2d_wgrad(x, dy, w, dw, stride, pad, dil, beta=0))
        print(f"{name:44s} N={N} {flops/1e9:8.1f} GF  fwd {tf*1e3:8.3f} ms {flops/tf/1e12:6.1f} TF/s | "
              f"dgrad {td*1e3:8.3f} ms {flops/td/1e12:6.1f} TF/s | wgrad {tw*1e3:8.3f} ms {flops/tw/1e12:6.1f} TF/s",
              flush=True)


if __name__ == "__main__":
    main()
