"""Micro-benchmark of the bf16-operand conv kernels (csrc/igemm_bf16.hip) on the RCF layer shapes at 480x854.
usage: python tools/bench_conv_bf16.py [frames] [tile,tile,...]   (tile: -1 heuristic, 0 128x128, 1 128x256, 2 256x256, 3 128x64)
-> TF/s per kernel (fwd / dgrad / wgrad) against the dense bf16 MFMA peak (2500 TF/s)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import _lib, ops

# name, Cin, Cout, k, stride, pad, dil, H, W (input), frames divisor (decode_head3 sees pairs, not frames)
SHAPES = [
    ("layer4.conv2 3x3d4 512->512 @60x107", 512, 512, 3, 1, 4, 4, 60, 107, 1),
    ("layer4.conv3 1x1 512->2048 @60x107", 512, 2048, 1, 1, 0, 1, 60, 107, 1),
    ("layer4.conv1 1x1 2048->512 @60x107", 2048, 512, 1, 1, 0, 1, 60, 107, 1),
    ("layer3.conv2 3x3d2 256->256 @60x107", 256, 256, 3, 1, 2, 2, 60, 107, 1),
    ("layer3.conv3 1x1 256->1024 @60x107", 256, 1024, 1, 1, 0, 1, 60, 107, 1),
    ("layer3.conv1 1x1 1024->256 @60x107", 1024, 256, 1, 1, 0, 1, 60, 107, 1),
    ("dh2 coarse 3x3d3 2048->256 @60x107", 2048, 256, 3, 1, 3, 3, 60, 107, 1),
    ("dh2.convs.1 3x3d6 256->256 @120x214", 256, 256, 3, 1, 6, 6, 120, 214, 1),
    ("dh3.convs.0 3x3d6 4096->256 @60x107", 4096, 256, 3, 1, 6, 6, 60, 107, 2),
    ("layer2.conv2 3x3 128->128 @60x107", 128, 128, 3, 1, 1, 1, 60, 107, 1),
    ("layer1.conv2 3x3 64->64 @120x214", 64, 64, 3, 1, 1, 1, 120, 214, 1),
    ("layer1.conv3 1x1 64->256 @120x214", 64, 256, 1, 1, 0, 1, 120, 214, 1),
    ("layer1.conv1 1x1 256->64 @120x214", 256, 64, 1, 1, 0, 1, 120, 214, 1),
]


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


def run(N, only=None):
    dev = "cuda:0"
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    for name, Cin, Cout, k, stride, pad, dil, H, W, div in SHAPES:
        if only and only not in name:
            continue
        n = max(N // div, 1)
        x = torch.randn(n, H, W, Cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        wb = ops.weight_bf16(w)
        y = ops.conv2d_fwd_bf16(x, w, wb, None, stride, pad, dil)
        dy = torch.randn(y.shape, device=dev).to(torch.bfloat16)
        dw = torch.zeros_like(w)
        dx = torch.empty_like(x)
        flops = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * Cout * Cin * k * k
        tf = timeit(lambda: ops.conv2d_fwd_bf16(x, w, wb, None, stride, pad, dil, out=y))
        td = timeit(lambda: ops.conv2d_dgrad_bf16(dy, w, x.shape, stride, pad, dil, out=dx))
        tw = timeit(lambda: ops.conv2d_wgrad_bf16(x, dy, w, dw, stride, pad, dil, beta=0))
        for key, t in (("fwd", tf), ("dgrad", td), ("wgrad", tw)):
            tot[key][0] += flops
            tot[key][1] += t
        print(f"{name:40s} N={n:2d} {flops/1e9:8.1f} GF  fwd {tf*1e3:7.3f} ms {flops/tf/1e12:6.0f} TF/s | "
              f"dgrad {td*1e3:7.3f} ms {flops/td/1e12:6.0f} TF/s | wgrad {tw*1e3:7.3f} ms {flops/tw/1e12:6.0f} TF/s",
              flush=True)
    print("   total: " + " | ".join(f"{k} {v[1]*1e3:7.2f} ms {v[0]/max(v[1],1e-9)/1e12:6.0f} TF/s" for k, v in tot.items()))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    tiles = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1]
    only = sys.argv[3] if len(sys.argv) > 3 else None
    for t in tiles:
        _lib.load().rcf_conv_bf16_set_tile(t)
        print(f"==== bf16 conv tile {t}")
        run(N, only)


if __name__ == "__main__":
    main()
