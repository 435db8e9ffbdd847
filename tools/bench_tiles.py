"""Which tile of igemm_conv_x3_kernel suits the short-K 1x1 convs?  Forward and data gradient of the step's 1x1 shapes with the
tile pinned (rcf_conv_set_variant(0x108 | tile << 4): 0 = 128x128, 1 = 128x256 (default for > 128 columns), 3 = 256x128).
usage: python tools/bench_tiles.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import _lib, ops


def timeit(fn, iters=10):
    for _ in range(2):
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
    N, H, W = 16, 60, 107
    ops.conv_set_h2p(0)
    for Cin, Cout, k, pad, dil in ((256, 1024, 1, 0, 1), (512, 2048, 1, 0, 1), (1024, 256, 1, 0, 1), (2048, 512, 1, 0, 1),
                                   (1024, 2048, 1, 0, 1), (256, 256, 3, 2, 2)):
        x = torch.randn(N, H, W, Cin, device=dev)
        w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        ax, aw = ops.absmax(x), ops.absmax(ops.weight_rsck(w))
        wp, wpt = ops.weight_pairs(w, aw), ops.weight_pairs_t(w, aw)
        y = torch.empty(N, H, W, Cout, device=dev)
        dy = torch.randn(N, H, W, Cout, device=dev)
        ag = ops.absmax(dy)
        dx = torch.empty_like(x)
        flops = 2.0 * N * H * W * Cout * Cin * k * k
        out = []
        for rnd in range(2):
            for name, var in (("128x256", -1), ("128x128", 0x108), ("256x128", 0x138)):
                _lib.load().rcf_conv_set_variant(var)
                tf = timeit(lambda: ops.conv2d_fwd_stats(x, w, 1, pad, dil, amax=(ax, aw), w_pairs=wp))
                td = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, 1, pad, dil, out=dx, beta=1, amax=(ag, aw), w_pairs_t=wpt))
                out.append(f"{name} fwd+stats {tf * 1e3:.3f} dgrad(beta) {td * 1e3:.3f}")
            _lib.load().rcf_conv_set_variant(-1)
        print(f"{k}x{k} {Cin}->{Cout} ({flops / 1e9:.0f} GF): " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
