"""Timing experiment: are the short-K 1x1 convs bound by the epilogue's store pattern?  Forward 1x1 convs of the step with the
normal epilogue and with lane-linear stores of the same bytes (rcf_conv_set_variant(0x100008): results are garbage).
usage: python tools/bench_epilogue.py"""
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
    for Cin, Cout in ((256, 1024), (512, 2048), (1024, 256), (2048, 512), (1024, 2048)):
        x = torch.randn(N, H, W, Cin, device=dev)
        w = (torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        ax, aw = ops.absmax(x), ops.absmax(ops.weight_rsck(w))
        wp = ops.weight_pairs(w, aw)
        y = torch.empty(N, H, W, Cout, device=dev)
        flops = 2.0 * N * H * W * Cout * Cin
        mb = (x.numel() + y.numel()) * 4 / 1e6
        out = []
        for rnd in range(2):
            for name, var in (("normal", -1), ("lane-linear stores", 0x100008), ("no global loads", 0x8008), ("both", 0x108008)):
                _lib.load().rcf_conv_set_variant(var)
                t = timeit(lambda: ops.conv2d_fwd(x, w, None, 1, 0, 1, out=y, amax=(ax, aw), w_pairs=wp))
                out.append(f"{name} {t * 1e3:.3f} ms")
            _lib.load().rcf_conv_set_variant(-1)
        print(f"1x1 {Cin}->{Cout}: {flops / 1e9:.0f} GF, {mb:.0f} MB in+out ({mb / 5e3:.3f} ms at 5 TB/s) | " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
