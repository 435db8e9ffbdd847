"""A/B of the weight-gradient kernels' workgroup -> (output tile, pixel range) mapping (csrc/rcf_common.h rcf_wgrad_item):
plain grid order (every XCD's L2 fetches every pixel range) against "an XCD's workgroups share their pixel range".
Same process, interleaved, results compared bit for bit; fp16-pair (fp32 step) and bf16 kernels.
usage: python tools/ab_wgrad_xcd.py [frames]            RCF_AB_ONE=<mode>: one mode only, 3 launches per shape (PMC passes)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import ops

# name, Cin, Cout, k, pad, dil, H, W
SHAPES = [
    ("layer4.conv2 3x3d4 512->512", 512, 512, 3, 4, 4, 60, 107),
    ("layer3.conv2 3x3d2 256->256", 256, 256, 3, 2, 2, 60, 107),
    ("dh2 coarse 3x3d3 2048->256", 2048, 256, 3, 3, 3, 60, 107),
    ("dh2.convs.1 3x3d6 256->256 @120x214", 256, 256, 3, 6, 6, 120, 214),
    ("layer4.conv1 1x1 2048->512", 2048, 512, 1, 0, 1, 60, 107),
    ("layer4.conv3 1x1 512->2048", 512, 2048, 1, 0, 1, 60, 107),
    ("layer3.conv1 1x1 1024->256", 1024, 256, 1, 0, 1, 60, 107),
    ("layer3.conv3 1x1 256->1024", 256, 1024, 1, 0, 1, 60, 107),
    ("layer4.ds 1x1 1024->2048", 1024, 2048, 1, 0, 1, 60, 107),
    ("layer2.conv2 3x3 128->128", 128, 128, 3, 1, 1, 60, 107),
    ("layer1.conv2 3x3 64->64 @120x214", 64, 64, 3, 1, 1, 120, 214),
    ("layer1.conv3 1x1 64->256 @120x214", 64, 256, 1, 0, 1, 120, 214),
]


def timeit(fn, iters=6):
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
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    one = os.environ.get("RCF_AB_ONE")
    dev = "cuda:0"
    tot = {("fp32", 0): 0.0, ("fp32", 1): 0.0, ("bf16", 0): 0.0, ("bf16", 1): 0.0}
    for name, Cin, Cout, k, pad, dil, H, W in SHAPES:
        x = torch.randn(N, H, W, Cin, device=dev)
        dy = torch.randn(N, H, W, Cout, device=dev)
        w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        ax, ag = ops.absmax(x), ops.absmax(dy)
        xb, dyb = x.bfloat16(), dy.bfloat16()
        flops = 2.0 * N * H * W * Cout * Cin * k * k
        if one is not None:
            ops.conv_set_wgrad_xcd(int(one))
            dw = torch.zeros_like(w)
            for _ in range(3):
                ops.conv2d_wgrad(x, dy, w, dw, 1, pad, dil, beta=0, amax=(ax, ag))
                ops.conv2d_wgrad_bf16(xb, dyb, w, dw, 1, pad, dil, beta=0)
            torch.cuda.synchronize()
            continue
        res = {}
        for mode in (0, 1, 0, 1):
            ops.conv_set_wgrad_xcd(mode)
            dw = torch.zeros_like(w)
            t32 = timeit(lambda: ops.conv2d_wgrad(x, dy, w, dw, 1, pad, dil, beta=0, amax=(ax, ag)))
            d32 = dw.clone()
            t16 = timeit(lambda: ops.conv2d_wgrad_bf16(xb, dyb, w, dw, 1, pad, dil, beta=0))
            res.setdefault(mode, []).append((t32, t16, d32, dw.clone()))
        ops.conv_set_wgrad_xcd(1)
        a, b = res[0][-1], res[1][-1]
        t = {(p, m): min(r[i] for r in res[m]) for i, p in ((0, "fp32"), (1, "bf16")) for m in (0, 1)}
        for key in t:
            tot[key] += t[key]
        print(f"{name:38s} N={N} fp16 pairs: grid order {t['fp32', 0]*1e3:7.3f} ms {flops/t['fp32', 0]/1e12:6.1f} TF/s | XCD {t['fp32', 1]*1e3:7.3f} ms "
              f"{flops/t['fp32', 1]/1e12:6.1f} TF/s ({t['fp32', 0]/t['fp32', 1]:4.2f}x) || bf16: {t['bf16', 0]*1e3:7.3f} ms {flops/t['bf16', 0]/1e12:6.1f} | "
              f"{t['bf16', 1]*1e3:7.3f} ms {flops/t['bf16', 1]/1e12:6.1f} ({t['bf16', 0]/t['bf16', 1]:4.2f}x) || identical "
              f"{torch.equal(a[2], b[2])}/{torch.equal(a[3], b[3])}", flush=True)
    if one is None:
        print(f"sum over the shapes: fp16 pairs {tot['fp32', 0]*1e3:.3f} -> {tot['fp32', 1]*1e3:.3f} ms, bf16 {tot['bf16', 0]*1e3:.3f} -> {tot['bf16', 1]*1e3:.3f} ms")


if __name__ == "__main__":
    main()
