"""A/B of the persistent LDS-DMA conv kernels (csrc/igemm_h2p.inc: deep 3x3 layers; csrc/igemm_h2s.inc: the 1x1 layers as a
stream of K-steps) against the 128x256 kernel on the step's layer shapes:
same process, interleaved, outputs compared bit for bit.  usage: python tools/bench_h2p.py [frames]"""
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
    ("layer2.conv3 1x1 128->512 (x3 only)", 128, 512, 1, 0, 1, 60, 107),
    ("layer1.conv1 1x1 256->64 @120x214 (x3 only)", 256, 64, 1, 0, 1, 120, 214),
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
    dev = "cuda:0"
    for name, Cin, Cout, k, pad, dil, H, W in SHAPES:
        x = torch.randn(N, H, W, Cin, device=dev)
        w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        ax, aw = ops.absmax(x), ops.absmax(ops.weight_rsck(w))
        ops.conv_set_h2p(1)
        ops.conv_set_h2s(1)
        wp, wpt = ops.weight_pairs(w, aw), ops.weight_pairs_t(w, aw)      # both layouts (built with the kernels forced on)
        y = torch.empty(N, H, W, Cout, device=dev)
        dy = torch.randn(N, H, W, Cout, device=dev)
        ag = ops.absmax(dy)
        dx = torch.empty_like(x)
        flops = 2.0 * N * H * W * Cout * Cin * k * k
        res = {}
        for mode in (0, 1, 0, 1):
            ops.conv_set_h2p(mode)
            ops.conv_set_h2s(mode)
            tf = timeit(lambda: ops.conv2d_fwd(x, w, None, 1, pad, dil, out=y, amax=(ax, aw), w_pairs=wp))
            yk = y.clone()
            ys, sums = ops.conv2d_fwd_stats(x, w, 1, pad, dil, amax=(ax, aw), w_pairs=wp)
            td = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, 1, pad, dil, out=dx, amax=(ag, aw), w_pairs_t=wpt))
            dxa = x.clone()
            ops.conv2d_dgrad(dy, w, x.shape, 1, pad, dil, out=dxa, beta=1, amax=(ag, aw), w_pairs_t=wpt)
            res.setdefault(mode, []).append((tf, td, yk, dx.clone(), ys, sums.clone(), dxa))
        ops.conv_set_h2p(-1)
        ops.conv_set_h2s(-1)
        a, b = res[0][-1], res[1][-1]
        same = (torch.equal(a[2], b[2]), torch.equal(a[3], b[3]) and torch.equal(a[6], b[6]), torch.equal(a[4], b[4]),
                float((a[5] - b[5]).abs().max() / a[5].abs().max()))
        t0f, t0d = min(r[0] for r in res[0]), min(r[1] for r in res[0])
        t1f, t1d = min(r[0] for r in res[1]), min(r[1] for r in res[1])
        print(f"{name:38s} N={N} fwd x3 {t0f*1e3:7.3f} ms {flops/t0f/1e12:6.1f} TF/s | h2p {t1f*1e3:7.3f} ms {flops/t1f/1e12:6.1f} TF/s "
              f"({t0f/t1f:4.2f}x) || dgrad x3 {t0d*1e3:7.3f} ms {flops/t0d/1e12:6.1f} | h2p {t1d*1e3:7.3f} ms {flops/t1d/1e12:6.1f} "
              f"({t0d/t1d:4.2f}x) || identical fwd/dgrad/stats-y {same[0]}/{same[1]}/{same[2]} stats rel {same[3]:.1e}", flush=True)


if __name__ == "__main__":
    main()
