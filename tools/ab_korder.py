"""A/B of the K order of the 3x3 forward / data-gradient convs (csrc/rcf_common.h rcf_kchunk): tap outer (the weight's memory
order) against channel chunks of 64 outer, taps inner.  Same process, interleaved; the two orders re-associate the sums, so
the outputs are compared with each other and with float64 on a sample.  usage: python tools/ab_korder.py [bf16|fp32] [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import ops

# name, Cin, Cout, k, pad, dil, H, W, N
SHAPES = [
    ("layer4.conv2 3x3d4 512->512", 512, 512, 3, 4, 4, 60, 107, 16),
    ("layer4.conv2 3x3d2 512->512", 512, 512, 3, 2, 2, 60, 107, 16),
    ("layer3.conv2 3x3d2 256->256", 256, 256, 3, 2, 2, 60, 107, 16),
    ("dh2 coarse 3x3d3 2048->256", 2048, 256, 3, 3, 3, 60, 107, 16),
    ("dh3 coarse 3x3d6 4096->256", 4096, 256, 3, 6, 6, 60, 107, 8),
    ("dh2.convs.1 3x3d6 256->256 @120x214", 256, 256, 3, 6, 6, 120, 214, 16),
    ("layer2.conv2 3x3 128->128", 128, 128, 3, 1, 1, 60, 107, 16),
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
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    dev = "cuda:0"
    tot = {(d, m): 0.0 for d in ("fwd", "dgrad") for m in (0, 1)}
    for name, Cin, Cout, k, pad, dil, H, W, N in SHAPES:
        if len(sys.argv) > 2:
            N = int(sys.argv[2])
        x = torch.randn(N, H, W, Cin, device=dev)
        dy = torch.randn(N, H, W, Cout, device=dev)
        w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        flops = 2.0 * N * H * W * Cout * Cin * k * k
        res = {}
        for mode in (0, 1, 0, 1):
            ops.conv_set_korder(mode)
            if prec == "bf16":
                xb, dyb = x.bfloat16(), dy.bfloat16()
                wb, wbt = ops.weight_bf16(w), ops.weight_bf16(w, transpose=True)
                y = torch.empty(N, H, W, Cout, device=dev, dtype=torch.float32)
                dx = torch.empty_like(xb)
                tf = timeit(lambda: ops.conv2d_fwd_bf16(xb, w, wb, None, 1, pad, dil, out=y, out_dtype=torch.float32))
                td = timeit(lambda: ops.conv2d_dgrad_bf16(dyb, w, xb.shape, 1, pad, dil, out=dx, w_t_bf16=wbt))
            else:
                ax, aw, ag = ops.absmax(x), ops.absmax(ops.weight_rsck(w)), ops.absmax(dy)
                wp, wpt = ops.weight_pairs(w, aw), ops.weight_pairs_t(w, aw)
                y, dx = torch.empty_like(dy), torch.empty_like(x)
                tf = timeit(lambda: ops.conv2d_fwd(x, w, None, 1, pad, dil, out=y, amax=(ax, aw), w_pairs=wp))
                td = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, 1, pad, dil, out=dx, amax=(ag, aw), w_pairs_t=wpt))
            res.setdefault(mode, []).append((tf, td, y.float().clone(), dx.float().clone()))
        ops.conv_set_korder(1)
        a, b = res[0][-1], res[1][-1]
        t = {(d, m): min(r[i] for r in res[m]) for i, d in ((0, "fwd"), (1, "dgrad")) for m in (0, 1)}
        for key in t:
            tot[key] += t[key]
        ef = float((a[2] - b[2]).abs().max() / a[2].abs().max())
        ed = float((a[3] - b[3]).abs().max() / a[3].abs().max())
        # float64 reference of the forward on one image
        xr = (x[:1].bfloat16().double() if prec == "bf16" else x[:1].double()).permute(0, 3, 1, 2).cpu()
        wr = (w.bfloat16().double() if prec == "bf16" else w.double()).cpu()
        ref = torch.nn.functional.conv2d(xr, wr, None, 1, pad, dil).permute(0, 2, 3, 1)
        e64 = [float((r[2][:1].double().cpu() - ref).abs().max() / ref.abs().max()) for r in (a, b)]
        print(f"{name:38s} N={N} {prec} fwd: tap outer {t['fwd', 0]*1e3:7.3f} ms {flops/t['fwd', 0]/1e12:7.1f} TF/s | chunk outer {t['fwd', 1]*1e3:7.3f} ms "
              f"{flops/t['fwd', 1]/1e12:7.1f} TF/s ({t['fwd', 0]/t['fwd', 1]:4.2f}x) || dgrad: {t['dgrad', 0]*1e3:7.3f} ms {flops/t['dgrad', 0]/1e12:7.1f} | "
              f"{t['dgrad', 1]*1e3:7.3f} ms {flops/t['dgrad', 1]/1e12:7.1f} ({t['dgrad', 0]/t['dgrad', 1]:4.2f}x) || max diff between the orders "
              f"fwd {ef:.1e} dgrad {ed:.1e}; fwd vs float64 {e64[0]:.1e} / {e64[1]:.1e}", flush=True)
    print(f"sum over the shapes: fwd {tot['fwd', 0]*1e3:.3f} -> {tot['fwd', 1]*1e3:.3f} ms, dgrad {tot['dgrad', 0]*1e3:.3f} -> {tot['dgrad', 1]*1e3:.3f} ms")


if __name__ == "__main__":
    main()
