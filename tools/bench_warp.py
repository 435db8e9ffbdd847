#!/usr/bin/env python3
"""A/B of the warp kernels (bench.py's warp leg shapes): per-pixel kernels (variant 0) against the RGB / border tile kernels
(variant 1); checks that both give the same bits."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import rcf_amd
from rcf_amd import _lib, ops, synth


def timeit(fn, n=5 if "big" in sys.argv else 20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    dev = "cuda:0"
    shapes = [(480, 854, 64)] if "big" in sys.argv else [(480, 854, 64), (96, 160, 64), (384, 384, 16)]
    for (H, W, nframes) in shapes:
        base = np.stack([synth.voronoi_affine_flow(H, W, 7000 + i)[0] for i in range(8)])
        fl = torch.from_numpy(np.tile(base, (nframes // 8, 1, 1, 1))).to(dev)
        fl[0, :, :4, :4] = 1e9; fl[1, :, :4, :4] = -1e9; fl[2, 0, 5, 5] = float("nan")
        x = torch.rand(nframes, 3, H, W, device=dev); y = torch.rand(nframes, 3, H, W, device=dev)
        occ = (torch.rand(nframes, 1, H, W, device=dev) > 0.2).float()
        px = nframes * H * W
        res = {}
        for v in (0, 1):                         # 0: RCF_WARP_PER_PIXEL (per call), 1: the tile kernels
            pad = "border" if v else "border_per_pixel"
            w = ops.flow_warp(x, fl, pad)
            l1 = ops.warp_l1_residual(y, x, fl, occ, pad)
            t_l1 = timeit(lambda: ops.warp_l1_residual(y, x, fl, occ, pad))
            t_w = timeit(lambda: ops.flow_warp(x, fl, pad))
            res[v] = (w, l1)
            print(f"{H}x{W}x{nframes} variant {v:#x}: warp_l1 {t_l1*1e6:8.1f} us  {px*36/t_l1/1e9:7.1f} GB/s ({px*36/t_l1/8e12:.3f})   "
                  f"flow_warp {t_w*1e6:8.1f} us {px*32/t_w/1e9:7.1f} GB/s ({px*32/t_w/8e12:.3f})", flush=True)
        for v in (1,):
            print(f"   variant {v} vs 0: warped identical {torch.equal(res[0][0], res[v][0])}; l1 sums",
                  [float(a) for a in res[0][1]], [float(a) for a in res[v][1]])


if __name__ == "__main__":
    main()
