"""The step's conv layer shapes, each launched twice in a fixed order (for per-dispatch counter passes: tools/pmc_dispatch.sh
prints the counters of the SECOND launch of each shape next to its algorithmic bytes).
usage: python tools/pmc_shapes.py <fp32|bf16> <fwd|dgrad|wgrad> [--list]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

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
    ("layer2.conv2 3x3 128->128", 128, 128, 3, 1, 1, 60, 107, 16),
    ("layer1.conv3 1x1 64->256 @120x214", 64, 256, 1, 0, 1, 120, 214, 16),
]


def algorithmic_bytes(prec, which, Cin, Cout, k, H, W, N):
    e = 2 if prec == "bf16" else 4
    px = N * H * W
    wbytes = Cout * Cin * k * k * 4
    if which == "fwd":
        return px * Cin * e + px * Cout * e + wbytes
    if which == "dgrad":
        return px * Cout * e + px * Cin * e + wbytes
    return px * Cin * e + px * Cout * e + wbytes


def main():
    prec, which = sys.argv[1], sys.argv[2]
    if "--list" in sys.argv:
        for name, Cin, Cout, k, pad, dil, H, W, N in SHAPES:
            print(f"{name}|{algorithmic_bytes(prec, which, Cin, Cout, k, H, W, N)}|{2.0 * N * H * W * Cin * Cout * k * k}")
        return
    import torch
    import rcf_amd  # noqa
    from rcf_amd import ops
    dev = "cuda:0"
    for name, Cin, Cout, k, pad, dil, H, W, N in SHAPES:
        x = torch.randn(N, H, W, Cin, device=dev)
        dy = torch.randn(N, H, W, Cout, device=dev)
        w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        if prec == "fp32":
            ax, aw, ag = ops.absmax(x), ops.absmax(ops.weight_rsck(w)), ops.absmax(dy)
            wp, wpt = ops.weight_pairs(w, aw), ops.weight_pairs_t(w, aw)
            y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.zeros_like(w)
            for _ in range(2):
                if which == "fwd":
                    ops.conv2d_fwd(x, w, None, 1, pad, dil, out=y, amax=(ax, aw), w_pairs=wp)
                elif which == "dgrad":
                    ops.conv2d_dgrad(dy, w, x.shape, 1, pad, dil, out=dx, amax=(ag, aw), w_pairs_t=wpt)
                else:
                    ops.conv2d_wgrad(x, dy, w, dw, 1, pad, dil, beta=0, amax=(ax, ag))
        else:
            xb, dyb = x.bfloat16(), dy.bfloat16()
            wb, wbt = ops.weight_bf16(w), ops.weight_bf16(w, transpose=True)
            y, dx, dw = torch.empty_like(dyb), torch.empty_like(xb), torch.zeros_like(w)
            for _ in range(2):
                if which == "fwd":
                    ops.conv2d_fwd_bf16(xb, w, wb, None, 1, pad, dil, out=y)
                elif which == "dgrad":
                    ops.conv2d_dgrad_bf16(dyb, w, xb.shape, 1, pad, dil, out=dx, w_t_bf16=wbt)
                else:
                    ops.conv2d_wgrad_bf16(xb, dyb, w, dw, 1, pad, dil, beta=0)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
