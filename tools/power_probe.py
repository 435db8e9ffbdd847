"""Is the conv kernel's speed set by the chip's power limit?  The same launch (3x3 d4 512->512, 16 frames of 60x107: the
same instruction stream, the same memory traffic) on operands of different bit activity: random normal data, all zeros,
all ones, and random data whose low fp16 part is zero (values exactly representable in fp16: the m planes are all zero).
A kernel bound by instruction issue / memory takes the same time on all of them; one that runs against the power limit
gets faster as the operands toggle fewer bits (MI355X_MICROARCH.md, 'DVFS give-back').  usage: python tools/power_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import ops


def timeit(fn, iters=20):
    for _ in range(3):
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
    N, Cin, Cout, H, W, k, pad, dil = 16, 512, 512, 60, 107, 3, 4, 4
    flops = 2.0 * N * H * W * Cout * Cin * k * k
    y = torch.empty(N, H, W, Cout, device=dev)
    gens = {
        "random normal": lambda *s: torch.randn(*s, device=dev),
        "random, fp16-exact (m planes zero)": lambda *s: torch.randn(*s, device=dev).half().float(),
        "all ones": lambda *s: torch.ones(*s, device=dev),
        "all zeros": lambda *s: torch.zeros(*s, device=dev),
    }
    for rounds in range(2):
        for name, gen in gens.items():
            x = gen(N, H, W, Cin)
            w = (gen(Cout, Cin, k, k) * (0.05 if "random" in name else 1.0)).contiguous(memory_format=torch.channels_last)
            if "fp16-exact" in name:
                w = w.half().float().contiguous(memory_format=torch.channels_last)
            ax, aw = ops.absmax(x), ops.absmax(ops.weight_rsck(w))
            ops.conv_set_h2p(1)
            wp = ops.weight_pairs(w, aw)
            out = []
            for mode in (0, 1):
                ops.conv_set_h2p(mode)
                t = timeit(lambda: ops.conv2d_fwd(x, w, None, 1, pad, dil, out=y, amax=(ax, aw), w_pairs=wp))
                out.append(f"{'persistent' if mode else '128x256   '} {t * 1e3:6.3f} ms {flops / t / 1e12:6.1f} TF/s")
            ops.conv_set_h2p(-1)
            # the weight gradient of the same layer (the step's largest kernel family) and its bf16 twin
            dy = gen(N, H, W, Cout)
            ag = ops.absmax(dy)
            dw = torch.zeros_like(w)
            t = timeit(lambda: ops.conv2d_wgrad(x, dy, w, dw, 1, pad, dil, beta=0, amax=(ax, ag)))
            out.append(f"weight gradient {t * 1e3:6.3f} ms {flops / t / 1e12:6.1f} TF/s")
            xb, dyb = x.bfloat16(), dy.bfloat16()
            t = timeit(lambda: ops.conv2d_wgrad_bf16(xb, dyb, w, dw, 1, pad, dil, beta=0))
            out.append(f"bf16 weight gradient {t * 1e3:6.3f} ms {flops / t / 1e12:6.1f} TF/s")
            wb = ops.weight_bf16(w)
            yb = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
            t = timeit(lambda: ops.conv2d_fwd_bf16(xb, w, wb, None, 1, pad, dil, out=yb))
            out.append(f"bf16 forward {t * 1e3:6.3f} ms {flops / t / 1e12:6.1f} TF/s")
            print(f"{name:36s} | " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
