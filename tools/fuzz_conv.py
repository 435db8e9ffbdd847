"""Random conv shapes (both precisions of the step's kernels) against float64: forward, data gradient, weight gradient, with and
without a region; the yardstick is torch's own fp32 conv on the same inputs.  usage: python tools/fuzz_conv.py [cases=60] [seed=0]"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import ops

DEV = "cuda:0"
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)
nchw = lambda t: t.permute(0, 3, 1, 2).cpu()


def rms(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float(((a - b) ** 2).mean().sqrt() / ((b ** 2).mean().sqrt() + 1e-300))


bad = 0
for it in range(ncases):
    Cin = int(rng.choice([32, 64, 96, 128, 192, 256, 320, 512]))
    Cout = int(rng.choice([32, 64, 72, 128, 136, 256, 512]))
    k = int(rng.choice([1, 3]))
    stride = int(rng.choice([1, 1, 2]))
    dil = int(rng.choice([1, 2, 3])) if k == 3 else 1
    pad = dil * (k // 2)
    N, H, W = int(rng.randint(1, 4)), int(rng.randint(6, 40)), int(rng.randint(6, 40))
    g = torch.Generator().manual_seed(1000 + it)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yref = F.conv2d(xd, wd, None, stride, pad, dil)
    Ho, Wo = yref.shape[2:]
    reg = None
    if stride == 1 and rng.rand() < 0.4 and Ho >= 6 and Wo >= 6:
        rh, rw = int(rng.randint(3, Ho + 1)), int(rng.randint(3, Wo + 1))
        y0, x0 = int(rng.randint(0, Ho - rh + 1)), int(rng.randint(0, Wo - rw + 1))
        t = int(rng.randint(1, min(rh, rw) // 2)) if min(rh, rw) >= 4 and rng.rand() < 0.5 else 0
        reg = (y0, x0, rh, rw, t) if t else (y0, x0, rh, rw)
    dy = torch.randn(yref.shape, generator=g)
    mask = torch.ones(dy.shape)
    if reg is not None:
        mask = torch.zeros(dy.shape)
        mask[:, :, reg[0]:reg[0] + reg[2], reg[1]:reg[1] + reg[3]] = 1
        if len(reg) > 4:
            t = reg[4]
            mask[:, :, reg[0] + t:reg[0] + reg[2] - t, reg[1] + t:reg[1] + reg[3] - t] = 0
    yref.backward((dy * mask).double())
    w32 = w.clone().requires_grad_(True)
    x32 = x.clone().requires_grad_(True)
    F.conv2d(x32, w32, None, stride, pad, dil).backward(dy * mask)
    xg, gg = nhwc(x), nhwc(dy)
    wg = w.to(DEV).contiguous(memory_format=torch.channels_last)
    ax, aw, ag = ops.absmax(xg), ops.absmax(ops.weight_rsck(wg)), ops.absmax(gg)
    res = {}
    dw = torch.zeros_like(wg)
    ops.conv2d_wgrad(xg, gg, wg, dw, stride, pad, dil, beta=1, region=reg, amax=(ax, ag))
    res["wgrad f32"] = (rms(dw.cpu(), wd.grad), rms(w32.grad, wd.grad), 4.0, 5e-7)
    if reg is None:
        y = ops.conv2d_fwd(xg, wg, None, stride, pad, dil, amax=(ax, aw))
        res["fwd f32"] = (rms(nchw(y), yref), 0.0, 0.0, 2e-6)
        dx = ops.conv2d_dgrad(gg, wg, xg.shape, stride, pad, dil, amax=(ag, aw))
        res["dgrad f32"] = (rms(nchw(dx), xd.grad), rms(x32.grad, xd.grad), 4.0, 5e-7)
    if Cin % 8 == 0 and Cout % 8 == 0:
        xb, gb = xg.to(torch.bfloat16), gg.to(torch.bfloat16)
        xq, gq = nchw(xb.float()).double(), (nchw(gb.float()) * mask).double()
        wq = w.to(torch.bfloat16).double().requires_grad_(True)
        xqd = xq.clone().requires_grad_(True)
        F.conv2d(xqd, wq, None, stride, pad, dil).backward(gq)
        dwb = torch.zeros_like(wg)
        ops.conv2d_wgrad_bf16(xb, gb, wg, dwb, stride, pad, dil, beta=1, region=reg)
        res["wgrad bf16"] = (rms(dwb.cpu(), wq.grad), 0.0, 0.0, 2e-6)        # bf16 operands are exact inputs: fp32 accumulation error only
    line = f"{it:3d} N{N} {Cin}->{Cout} k{k} s{stride} d{dil} {H}x{W} reg={reg}: " + " ".join(f"{n} {v[0]:.1e}" for n, v in res.items())
    ok = all(v[0] < max(v[2] * v[1], v[3]) for v in res.values())
    if not ok:
        bad += 1
    print(line + ("" if ok else "   <-- FAIL"), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
