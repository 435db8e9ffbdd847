"""Are the convs so power-bound that they run as fast on fewer CUs?  The same conv launches (128x256-tile forward and the
128x256-tile weight gradient, 3x3 d4 512->512, 16 frames) on streams created with hipExtStreamCreateWithCUMask: all 256 CUs,
7 of every 8, 3 of every 4, 1 of every 2; and a streaming batch-norm pass on the complementary CUs at the same time.
usage: python tools/cu_mask_probe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import ops

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(pattern, period, ncu=256):
    """a torch stream whose kernels may only use CU i when pattern[i % period] == 1"""
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(ncu):
        if pattern[i % period]:
            mask[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    err = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, mask)
    assert err == 0, err
    return torch.cuda.ExternalStream(st.value)


def timeit(fn, stream, iters=10):
    with torch.cuda.stream(stream):
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
    x = torch.randn(N, H, W, Cin, device=dev)
    dy = torch.randn(N, H, W, Cout, device=dev)
    w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    ax, aw, ag = ops.absmax(x), ops.absmax(ops.weight_rsck(w)), ops.absmax(dy)
    ops.conv_set_h2p(0)                       # the 128 x 256 kernel: its grid adapts to any CU count
    ops.conv_set_wgrad_big(0)
    wp = ops.weight_pairs(w, aw)
    y = torch.empty_like(dy)
    dw = torch.zeros_like(w)
    big = torch.randn(16, 60, 107, 2048, device=dev)
    bo = torch.empty_like(big)
    mean, invstd = torch.zeros(2048, device=dev), torch.ones(2048, device=dev)
    gamma, beta = torch.ones(2048, device=dev), torch.zeros(2048, device=dev)
    fwd = lambda: ops.conv2d_fwd(x, w, None, 1, pad, dil, out=y, amax=(ax, aw), w_pairs=wp)
    wgr = lambda: ops.conv2d_wgrad(x, dy, w, dw, 1, pad, dil, beta=0, amax=(ax, ag))
    bn = lambda: ops.bn_apply(big, mean, invstd, gamma, beta, True, out=bo)
    full = masked_stream([1], 1)
    t_bn_full = timeit(bn, full)
    print(f"all 256 CUs: forward {timeit(fwd, full)*1e3:6.3f} ms, weight gradient {timeit(wgr, full)*1e3:6.3f} ms, "
          f"batch-norm apply on 841 MB {t_bn_full*1e6:6.1f} us", flush=True)
    for name, pat in (("7 of 8", [1, 1, 1, 1, 1, 1, 1, 0]), ("3 of 4", [1, 1, 1, 0]), ("1 of 2", [1, 0])):
        s_on = masked_stream(pat, len(pat))
        s_off = masked_stream([1 - v for v in pat], len(pat))
        tf, tw = timeit(fwd, s_on), timeit(wgr, s_on)
        tb = timeit(bn, s_off)
        # both at once: convs on their CUs, batch norm passes on the others, until the convs are done
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s_on):
            e0.record()
            for _ in range(10):
                fwd()
            e1.record()
        with torch.cuda.stream(s_off):
            for _ in range(int(10 * tf / tb) + 1):
                bn()
        torch.cuda.synchronize()
        tboth = e0.elapsed_time(e1) / 10 * 1e-3
        print(f"convs on {name} CUs: forward {tf*1e3:6.3f} ms ({flops/tf/1e12:5.1f} TF/s), weight gradient {tw*1e3:6.3f} ms | batch norm on the "
              f"other CUs {tb*1e6:7.1f} us ({t_bn_full/tb:4.2f} of its full-chip rate) | forward with the batch norm running beside it {tboth*1e3:6.3f} ms", flush=True)


if __name__ == "__main__":
    main()
