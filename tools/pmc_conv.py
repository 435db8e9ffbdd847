"""Runs one conv shape a few times (for rocprofv3 --pmc passes). usage: pmc_conv.py [fwd|dgrad|wgrad]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
ops.conv_set_h2p(int(os.environ.get("RCF_H2P", "-1")))          # 0: 128x256 kernel, 1 / -1: persistent LDS-DMA kernel
N, Cin, Cout, H, W = 16, 512, 512, 60, 107
x = torch.randn(N, H, W, Cin, device="cuda:0")
w = (torch.randn(Cout, Cin, 3, 3, device="cuda:0") * 0.05).contiguous(memory_format=torch.channels_last)
y = ops.conv2d_fwd(x, w, None, 1, 4, 4)
dy = torch.randn_like(y); dw = torch.zeros_like(w)
ax, aw, ag = ops.absmax(x), ops.absmax(ops.weight_rsck(w)), ops.absmax(dy)      # fp16-pair kernels
wp = ops.weight_pairs(w, aw)
if os.environ.get("RCF_BENCH_PAIRS", "1") == "0":
    ax = aw = ag = wp = None
for _ in range(3):
    if which == "fwd": ops.conv2d_fwd(x, w, None, 1, 4, 4, out=y, amax=(ax, aw), w_pairs=wp)
    elif which == "dgrad": ops.conv2d_dgrad(dy, w, x.shape, 1, 4, 4, out=x, amax=(ag, aw))
    else: ops.conv2d_wgrad(x, dy, w, dw, 1, 4, 4, beta=0, amax=(ax, ag))
torch.cuda.synchronize()
