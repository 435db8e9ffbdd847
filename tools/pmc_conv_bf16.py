"""Runs one bf16 conv shape a few times (for rocprofv3 --pmc passes). usage: pmc_conv_bf16.py [fwd|dgrad|wgrad] [tile]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import _lib, ops
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
if len(sys.argv) > 2:
    _lib.load().rcf_conv_bf16_set_tile(int(sys.argv[2]))
N, Cin, Cout, H, W = 16, 512, 512, 60, 107
x = torch.randn(N, H, W, Cin, device="cuda:0").to(torch.bfloat16)
w = (torch.randn(Cout, Cin, 3, 3, device="cuda:0") * 0.05).contiguous(memory_format=torch.channels_last)
wb = ops.weight_bf16(w)
y = ops.conv2d_fwd_bf16(x, w, wb, None, 1, 4, 4)
dy = torch.randn(y.shape, device="cuda:0").to(torch.bfloat16); dw = torch.zeros_like(w); dx = torch.empty_like(x)
for _ in range(3):
    if which == "fwd": ops.conv2d_fwd_bf16(x, w, wb, None, 1, 4, 4, out=y)
    elif which == "dgrad": ops.conv2d_dgrad_bf16(dy, w, x.shape, 1, 4, 4, out=dx)
    else: ops.conv2d_wgrad_bf16(x, dy, w, dw, 1, 4, 4, beta=0)
torch.cuda.synchronize()
