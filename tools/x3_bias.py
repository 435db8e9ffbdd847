"""signed error of the conv kernels vs float64 on all-positive data (detects non-RNE accumulation)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import ops, _lib
torch.manual_seed(0)
N, Cin, Cout, H, W = 1, 512, 128, 16, 16
for name, gen in (("positive", lambda *s: torch.rand(*s) + 0.5), ("signed", lambda *s: torch.randn(*s))):
    x = gen(N, Cin, H, W); w = gen(Cout, Cin, 3, 3)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda(); wg = w.cuda().contiguous(memory_format=torch.channels_last)
    for v in (0, 8):
        _lib.load().rcf_conv_set_variant(v)
        y = ops.conv2d_fwd(xg, wg, None, 1, 1, 1).permute(0, 3, 1, 2).cpu().double()
        rel = (y - ref) / ref.abs().mean()
        print(f"{name} variant {v}: mean signed rel err {rel.mean():+.3e}  rms {rel.pow(2).mean().sqrt():.3e} max {rel.abs().max():.3e}")
