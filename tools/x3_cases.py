import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rcf_amd
from rcf_amd import ops, _lib
def relerr(a, b): return float((a.double().cpu() - b.double().cpu()).abs().max() / (b.abs().max() + 1e-30))
CASES = [(2, 4096, 256, 3, 1, 6, 6, 8, 12), (2, 256, 256, 3, 1, 6, 6, 8, 12), (2, 256, 16, 1, 1, 0, 1, 8, 12),
         (4, 2304, 256, 3, 1, 6, 6, 16, 24), (4, 2048, 512, 1, 1, 0, 1, 8, 12), (4, 512, 512, 3, 1, 4, 4, 8, 12), (4, 512, 2048, 1, 1, 0, 1, 8, 12)]
for case in CASES:
    N, Cin, Cout, k, stride, pad, dil, H, W = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y = torch.nn.functional.conv2d(xd, wd, None, stride, pad, dil)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    xg = x.permute(0, 2, 3, 1).contiguous().cuda(); wg = w.cuda().contiguous(memory_format=torch.channels_last)
    gg = dy.permute(0, 2, 3, 1).contiguous().cuda()
    for v in (0, -1):
        _lib.load().rcf_conv_set_variant(v)
        yy = ops.conv2d_fwd(xg, wg, None, stride, pad, dil).permute(0, 3, 1, 2)
        dx = ops.conv2d_dgrad(gg, wg, xg.shape, stride, pad, dil).permute(0, 3, 1, 2)
        dw = torch.zeros_like(wg); ops.conv2d_wgrad(xg, gg, wg, dw, stride, pad, dil, beta=1)
        print(case, "variant", v, "fwd %.2e dgrad %.2e wgrad %.2e" % (relerr(yy, y.detach()), relerr(dx, xd.grad), relerr(dw, wd.grad)))
