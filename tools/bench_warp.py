"""HBM roofline of the warp family at 480x854: algorithmic bytes 4*(2+2C) per pixel (flow + source + output)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import ops

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

B, C, H, W = 64, 3, 480, 854
x = torch.rand(B, C, H, W, device="cuda:0"); y = torch.rand(B, C, H, W, device="cuda:0")
# representative flow: the piecewise-affine synthetic field of SURVEY.md §8(d) (per-pixel N(0,0.5) jitter), tiled
import numpy as np
from rcf_amd import synth
base = np.stack([synth.voronoi_affine_flow(H, W, 7000 + i)[0] for i in range(8)])
fl = torch.from_numpy(np.tile(base, (B // 8, 1, 1, 1))).to("cuda:0")
if len(sys.argv) > 1 and sys.argv[1] == "random":
    fl = torch.randn(B, 2, H, W, device="cuda:0") * 4     # adversarial: every lane samples its own cache line
occ = torch.ones(B, 1, H, W, device="cuda:0")
px = B * H * W
t = timeit(lambda: ops.flow_warp(x, fl, "border"))
print(f"flow_warp        B={B}: {t*1e3:.3f} ms  {px*4*(2+2*C)/t/1e12:.2f} TB/s algorithmic ({t/B*1e6:.1f} us/frame)")
t = timeit(lambda: ops.warp_l1_residual(y, x, fl, occ, "border"))
print(f"warp+L1 fused    B={B}: {t*1e3:.3f} ms  {px*4*(2+2*C+1)/t/1e12:.2f} TB/s algorithmic ({t/B*1e6:.1f} us/frame)")
t = timeit(lambda: ops.occu_mask_bidirection(fl, -fl))
print(f"occ bidirection  B={B}: {t*1e3:.3f} ms  {px*4*(4+1)/t/1e12:.2f} TB/s algorithmic")
t = timeit(lambda: ops.occu_mask_backward(fl))
print(f"occ backward     B={B}: {t*1e3:.3f} ms  {px*4*(2+1)/t/1e12:.2f} TB/s algorithmic (4 float atomics / px)")
wb = ops.flow_warp(x, fl)
t = timeit(lambda: ops.photometric_loss(y, wb, occ))
print(f"photometric      B={B}: {t*1e3:.3f} ms  {px*4*(2*C+1)/t/1e12:.2f} TB/s algorithmic")
