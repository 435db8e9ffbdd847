"""CRFHead ms/frame (8 frames 480x854): smooth and noise frames, T=5 and T=50"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rcf_amd
from rcf_amd import synth
H, W, n = 480, 854, 8
for kind, iters in (("smooth", 5), ("smooth", 50), ("noise", 5)):
    make = synth.noise_rgb if kind == "noise" else synth.smooth_rgb
    imgs = torch.from_numpy(np.stack([synth.normalize_rgb(make(H, W, 4000 + i)) for i in range(n)])).cuda()
    masks = torch.from_numpy(np.stack([synth.soft_blob_mask(H, W, 4000 + i) for i in range(n)])).cuda()
    head = rcf_amd.CRFHead(None, refine_iters=iters)
    head(imgs, masks); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        head(imgs, masks)
    e1.record(); torch.cuda.synchronize()
    print(f"CRFHead {kind} T={iters}: {e0.elapsed_time(e1) / 5 / n:.4f} ms/frame")
