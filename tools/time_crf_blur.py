"""CRF ms/frame at 480x854 (8 frames per call) over the iteration variants of round 6: blur passes in pairs (default) against one
launch per axis, and the vertex kernels' grid.  usage: python tools/time_crf_blur.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rcf_amd
from rcf_amd import synth
from rcf_amd.crf import crf_soft_batched
n, H, W = 8, 480, 854


def timed(fn, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


GRIDS = [1024, 256, 384, 512, 768, 2048, 128, 192]
for kind, iters, build in (("smooth", 5, 0), ("smooth", 50, 0), ("noise", 5, 3)):
    make = synth.noise_rgb if kind == "noise" else synth.smooth_rgb
    imgs = torch.from_numpy(np.stack([synth.normalize_rgb(make(H, W, 4000 + i)) for i in range(n)])).cuda()
    masks = torch.from_numpy(np.stack([synth.soft_blob_mask(H, W, 4000 + i) for i in range(n)])).cuda()
    head = rcf_amd.CRFHead(None, refine_iters=iters)
    rgb, unary = head.prepare(imgs, masks)
    row = []
    for seq in (1, 0):
        for gi in (0, 3, 2, 1, 7, 6):
            tune = seq | (gi << 1)
            t = timed(lambda: crf_soft_batched(rgb, unary, W, H, 0.0, 0.0, head.scomp, head.sxy, head.srgb, iters, build=build | (tune << 2))) / n
            row.append(f"{'seq ' if seq else 'pair'} grid {GRIDS[gi]:4d}: {t:.4f}")
    print(f"CRF {kind} T={iters} x{n} ms/frame | " + " | ".join(row))
