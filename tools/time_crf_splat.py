"""CRF ms/frame at 480x854 (8 frames per call): the tile splat (default on natural frames, round 6) against the gather over the
CSR lists (RCF_CRF_SPLAT_GATHER), with slice and per-tile sums as two kernels (RCF_CRF_SLICE_SPLAT_SEPARATE), and forced on noise frames.  usage: python tools/time_crf_splat.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rcf_amd
from rcf_amd import synth
from rcf_amd.crf import crf_soft_batched
n, H, W = 8, 480, 854
GATHER, TILES, SEPARATE = 0x4000 >> 8, 0x8000 >> 8, 0x10000 >> 8


def timed(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for kind, iters, build in (("smooth", 5, 0), ("smooth", 50, 0), ("smooth", 0, 0), ("noise", 5, 0), ("noise", 5, 3)):
    make = synth.noise_rgb if kind == "noise" else synth.smooth_rgb
    imgs = torch.from_numpy(np.stack([synth.normalize_rgb(make(H, W, 4000 + i)) for i in range(n)])).cuda()
    masks = torch.from_numpy(np.stack([synth.soft_blob_mask(H, W, 4000 + i) for i in range(n)])).cuda()
    head = rcf_amd.CRFHead(None, refine_iters=iters)
    rgb, unary = head.prepare(imgs, masks)
    row = []
    for rnd in range(2):
        for name, fl in (("gather", GATHER), ("default", 0), ("separate", SEPARATE), ("tiles", TILES)):
            t = timed(lambda: crf_soft_batched(rgb, unary, W, H, 0.0, 0.0, head.scomp, head.sxy, head.srgb, iters, build=build | fl)) / n
            row.append(f"{name} {t:.4f}")
    print(f"CRF {kind} T={iters} build {build} x{n} ms/frame | " + " | ".join(row))
