"""Where does the tile splat stop paying?  Natural-looking frames with more and more pixel noise on top (amplitude 0 ... 64 of 255):
CRF ms/frame (8 frames of 480x854, T = 5 and the per-pass cost from T = 25) with the list walk (RCF_CRF_SPLAT_GATHER), the tile splat
forced (RCF_CRF_SPLAT_TILES) and the default rule (per frame: tile lists total <= 1/4 of the entries), beside the lattice's vertex count.
usage: python tools/crf_texture_sweep.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rcf_amd
from rcf_amd import synth
from rcf_amd.crf import crf_soft_batched
n, H, W = 8, 480, 854
GATHER, TILES = 0x4000 >> 8, 0x8000 >> 8


def timed(fn, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


head = rcf_amd.CRFHead(None, refine_iters=5)
for amp in (0, 2, 4, 6, 8, 12, 16, 24, 32, 64):
    rng = np.random.default_rng(amp)
    frames = []
    for i in range(n):
        base = synth.smooth_rgb(H, W, 4000 + i).astype(np.int32)
        frames.append(np.clip(base + rng.integers(-amp, amp + 1, base.shape), 0, 255).astype(np.uint8))
    rgb = torch.from_numpy(np.stack(frames)).cuda()
    masks = torch.from_numpy(np.stack([synth.soft_blob_mask(H, W, 4000 + i) for i in range(n)])).cuda()
    m = np.clip(masks.cpu().numpy(), 1e-4, 1 - 1e-4).reshape(n, -1)
    unary = torch.from_numpy(np.stack([-np.log(1 - m), -np.log(m)], axis=2).astype(np.float32)).cuda()
    row = []
    for name, fl in (("list walk", GATHER), ("tiles", TILES), ("default", 0), ("sort build", 3)):
        t5 = timed(lambda: crf_soft_batched(rgb, unary, W, H, 0.0, 0.0, head.scomp, head.sxy, head.srgb, 5, build=fl))
        t25 = timed(lambda: crf_soft_batched(rgb, unary, W, H, 0.0, 0.0, head.scomp, head.sxy, head.srgb, 25, build=fl))
        row.append(f"{name} {t5 / n:.3f} ({(t25 - t5) / 20 * 1e3:.0f} us/pass)")
    nv = crf_soft_batched(rgb, unary, W, H, 0.0, 0.0, head.scomp, head.sxy, head.srgb, 1, want_nvert=True)[1][:, 1].float().mean().item()
    print(f"noise +-{amp:2d}: vertices/frame {nv:9.0f} ({nv / (H * W * 6) * 100:4.1f} % of the entries) | " + " | ".join(row))
