"""CRFHead ms/frame (n frames per call): smooth and noise frames, T=5 and T=50, packed build vs sort build (RCF_CRF_BUILD_SORT),
at 480x854 and at the sizes in between that place the automatic switch (rcf_amd.crf.CRFHead.SORT_ABOVE).
usage: python tools/time_crf.py [frames_per_call=8]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rcf_amd
from rcf_amd import synth
from rcf_amd.crf import crf_soft_batched
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (H, W), kind, iters in (((480, 854), "smooth", 5), ((480, 854), "smooth", 50), ((480, 854), "noise", 5), ((480, 854), "noise", 0),
                            ((480, 854), "smooth", 0), ((240, 427), "noise", 5), ((120, 214), "noise", 5), ((240, 427), "smooth", 5)):
    make = synth.noise_rgb if kind == "noise" else synth.smooth_rgb
    imgs = torch.from_numpy(np.stack([synth.normalize_rgb(make(H, W, 4000 + i)) for i in range(n)])).cuda()
    masks = torch.from_numpy(np.stack([synth.soft_blob_mask(H, W, 4000 + i) for i in range(n)])).cuda()
    head = rcf_amd.CRFHead(None, refine_iters=iters)
    rgb, unary = head.prepare(imgs, masks)
    _, nv = crf_soft_batched(rgb, unary, W, H, 0.0, 0.0, head.scomp, head.sxy, head.srgb, 1, want_nvert=True)
    L = float(nv[:, 1].float().mean())
    t = {}
    for b in (0, 3):
        t[b] = timed(lambda: crf_soft_batched(rgb, unary, W, H, head.scomp_smooth, head.sxy_smooth, head.scomp, head.sxy, head.srgb, iters, build=b)) / n
    print(f"CRF {H}x{W} {kind:6s} T={iters:2d} x{n}: {L:9.0f} vertices/frame ({L / (H * W):.2f} per pixel)  packed build {t[0]:.4f}  sort build {t[3]:.4f} ms/frame"
          f"  ({t[0] / t[3]:.2f}x)")
