import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
import rcf_amd
from rcf_amd import synth, _lib
from rcf_amd.crf import crf_soft_batched
H, W, n = 480, 854, 8
imgs = torch.from_numpy(np.stack([synth.normalize_rgb(synth.smooth_rgb(H, W, 4000 + i)) for i in range(n)])).cuda()
masks = torch.from_numpy(np.stack([synth.soft_blob_mask(H, W, 4000 + i) for i in range(n)])).cuda()
head = rcf_amd.CRFHead(None, refine_iters=0)
rgb, unary = head.prepare(imgs, masks)
for dbg in (0, 1, 2, 0):
    _lib.load().rcf_crf_set_variant(dbg << 8)
    run = lambda: crf_soft_batched(rgb, unary, W, H, head.scomp_smooth, head.sxy_smooth, head.scomp, head.sxy, head.srgb, 0)
    try:
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        print(f"dbg {dbg}: T=0 call (build only) {e0.elapsed_time(e1) / 5:.3f} ms per 8 frames")
    except Exception as e:
        print("dbg", dbg, "error", e)
_lib.load().rcf_crf_set_variant(0)
