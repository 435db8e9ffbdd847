"""CRF on 8 frames of 480x854 a few times (for rocprofv3).  usage: crf_prof.py [smooth|smooth+<amp>|noise] [iters=5] [sort|packed|auto]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import synth
kind = sys.argv[1] if len(sys.argv) > 1 else "smooth"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, W, n = 480, 854, 8
amp = int(kind.split("+")[1]) if "+" in kind else 0          # "smooth+8": natural-looking frames with +-8 of pixel noise on top
make = synth.noise_rgb if kind == "noise" else synth.smooth_rgb
rng = np.random.default_rng(amp)
frames = [np.clip(make(H, W, 4000 + i).astype(np.int32) + rng.integers(-amp, amp + 1, (H, W, 3)), 0, 255).astype(np.uint8) for i in range(n)]
imgs = torch.from_numpy(np.stack([synth.normalize_rgb(fr) for fr in frames])).cuda()
masks = torch.from_numpy(np.stack([synth.soft_blob_mask(H, W, 4000 + i) for i in range(n)])).cuda()
head = rcf_amd.CRFHead(None, refine_iters=iters)
mode = sys.argv[3] if len(sys.argv) > 3 else "auto"
head.sort_build = {"sort": True, "packed": False}.get(mode, "auto")
for _ in range(4):
    head(imgs, masks)
torch.cuda.synchronize()
