"""CRF on 8 frames of 480x854 a few times (for rocprofv3).  usage: crf_prof.py [smooth|noise] [iters=5] [sort|packed|auto]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import synth
kind = sys.argv[1] if len(sys.argv) > 1 else "smooth"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, W, n = 480, 854, 8
make = synth.noise_rgb if kind == "noise" else synth.smooth_rgb
imgs = torch.from_numpy(np.stack([synth.normalize_rgb(make(H, W, 4000 + i)) for i in range(n)])).cuda()
masks = torch.from_numpy(np.stack([synth.soft_blob_mask(H, W, 4000 + i) for i in range(n)])).cuda()
head = rcf_amd.CRFHead(None, refine_iters=iters)
mode = sys.argv[3] if len(sys.argv) > 3 else "auto"
head.sort_build = {"sort": True, "packed": False}.get(mode, "auto")
for _ in range(4):
    head(imgs, masks)
torch.cuda.synchronize()
