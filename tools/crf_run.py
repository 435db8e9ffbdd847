"""CRF-only driver for profiling: crf_run.py [smooth|noise] [frames] [iters]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import synth
kind = sys.argv[1] if len(sys.argv) > 1 else "smooth"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 5
gen = synth.smooth_rgb if kind == "smooth" else synth.noise_rgb
H, W = 480, 854
head = rcf_amd.CRFHead(None, refine_iters=T)
imgs = torch.from_numpy(np.stack([synth.normalize_rgb(gen(H, W, 4000 + i)) for i in range(nf)])).to("cuda:0")
masks = torch.from_numpy(np.stack([synth.soft_blob_mask(H, W, 4000 + i) for i in range(nf)])).to("cuda:0")
for _ in range(3):
    head(imgs, masks)
torch.cuda.synchronize()
