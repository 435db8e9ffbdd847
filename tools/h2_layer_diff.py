"""debug: every forward conv of one training step run twice -- fp16 pairs vs bf16 triples -- and the difference logged"""
import copy, os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import rcf_amd
from rcf_amd import ops, config, synth
import test_model_gpu as T

H, W = 96, 160
model = T._build(H, W, False, "cuda:0", rcf_amd.RCFModel)
tr = rcf_amd.Trainer(model, lr=1e-4, weight_decay=1e-4, device="cuda:0")
batch = T._batch(2, H, W, "cuda:0")
orig_f, orig_s, orig_d, orig_w = ops.conv2d_fwd, ops.conv2d_fwd_stats, ops.conv2d_dgrad, ops.conv2d_wgrad
log = []

def relmax(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300))

def fwd(x, w, bias=None, stride=1, pad=0, dil=1, act=0, slope=0.0, out=None, beta=0, region=None, amax=None, w_pairs=None):
    o0 = out.clone() if out is not None else None
    y = orig_f(x, w, bias, stride, pad, dil, act, slope, out=out, beta=beta, region=region, amax=amax, w_pairs=w_pairs)
    if amax is not None and amax[0] is not None:
        y2 = orig_f(x, w, bias, stride, pad, dil, act, slope, out=o0, beta=beta, region=region)
        log.append(("fwd", tuple(x.shape), tuple(w.shape), region, relmax(y, y2), float(amax[0].view(torch.float32)), float(x.abs().max()),
                    float(amax[1].view(torch.float32)), float(w.abs().max())))
    return y

def fwd_s(x, w, stride=1, pad=0, dil=1, amax=None, w_pairs=None):
    y, s = orig_s(x, w, stride, pad, dil, amax=amax, w_pairs=w_pairs)
    if amax is not None and amax[0] is not None:
        y2, s2 = orig_s(x, w, stride, pad, dil)
        log.append(("fwd+stats", tuple(x.shape), tuple(w.shape), None, relmax(y, y2), float(amax[0].view(torch.float32)), float(x.abs().max()),
                    float(amax[1].view(torch.float32)), float(w.abs().max()), relmax(s, s2)))
    return y, s

def dgrad(dy, w, xshape, stride=1, pad=0, dil=1, out=None, beta=0, region=None, amax=None):
    o0 = out.clone() if out is not None else None
    y = orig_d(dy, w, xshape, stride, pad, dil, out=out, beta=beta, region=region, amax=amax)
    if amax is not None and amax[0] is not None:
        y2 = orig_d(dy, w, xshape, stride, pad, dil, out=o0, beta=beta, region=region)
        m = None
        log.append(("dgrad", tuple(dy.shape), tuple(w.shape), region, relmax(y, y2) if region is None else -1.0, float(amax[0].view(torch.float32)), float(dy.abs().max()),
                    float(amax[1].view(torch.float32)), float(w.abs().max())))
    return y

def wgrad(x, dy, w_like, dw, stride=1, pad=0, dil=1, beta=1, region=None, amax=None):
    d0 = dw.clone()
    y = orig_w(x, dy, w_like, dw, stride, pad, dil, beta=beta, region=region, amax=amax)
    if amax is not None and amax[0] is not None:
        y2 = orig_w(x, dy, w_like, d0, stride, pad, dil, beta=beta, region=region)
        log.append(("wgrad", tuple(x.shape), tuple(w_like.shape), region, relmax(y, y2), float(amax[0].view(torch.float32)),
                    float(x.abs().max()) if region is None else -1.0, float(amax[1].view(torch.float32)), float(dy.abs().max())))
    return y

ops.conv2d_fwd, ops.conv2d_fwd_stats, ops.conv2d_dgrad, ops.conv2d_wgrad = fwd, fwd_s, dgrad, wgrad
rcf_amd.layers.OVERLAP_WGRAD = False
tr.step(batch)
torch.cuda.synchronize()
for r in log:
    flag = " <<<<" if r[4] > 5e-6 or (r[6] >= 0 and r[5] < r[6]) or r[7] < r[8] else ""
    print(r, flag)
