"""per-parameter gradient error of the HIP step vs the float64 oracle, for conv variants 0 (fp32 MFMA) and -1 (default)"""
import os, sys, copy, types, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rcf_amd, rcf_torch as orc
from rcf_amd import _lib
import test_model_gpu as T
H, W, B = 64, 96, 2
o64 = T._build(H, W, False, "cpu", orc.RCFModel).double()
b32 = T._batch(B, H, W, "cpu")
b64 = {k: ([t.double() for t in v] if k in ("imgs", "gt_fw_flows", "gt_bw_flows") else v) for k, v in b32.items()}
o64.train(); o64(b64)["loss"].backward()
g64 = dict(o64.named_parameters())
for v in (0, 8 | 0x6000, 8 | 0x5000, 8 | 0x3000):
    _lib.load().rcf_conv_set_variant(v)
    hip = T._build(H, W, False, "cuda:0", rcf_amd.RCFModel)
    tr = rcf_amd.Trainer(hip, device="cuda:0")
    tr.fp.zero_grad(); hip.train()
    hip(T._batch(B, H, W, "cuda:0"))["loss"].backward()
    rows = []
    for n, p in hip.named_parameters():
        t = g64[n].grad; sc = float(t.abs().max())
        if sc < 1e-12: continue
        rows.append((float((p.grad.cpu().double() - t).abs().max()) / sc, n, tuple(p.shape)))
    cur = {n: e for e, n, _ in rows}
    if v == 0:
        base = cur
    else:
        rr = sorted(((cur[n] / max(base[n], 1e-7), cur[n], base[n], n) for n in cur), reverse=True)
        print("variant", hex(v))
        for r in rr[:4]: print("   ratio %.1f x3 %.2e fp32 %.2e %s" % r)
