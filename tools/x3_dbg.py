import os, sys, copy, types, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rcf_amd
from rcf_amd import _lib, layers
import test_model_gpu as T
H, W, B = 64, 96, 2
saved = {}
for v in (0, 8 | 0x6000):
    _lib.load().rcf_conv_set_variant(v)
    hip = T._build(H, W, False, "cuda:0", rcf_amd.RCFModel)
    tr = rcf_amd.Trainer(hip, device="cuda:0")
    tr.fp.zero_grad(); hip.train()
    rec = {}
    for name, mod in hip.decode_head3.named_modules():
        if isinstance(mod, (layers.Conv2d, layers.BatchNorm2d)):
            def wrap(mod=mod, name=name, orig=mod.fwd):
                def f(x, *a, **k):
                    y = orig(x, *a, **k)
                    rec[name + ".in"] = x; rec[name + ".out"] = y
                    return y
                return f
            mod.fwd = wrap()
    hip(T._batch(B, H, W, "cuda:0"))["loss"].backward()
    out = {k: a.t.clone() for k, a in rec.items()}
    out.update({"grad." + n: p.grad.clone() for n, p in hip.decode_head3.named_parameters()})
    saved[v] = out
a, b = saved[0], saved[8 | 0x6000]
for k in a:
    d = float((a[k] - b[k]).abs().max()) / (float(a[k].abs().max()) + 1e-30)
    print(f"{k:40s} rel diff {d:.2e}  shape {tuple(a[k].shape)}")
