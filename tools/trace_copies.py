"""Where do a training step's device copies come from?  Runs one step under torch's profiler and prints the Python call sites of
every op that ends in a memcpy (aten::copy_, aten::clone, aten::to, aten::contiguous ...).  usage: python tools/trace_copies.py [fp32|bf16]"""
import collections
import os
import sys
import traceback
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import config, synth

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
H, W, B = 480, 854, 8
dev = torch.device("cuda:0")
args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_bench", object_channel=None, eval_save=False, eval_export=False)
model = rcf_amd.RCFModel(args, **config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.1, norm="SyncBN"))
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
tr = rcf_amd.Trainer(model, device=dev, precision=prec)
nb = synth.make_batch(B, H, W, config_id=2)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
batch = {k: [t(x) for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
sites = collections.Counter()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if fr.filename.startswith(root) and "trace_copies" not in fr.filename:
            return f"{os.path.relpath(fr.filename, root)}:{fr.lineno} {fr.line}"
    return "?"


def wrap(obj, name):
    orig = getattr(obj, name)

    def f(*a, **k):
        sites[(name, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)


for n in ("copy_", "clone", "to", "contiguous", "float", "double", "item", "fill_", "zero_", "add_", "mul_"):
    wrap(torch.Tensor, n)
for n in ("tensor", "zeros", "ones", "cat", "full", "as_tensor", "from_numpy", "zeros_like", "ones_like"):
    wrap(torch, n)
tr.step(batch)
torch.cuda.synchronize()
for (name, s), c in sorted(sites.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c:4d}  {name:12s} {s[:150]}")
