"""Which tensors still get a standalone range pass (ops.absmax) in a training step?  usage: python tools/absmax_trace.py [fp32]"""
import os, sys, types, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import config, synth, ops
H, W, B = 480, 854, 8
dev = torch.device("cuda:0")
args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_bench", object_channel=None, eval_save=False, eval_export=False)
model = rcf_amd.RCFModel(args, **config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.1, norm="SyncBN"))
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
tr = rcf_amd.Trainer(model, device=dev, precision=sys.argv[1] if len(sys.argv) > 1 else "fp32")
nb = synth.make_batch(B, H, W, config_id=2)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
batch = {k: [t(x) for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
for _ in range(3):
    tr.step(batch)
calls = collections.Counter()
orig = ops.absmax
import traceback
def traced(x, *a, **k):
    fr = [f for f in traceback.extract_stack()[:-1] if "rcf" in f.filename][-3:]
    calls[(tuple(x.shape), " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr))] += 1
    return orig(x, *a, **k)
ops.absmax = traced
import rcf_amd.layers as L
tr.step(batch)
ops.absmax = orig
for (shape, where), n in sorted(calls.items(), key=lambda kv: -np.prod(kv[0][0])):
    print(f"{n} x {shape} {np.prod(shape)*4/1e6:8.1f} MB  {where}")
