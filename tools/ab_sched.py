"""Same-process A/B of the training step under two settings of ONE config.SCHED field (interleaved rounds, median of AB_ROUNDS x 6
steps); prints ms per step and the loss of a step taken from the same weights under each setting.
usage: python tools/ab_sched.py <fp32|bf16> <field> <valueA> <valueB> [pairs]
(field `conv_flags`: the RCF_CONV_* bits OR-ed into every conv launch, e.g. 0 against 0x100 = RCF_CONV_NO_THIN)"""
import copy
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import config, ops, synth


def apply(field, v):
    if field == "conv_flags":
        ops.set_conv_flags(int(v, 0))
    else:
        config.SCHED.parse([f"{field}={v}"])

prec, field, va, vb = sys.argv[1:5]
B = int(sys.argv[5]) if len(sys.argv) > 5 else 8
H, W = 480, 854
dev = torch.device("cuda:0")
args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_bench", object_channel=None, eval_save=False, eval_export=False)
model = rcf_amd.RCFModel(args, **config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, norm="SyncBN"))
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
tr = rcf_amd.Trainer(model, device=dev, precision=prec)
nb = synth.make_batch(B, H, W, config_id=2)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
batch = {k: [t(x) for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
settings = {f"{field}={va}": va, f"{field}={vb}": vb}
res = {k: [] for k in settings}
for name, v in settings.items():
    apply(field, v)
    for _ in range(3):
        tr.step(batch)
for r in range(int(os.environ.get("AB_ROUNDS", "3"))):
    for name, v in settings.items():
        apply(field, v)
        tr.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            tr.step(batch)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 6 * 1e3)
print(f"{prec} step, {B} pairs {H}x{W}")
for name in settings:
    v = sorted(res[name])
    print(f"  {name:32s} {v[len(v) // 2]:8.2f} ms/step  (rounds: {', '.join(f'{x:.2f}' for x in res[name])})")
