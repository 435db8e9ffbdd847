"""Runs `steps` training steps of one precision ("fp32" | "bf16") at the bench geometry -- the program to put behind
`rocprofv3 --kernel-trace --stats --` when only one leg of bench.py is wanted.
usage: python tools/step_prof.py bf16 [steps=4] [pairs=8] [schedule_field=value ...]   (rcf_amd.config.Schedule fields, e.g. fold_bn=0)"""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import config, synth

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
config.SCHED.parse(sys.argv[4:])
H, W = 480, 854
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
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(batch)
torch.cuda.synchronize()
print(f"{prec}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step over {steps} steps (+3 untimed)")
