"""A few stage-2.1 training steps (the stage-1 step + EMA teacher forward + CRF on 2B frames + EMA update: BASELINE configs[3]) at the
bench geometry -- the program to put behind `rocprofv3 --kernel-trace --stats --` / tools/step_timeline.py.
usage: python tools/stage2_prof.py [fp32|bf16] [steps=3] [pairs=8] [crf_iters=5] [schedule_field=value ...]"""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd  # noqa
from rcf_amd import config, synth

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
config.SCHED.parse(sys.argv[5:])
H, W = 480, 854
dev = torch.device("cuda:0")
args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_bench", object_channel=1, eval_save=False, eval_export=False)
model = rcf_amd.RCFModel(args, **config.stage21_model_kwargs(config.mask_size_for(H, W), dropout=0.1, norm="BN", refine_iters=iters))
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
tr = rcf_amd.Trainer(model, lr=1e-4, weight_decay=1e-4, device=dev, precision=None if prec == "fp32" else prec)
nb = synth.make_batch(B, H, W, config_id=2)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
batch = {"imgs": [t(x) for x in nb["imgs"]], "gt_fw_flows": [t(x) for x in nb["gt_fw_flows"]], "gt_bw_flows": [t(x) for x in nb["gt_bw_flows"]],
         "seq_ids": nb["seq_ids"], "seq_names": nb["seq_names"], "paths": nb["paths"]}
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(batch)
torch.cuda.synchronize()
print(f"stage 2.1 {prec}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step over {steps} steps (+3 untimed), CRF T={iters}")
