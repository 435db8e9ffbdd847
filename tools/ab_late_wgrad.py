"""experiment: second stream for the weight gradients, started beside this layer's data gradient (round 2's form) or after it,
beside the next layer's batch-norm backward; with both weight-gradient tiles.  usage: python tools/ab_late_wgrad.py"""
import os, sys, time, types
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import config, synth, layers, ops

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
configs = {"one stream, 256x256 wgrad": (False, False, 1), "one stream, 128x256 wgrad": (False, False, 0),
           "two streams (beside dgrad), 256x256": (True, False, 1), "two streams (beside dgrad), 128x256": (True, False, 0),
           "two streams, LATE wgrad, 256x256": (True, True, 1), "two streams, LATE wgrad, 128x256": (True, True, 0)}
for _ in range(4):
    tr.step(batch)
res = {k: [] for k in configs}
for r in range(3):
    for name, (ov, late, big) in configs.items():
        layers.OVERLAP_WGRAD, layers.LATE_WGRAD = ov, late
        ops.conv_set_wgrad_big(big)
        tr.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            tr.step(batch)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 6 * 1e3)
for name, v in res.items():
    print(f"{name:42s}: " + " ".join(f"{x:7.2f}" for x in v) + f"   median {np.median(v):7.2f} ms/step", flush=True)
