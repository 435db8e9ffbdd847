"""Same-process A/B of the bf16 step with the folded conv3 -> bn3 -> join (layers.SCHED.fold_bn) on and off, and of the masked data
gradient (layers.SCHED.fold_masked_dgrad).  Interleaved rounds, median of AB_ROUNDS x 6 steps; prints ms per step and the losses.
usage: python tools/ab_fold.py [pairs]"""
import os, sys, time, types
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import config, synth, layers

H, W = 480, 854
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_bench", object_channel=None, eval_save=False, eval_export=False)
model = rcf_amd.RCFModel(args, **config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.1, norm="SyncBN"))
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
tr = rcf_amd.Trainer(model, device=dev, precision="bf16")
nb = synth.make_batch(B, H, W, config_id=2)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
batch = {k: [t(x) for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
configs = {"three-pass": (False, False), "fold, mask pass": (True, False), "fold, masked dgrad": (True, True)}
if os.environ.get("AB_ONE_STREAM"):
    layers.SCHED.overlap_wgrad = False
res = {k: [] for k in configs}
loss = {}
for name, (f, m) in configs.items():
    layers.SCHED.fold_bn, layers.SCHED.fold_masked_dgrad = f, m
    for _ in range(3):
        l = tr.step(batch)
    loss[name] = float(l["loss"])
for r in range(int(os.environ.get("AB_ROUNDS", "3"))):
    for name, (f, m) in configs.items():
        layers.SCHED.fold_bn, layers.SCHED.fold_masked_dgrad = f, m
        tr.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            tr.step(batch)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 6 * 1e3)
print(f"bf16 step, {B} pairs {H}x{W}, second stream {layers.SCHED.overlap_wgrad}")
for name in configs:
    v = sorted(res[name])
    print(f"  {name:24s} {v[len(v) // 2]:8.2f} ms/step  (rounds: {', '.join(f'{x:.2f}' for x in res[name])})  loss after 3 steps {loss[name]:.6f}")
