"""How long the host needs to ENQUEUE one training step (no synchronisation inside the step) against how long the GPU needs
to run it: tells how far the kernels can speed up before the step becomes launch-bound.
usage: python tools/host_time.py [fp32|bf16] [pairs=8]"""
import os, sys, time, types
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import config, synth, _lib

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
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
# count library calls per step
n_calls = [0]
orig = _lib.call
def counting(name, *a):
    n_calls[0] += 1
    return orig(name, *a)
_lib.call = counting
for m in (rcf_amd.ops, rcf_amd.layers if hasattr(rcf_amd, "layers") else None):
    if m is not None and hasattr(m, "call"):
        m.call = counting
tr.step(batch); torch.cuda.synchronize()
print("library calls per step (through _lib.call):", n_calls[0])
_lib.call = orig
for m in (rcf_amd.ops,):
    if hasattr(m, "call"):
        m.call = orig
N = 6
host, total = [], []
for _ in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = tr.step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print(f"{prec}: host enqueue {np.median(host):.1f} ms (min {min(host):.1f}), step wall {np.median(total):.1f} ms")
