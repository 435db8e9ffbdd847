"""per-step wall time of the training step over many steps + allocator counters: does the step settle?
usage: python tools/step_trace.py [fp32|bf16] [steps=40]"""
import os, sys, time, types
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import config, synth

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
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
out = []
for i in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    st = torch.cuda.memory_stats()
    out.append(f"{i:3d} {ms:7.1f} ms  reserved {st['reserved_bytes.all.current']/2**30:6.2f} GiB  device mallocs {st.get('num_device_alloc', -1)} "
               f"frees {st.get('num_device_free', -1)}  alloc retries {st['num_alloc_retries']}")
print("\n".join(out))
