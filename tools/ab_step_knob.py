"""Same-process, interleaved A/B of the training step (fp32 and bf16) over one library knob.
usage: python tools/ab_step_knob.py <knob> [rounds=3] [steps=6]     knob: wgrad_xcd | korder | colmap | bn_sweep"""
import os, sys, time, types
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import config, synth, ops

KNOBS = {"wgrad_xcd": (ops.conv_set_wgrad_xcd, {"grid order": 0, "XCD-aware": 1}, 1),
         "korder": (ops.conv_set_korder, {"tap outer": 0, "chunk outer": 1}, 1),
         "wgrad_big": (ops.conv_set_wgrad_big, {"128x256 x 2 per CU": 0, "256x256 x 1 per CU (fp16 pairs)": 1, "256x256 (both)": 3}, 1),
         "colmap": (ops.conv_set_colmap, {"row bands": 0, "byte model": 1}, 1),
         "bn_sweep": (ops.bn_set_sweep, {"front to back": 0, "cache aware >= 192 MB": 1, "cache aware, all": 2}, 1)}
knob = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
setter, settings, default = KNOBS[knob]
H, W, B = 480, 854, 8
dev = torch.device("cuda:0")
args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_bench", object_channel=None, eval_save=False, eval_export=False)
nb = synth.make_batch(B, H, W, config_id=2)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
batch = {k: [t(x) for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
for prec in ("fp32", "bf16"):
    model = rcf_amd.RCFModel(args, **config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.1, norm="SyncBN"))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
    tr = rcf_amd.Trainer(model, device=dev, precision=prec)
    for _ in range(4):
        tr.step(batch)
    res = {k: [] for k in settings}
    for r in range(rounds):
        for name, v in settings.items():
            setter(v)
            tr.step(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                tr.step(batch)
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / steps * 1e3)
    setter(default)
    for name, v in res.items():
        print(f"{prec} step, {knob} = {name:32s}: " + " ".join(f"{x:7.2f}" for x in v) + f"   median {np.median(v):7.2f} ms/step", flush=True)
    del tr, model
    torch.cuda.empty_cache()
