"""Many training steps on three alternating synthetic batches: the loss must stay finite and fall.
usage: python tools/stability_run.py [fp32|bf16|fp16|both|all] [steps] [field=value ...]   (fields of rcf_amd.config.SCHED, e.g. fold_bn=0)"""
import os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rcf_amd
from rcf_amd import config, synth, layers
which = sys.argv[1] if len(sys.argv) > 1 else "both"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
config.SCHED.parse(sys.argv[3:])
H, W, B = 480, 854, 8
dev = torch.device("cuda:0")
for prec in (("bf16", "fp32") if which == "both" else (("bf16", "fp32", "fp16") if which == "all" else (which,))):
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_bench", object_channel=None, eval_save=False, eval_export=False)
    model = rcf_amd.RCFModel(args, **config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.1, norm="SyncBN"))
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
    tr = rcf_amd.Trainer(model, device=dev, precision=prec)
    batches = []
    for j in range(3):
        nb = synth.make_batch(B, H, W, config_id=2, first_index=8 * j)
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        batches.append({k: [t(x) for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")})
    hist, first_bad = [], None
    for i in range(steps):
        l = float(tr.step(batches[i % 3])["loss"])
        if not np.isfinite(l) and first_bad is None:
            first_bad = i
            break
        if i % 5 == 0:
            hist.append(round(l, 3))
    print(prec, "loss every 5 steps:", hist, "| first non-finite step:", first_bad,
          "" if tr.scaler is None else f"| loss scale 2^{int(np.log2(tr.scaler.scale))}, {tr.scaler.skipped} skipped steps, {tr.step_count} optimizer steps", flush=True)
    del tr, model
    torch.cuda.empty_cache()
