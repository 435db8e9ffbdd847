"""Training step at unusual geometries / batch sizes in both precisions: finite losses, finite gradients, eval forward works.
usage: python tools/fuzz_shapes.py"""
import os, sys, types, traceback
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rcf_amd
from rcf_amd import config, synth
dev = torch.device("cuda:0")
cases = [(96, 160, 1), (100, 164, 3), (128, 128, 2), (250, 330, 1), (97, 161, 2), (384, 384, 3), (480, 854, 1), (64, 64, 5)]
bad = 0
for (H, W, B) in cases:
    for prec in ("fp32", "bf16", "fp16"):
        try:
            args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_fuzz", object_channel=None, eval_save=False, eval_export=False)
            model = rcf_amd.RCFModel(args, **config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.1, norm="BN"))
            shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
            model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
            # fp16: a loss scale at which this net's gradients are finite from the first step (the scaler's walk is tests/test_fp16_gpu.py's)
            tr = rcf_amd.Trainer(model, device=dev, precision=prec, loss_scaler=rcf_amd.trainer.LossScaler(2.0 ** 8) if prec == "fp16" else None)
            nb = synth.make_batch(B, H, W, config_id=3)
            t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
            batch = {k: [t(x) for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
            ls = [float(tr.step(batch)["loss"]) for _ in range(3)]
            g = tr.fp.grad
            ok = all(np.isfinite(ls)) and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0 and (tr.scaler is None or tr.scaler.skipped == 0)
            model.eval()
            with torch.no_grad():
                m = model({"imgs": [batch["imgs"][0]]})
            ok = ok and bool(torch.isfinite(m).all()) and tuple(m.shape[0:2]) == (B, 4)
            print(f"{H}x{W} B={B} {prec}: losses {[round(v, 4) for v in ls]} eval masks {tuple(m.shape)} {'ok' if ok else 'BAD'}", flush=True)
            bad += 0 if ok else 1
            del tr, model
        except Exception as e:                                  # noqa: BLE001
            bad += 1
            print(f"{H}x{W} B={B} {prec}: EXCEPTION {type(e).__name__}: {str(e)[:300]}", flush=True)
            traceback.print_exc(limit=3)
        torch.cuda.empty_cache()
print("failures:", bad)
