"""How far do two equally valid fp32 evaluations of the oracle differ in the per-module gradient norms of one training
step at BASELINE configs[1]'s real batch (8 pairs of 480x854)?  Run A: default threads, contiguous convs; run B:
channels_last convs.  The result (tests/golden/oracle_b8_selfdev.json) is the yardstick of
tests/test_model_gpu.py::test_fullsize_b8_gradients_vs_oracle: limit = 4 x this deviation, floor 1e-4.
Needs ~45 GB of host memory and a few minutes on a many-core host (usage: python tools/oracle_b8_selfdev.py [out.json]; the
committed fixture was produced on the GPU box's host, 128 cores)."""
import copy, json, os, sys, types
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rcf_torch as orc
from rcf_amd import config, synth

H, W, B = 480, 854, int(os.environ.get("B", "8"))


def build(cl):
    kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, affine=False, norm="BN")
    kw.update(log_interval=10 ** 9, train_iter=1)
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_test", object_channel=None, eval_save=False, eval_export=False)
    m = orc.RCFModel(args, **copy.deepcopy(kw))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
    return m.to(memory_format=torch.channels_last) if cl else m


def batch():
    nb = synth.make_batch(B, H, W, config_id=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    return {"imgs": [t(a) for a in nb["imgs"]], "gt_fw_flows": [t(a) for a in nb["gt_fw_flows"]],
            "gt_bw_flows": [t(a) for a in nb["gt_bw_flows"]], "seq_ids": nb["seq_ids"], "seq_names": nb["seq_names"], "paths": nb["paths"]}


def run(cl):
    m = build(cl)
    m.train()
    l = m(batch())
    l["loss"].backward()
    gn = {}
    for n, p in m.named_parameters():
        if p.grad is not None:
            gn[n.split(".")[0]] = gn.get(n.split(".")[0], 0.0) + float(p.grad.double().pow(2).sum())
    return {k: float(v) for k, v in l.items()}, {k: v ** 0.5 for k, v in gn.items()}


la, ga = run(False)
lb, gb = run(True)
out = {"B": B, "H": H, "W": W, "threads": torch.get_num_threads(), "loss_a": la, "loss_b": lb, "gradnorm_a": ga, "gradnorm_b": gb,
       "gradnorm_dev": {k: abs(ga[k] - gb[k]) / abs(ga[k]) for k in ga}}
print(json.dumps(out, indent=1))
dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "oracle_b8_selfdev.json")
json.dump(out, open(dst, "w"), indent=1)
