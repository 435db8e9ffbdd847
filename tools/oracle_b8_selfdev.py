"""How far do two equally valid fp32 evaluations of the oracle differ in the per-module gradient norms of one training
step at BASELINE configs[1]'s real batch (8 pairs of 480x854), and how far are they from the float64 evaluation of the same
step?  Run A: fp32, default threads; truth: float64; run B: fp32, half the threads (another blocking and summation order
inside the CPU conv / reduction kernels -- channels_last as run B was tried first and did not finish in 15 minutes on the
128-core host: its dilated convs take a slow path).  `gradnorm_fp32_err` = the worse of A and B against the truth, per module.  The result (tests/golden/oracle_b8_selfdev.json) is truth and yardstick of
tests/test_model_gpu.py::test_fullsize_b8_gradients_vs_oracle: the HIP step against `gradnorm_f64`, limit 4 x `gradnorm_fp32_err`,
floor 1e-4 (the same rule the small cases use with their `ref32_err_*` fixtures).
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


def build(cl=False):
    kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, affine=False, norm="BN")
    kw.update(log_interval=10 ** 9, train_iter=1)
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_test", object_channel=None, eval_save=False, eval_export=False)
    m = orc.RCFModel(args, **copy.deepcopy(kw))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
    return m.to(memory_format=torch.channels_last) if cl else m


def batch(double=False):
    nb = synth.make_batch(B, H, W, config_id=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double() if double else torch.from_numpy(np.ascontiguousarray(a))
    return {"imgs": [t(a) for a in nb["imgs"]], "gt_fw_flows": [t(a) for a in nb["gt_fw_flows"]],
            "gt_bw_flows": [t(a) for a in nb["gt_bw_flows"]], "seq_ids": nb["seq_ids"], "seq_names": nb["seq_names"], "paths": nb["paths"]}


def run(threads, double=False):
    import time
    torch.set_num_threads(threads)
    t0 = time.time()
    m = build()
    if double:
        m = m.double()
    m.train()
    l = m(batch(double))
    l["loss"].backward()
    gn = {}
    for n, p in m.named_parameters():
        if p.grad is not None:
            gn[n.split(".")[0]] = gn.get(n.split(".")[0], 0.0) + float(p.grad.double().pow(2).sum())
    print(f"run with {threads} threads{', float64' if double else ''}: {time.time() - t0:.0f} s", flush=True)
    return {k: float(v.detach()) for k, v in l.items()}, {k: v ** 0.5 for k, v in gn.items()}


def mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 2 ** 20
    return 0.0


dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "oracle_b8_selfdev.json")
NT = torch.get_num_threads()
out = {"B": B, "H": H, "W": W, "threads": [NT, max(1, NT // 2)]}


def save():
    json.dump(out, open(dst, "w"), indent=1)       # after every run: a time limit still leaves what was finished


print(f"{mem_available_gb():.0f} GB available, {NT} threads", flush=True)
la, ga = run(NT)
out.update(loss_a=la, gradnorm_a=ga)
save()
if mem_available_gb() > float(os.environ.get("FP64_NEEDS_GB", 24 * B)):      # ~2x the fp32 run's 45 GB at B = 8, with head room
    lt, truth = run(NT, True)
    out.update(loss_f64=lt, gradnorm_f64=truth, gradnorm_fp32_err={k: abs(ga[k] - truth[k]) / abs(truth[k]) for k in ga})
    save()
else:
    print(f"float64 run skipped: {mem_available_gb():.0f} GB available")
lb, gb = run(max(1, NT // 2))
out.update(loss_b=lb, gradnorm_b=gb, gradnorm_dev={k: abs(ga[k] - gb[k]) / abs(ga[k]) for k in ga})
if "gradnorm_f64" in out:
    t = out["gradnorm_f64"]
    out["gradnorm_fp32_err"] = {k: max(abs(ga[k] - t[k]), abs(gb[k] - t[k])) / abs(t[k]) for k in ga}
save()
print(json.dumps(out, indent=1))
