"""Truth and yardstick of tests/test_model_gpu.py::test_fullsize_b8_gradients_vs_oracle: the per-module gradient norms of one
training step at BASELINE configs[1]'s real batch (8 pairs of 480x854) from the oracle in FLOAT64, and how far equally valid
fp32 evaluations of the same oracle land from that truth:
  run A      fp32 as is;
  run P1, P2 fp32 on images perturbed by ONE fp32 ulp (every value times 1 + u 2^-24, u uniform in [-1, 1], seeds 1 and 2) --
             inputs no fp32 implementation can tell apart from the originals.
`gradnorm_fp32_spread` = the largest |norm - truth| / truth over A, P1, P2, per module; `vector_fp32_err` = the largest relative
VECTOR error |g - truth| / |truth| of those runs per module, estimated from `rcf_amd.synth.grad_sketch` fingerprints (32 signed sums
per parameter tensor; `sketch_f64` is the truth's fingerprint, what the test compares the HIP step's fingerprint with).  Why perturbations and not thread
counts or memory formats: half the threads moves the norms by 3e-7 (same blocking, nearly the same roundings: not an independent
sample; recorded as `gradnorm_dev` by an earlier version of this script), channels_last did not finish in 15 minutes on the
128-core host -- while the step itself is ill-conditioned: EVERY fp32 evaluation's gradient VECTOR is 1.5e-2 from the float64
one (tools/grad_error_b8.py) and the exact-fp32 HIP kernels under the same one-ulp perturbations, or in another summation order,
move the backbone's norm by 1-2e-3 (tools/grad_bias_probe.py, profiles/r04_grad_spread_probe.txt).  One sample (run A alone: 5e-5)
says nothing about that spread.
Needs ~100 GB of host memory and ~8 minutes on a many-core host (usage: python tools/oracle_b8_selfdev.py [out.json]; the
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


def batch(double=False, noise_seed=0):
    nb = synth.make_batch(B, H, W, config_id=1)
    if noise_seed:
        r = np.random.RandomState(noise_seed)
        nb["imgs"] = [(a * (1.0 + 2.0 ** -24 * r.uniform(-1, 1, size=a.shape))).astype(np.float32) for a in nb["imgs"]]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double() if double else torch.from_numpy(np.ascontiguousarray(a))
    return {"imgs": [t(a) for a in nb["imgs"]], "gt_fw_flows": [t(a) for a in nb["gt_fw_flows"]],
            "gt_bw_flows": [t(a) for a in nb["gt_bw_flows"]], "seq_ids": nb["seq_ids"], "seq_names": nb["seq_names"], "paths": nb["paths"]}


SKETCH = [None]        # of the last run


def run(threads, double=False, noise_seed=0):
    import time
    torch.set_num_threads(threads)
    t0 = time.time()
    m = build()
    if double:
        m = m.double()
    m.train()
    l = m(batch(double, noise_seed))
    l["loss"].backward()
    gn = {}
    for n, p in m.named_parameters():
        if p.grad is not None:
            gn[n.split(".")[0]] = gn.get(n.split(".")[0], 0.0) + float(p.grad.double().pow(2).sum())
    SKETCH[0] = synth.grad_sketch({n: p.grad for n, p in m.named_parameters() if p.grad is not None}, k=32)
    print(f"run with {threads} threads{', float64' if double else ''}{f', images perturbed (seed {noise_seed})' if noise_seed else ''}: "
          f"{time.time() - t0:.0f} s", flush=True)
    return {k: float(v.detach()) for k, v in l.items()}, {k: v ** 0.5 for k, v in gn.items()}


def mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 2 ** 20
    return 0.0


dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "oracle_b8_selfdev.json")
NT = torch.get_num_threads()
out = {"B": B, "H": H, "W": W, "threads": NT, "sketch_k": 32}


def save():
    json.dump(out, open(dst, "w"), indent=1)       # after every run: a time limit still leaves what was finished


print(f"{mem_available_gb():.0f} GB available, {NT} threads", flush=True)
if mem_available_gb() < float(os.environ.get("FP64_NEEDS_GB", 24 * B)):      # ~2x the fp32 run's 45 GB at B = 8, with head room
    sys.exit(f"the float64 run needs ~{12 * B} GB: {mem_available_gb():.0f} GB available")
lt, truth = run(NT, True)
sk_truth = SKETCH[0]
out.update(loss_f64=lt, gradnorm_f64=truth, sketch_f64=sk_truth)
save()
mods = sorted(truth)
fp32_runs = {}
for tag, seed in (("A", 0), ("P1", 1), ("P2", 2)):
    l, g = run(NT, False, seed)
    fp32_runs[tag] = dict(loss=l, gradnorm=g, norm_err={k: abs(g[k] - truth[k]) / truth[k] for k in mods},
                          vector_err={k: synth.sketch_error(SKETCH[0], sk_truth, k + ".") for k in mods})
    out["fp32_runs"] = fp32_runs
    out["gradnorm_fp32_spread"] = {k: max(r["norm_err"][k] for r in fp32_runs.values()) for k in mods}
    out["vector_fp32_err"] = {k: max(r["vector_err"][k] for r in fp32_runs.values()) for k in mods}
    save()
print(json.dumps({k: v for k, v in out.items() if k != "sketch_f64"}, indent=1))
