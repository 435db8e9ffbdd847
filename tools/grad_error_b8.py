"""Where does the HIP step's gradient differ from the float64 truth at BASELINE configs[1]'s real batch (8 pairs of 480x854)?
Per parameter: |g - truth| / |truth| of (a) the HIP fp32 step as shipped, (b) with SCHED.planes off, (c) on the exact fp32 matrix-core
kernels, and (d) of the oracle's own fp32 evaluation -- and the same for the per-module norms the parity test uses.
Needs a GPU, ~100 GB of host memory and ~5 minutes on the GPU box's host (usage: python tools/grad_error_b8.py [B] [out.txt])."""
import copy, os, sys, time, types
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rcf_torch as orc
import rcf_amd
from rcf_amd import config, synth, layers, ops, _lib

H, W = 480, 854
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
OUT = open(sys.argv[2], "w") if len(sys.argv) > 2 else None


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    if OUT:
        OUT.write(s + "\n"); OUT.flush()


def build(cls, device):
    kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, affine=False, norm="BN")
    kw.update(log_interval=10 ** 9, train_iter=1)
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_test", object_channel=None, eval_save=False, eval_export=False)
    m = cls(args, **copy.deepcopy(kw))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
    return m.to(device)


def batch(device, double=False):
    nb = synth.make_batch(B, H, W, config_id=1)
    t = lambda a: (torch.from_numpy(np.ascontiguousarray(a)).double() if double else torch.from_numpy(np.ascontiguousarray(a))).to(device)
    return {"imgs": [t(a) for a in nb["imgs"]], "gt_fw_flows": [t(a) for a in nb["gt_fw_flows"]],
            "gt_bw_flows": [t(a) for a in nb["gt_bw_flows"]], "seq_ids": nb["seq_ids"], "seq_names": nb["seq_names"], "paths": nb["paths"]}


def grads(m):
    return {n: p.grad.detach().double().cpu().numpy().ravel() for n, p in m.named_parameters() if p.grad is not None}


def hip_run():
    m = build(rcf_amd.RCFModel, "cuda")
    m.train()
    l = m(batch("cuda"))
    l["loss"].backward()
    torch.cuda.synchronize()
    g = grads(m)
    del m
    torch.cuda.empty_cache()
    return float(l["loss"]), g


def oracle_run(double):
    t0 = time.time()
    m = build(orc.RCFModel, "cpu")
    if double:
        m = m.double()
    m.train()
    l = m(batch("cpu", double))
    l["loss"].backward()
    say(f"# oracle {'float64' if double else 'fp32'}: {time.time() - t0:.0f} s")
    return float(l["loss"].detach()), grads(m)


runs = {}
runs["hip"] = hip_run()
layers.SCHED.planes = False
runs["hip_noplanes"] = hip_run()
layers.SCHED.planes = True
old = ops.set_conv_flags(_lib.CONV_FP32_MFMA(0))
ops.weights_changed()
runs["hip_fp32mfma"] = hip_run()
ops.set_conv_flags(old)
ops.weights_changed()
runs["oracle_fp32"] = oracle_run(False)
truth_loss, truth = oracle_run(True)

names = list(truth.keys())
say(f"# B={B} {H}x{W}; loss truth {truth_loss:.9f}; " + " ".join(f"{k} {v[0]:.9f}" for k, v in runs.items()))
mods = sorted({n.split(".")[0] for n in names})
say("# per-module gradient NORM error vs float64 | per-module gradient VECTOR error |g - truth| / |truth|")
for mod in mods:
    ns = [n for n in names if n.split(".")[0] == mod]
    tn = np.sqrt(sum(float((truth[n] ** 2).sum()) for n in ns))
    line = f"{mod:14s} |truth| {tn:12.6f}"
    for k, (_, g) in runs.items():
        gn = np.sqrt(sum(float((g[n] ** 2).sum()) for n in ns))
        ev = np.sqrt(sum(float(((g[n] - truth[n]) ** 2).sum()) for n in ns))
        line += f" | {k} norm {abs(gn - tn) / tn:.2e} vec {ev / tn:.2e}"
    say(line)
say("# per parameter, sorted by the shipped step's share of its module's squared vector error: |g - truth| / |truth| per run, "
    "(|g|^2 - |truth|^2) / |module truth|^2 of the shipped step")
rows = []
for n in names:
    mod = n.split(".")[0]
    tn2 = sum(float((truth[m_] ** 2).sum()) for m_ in names if m_.split(".")[0] == mod)
    t = truth[n]
    e = {k: float(np.linalg.norm(g[n] - t)) / max(float(np.linalg.norm(t)), 1e-300) for k, (_, g) in runs.items()}
    share = float(((runs["hip"][1][n] - t) ** 2).sum()) / tn2
    dn2 = (float((runs["hip"][1][n] ** 2).sum()) - float((t ** 2).sum())) / tn2
    rows.append((share, n, t.size, float(np.linalg.norm(t)), e, dn2))
rows.sort(reverse=True)
for share, n, size, tn, e, dn2 in rows[:40]:
    say(f"{n:52s} n {size:8d} |truth| {tn:10.4e} share {share:.2e} dnorm2 {dn2:+.2e} " + " ".join(f"{k} {v:.2e}" for k, v in e.items()))
say("# the same, sorted by dnorm2 magnitude (who moves the NORM)")
rows.sort(key=lambda r: -abs(r[5]))
for share, n, size, tn, e, dn2 in rows[:25]:
    say(f"{n:52s} n {size:8d} |truth| {tn:10.4e} share {share:.2e} dnorm2 {dn2:+.2e} " + " ".join(f"{k} {v:.2e}" for k, v in e.items()))
