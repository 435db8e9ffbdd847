"""Which conv direction moves the gradient NORMS of the early layers in the fp16-pair step?  (tools/grad_error_b8.py: the
shipped fp32 step's backbone gradient norm is 2.5e-3 above the float64 truth at 8 pairs of 480x854, the exact-fp32-MFMA
step and the oracle's fp32 are within 5e-5; the vector errors are the same 1.5e-2 for all three.)  Reference here: the
exact-fp32-MFMA step.  Runs: pairs in all directions; pairs in ONE direction (the others on bf16 triples); none.
usage: python tools/grad_bias_probe.py [B]"""
import copy, os, sys, types
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rcf_amd
from rcf_amd import config, synth, layers, ops, _lib

H, W = 480, 854
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
WATCH = ["backbone2.conv1.weight", "backbone2.layer1.0.conv1.weight", "backbone2.layer1.0.conv2.weight", "backbone2.layer1.0.downsample.0.weight",
         "backbone2.layer1.1.conv2.weight", "backbone2.layer1.2.conv2.weight", "backbone2.layer2.0.conv2.weight", "backbone2.layer2.3.conv2.weight",
         "backbone2.layer3.0.conv2.weight", "backbone2.layer3.5.conv2.weight", "backbone2.layer4.0.conv2.weight", "backbone2.layer4.2.conv2.weight",
         "decode_head2.convs.0.conv.weight", "decode_head2.convs.1.conv.weight", "decode_head3.convs.0.conv.weight"]


def run(noise=0.0, seed=0):
    kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, affine=False, norm="BN")
    kw.update(log_interval=10 ** 9, train_iter=1)
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_test", object_channel=None, eval_save=False, eval_export=False)
    m = rcf_amd.RCFModel(args, **copy.deepcopy(kw))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
    m.to("cuda").train()
    nb = synth.make_batch(B, H, W, config_id=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if noise:                                                     # relative perturbation of every image value (uniform in +-noise)
        r = np.random.RandomState(seed)
        nb["imgs"] = [(a * (1.0 + noise * r.uniform(-1, 1, size=a.shape))).astype(np.float32) for a in nb["imgs"]]
    b = {"imgs": [t(a) for a in nb["imgs"]], "gt_fw_flows": [t(a) for a in nb["gt_fw_flows"]], "gt_bw_flows": [t(a) for a in nb["gt_bw_flows"]],
         "seq_ids": nb["seq_ids"], "seq_names": nb["seq_names"], "paths": nb["paths"]}
    l = m(b)
    l["loss"].backward()
    torch.cuda.synchronize()
    g = {n: float(p.grad.double().norm()) for n, p in m.named_parameters() if p.grad is not None}
    mods = {}
    for n, v in g.items():
        mods[n.split(".")[0]] = mods.get(n.split(".")[0], 0.0) + v * v
    g.update({k: v ** 0.5 for k, v in mods.items()})
    del m
    torch.cuda.empty_cache()
    return float(l["loss"]), g


old = ops.set_conv_flags(_lib.CONV_FP32_MFMA(0))
ops.weights_changed()
ref_loss, ref = run()
ops.set_conv_flags(old)
ops.weights_changed()
print(f"reference (exact fp32 MFMA): loss {ref_loss:.9f}")
rows = ["backbone2", "decode_head", "decode_head2", "decode_head3"] + WATCH
table = {}
names = []


def record(name, g, loss):
    table[name] = {r: g[r] / ref[r] - 1.0 for r in rows}
    names.append(name)
    print(f"{name}: loss {loss:.9f}", flush=True)


if os.environ.get("PROBE", "kinds") == "spread":
    # how far do equally valid fp32 evaluations differ from each other?  The exact-fp32 kernels in their other three tuning variants
    # (K-step / LDS layout: another summation order), and the same kernels on images perturbed by one fp32 ulp / four ulps
    for v in (1, 2, 3):
        ops.set_conv_flags(_lib.CONV_FP32_MFMA(v)); ops.weights_changed()
        loss, g = run(); record(f"fp32 variant {v}", g, loss)
    ops.set_conv_flags(_lib.CONV_FP32_MFMA(0)); ops.weights_changed()
    for nz, sd in ((2.0 ** -24, 1), (2.0 ** -24, 2), (2.0 ** -22, 3), (2.0 ** -22, 4)):
        loss, g = run(nz, sd); record(f"fp32 img*(1+-2^{int(np.log2(nz))}) s{sd}", g, loss)
    ops.set_conv_flags(old); ops.weights_changed()
    layers.SCHED.planes = False
    loss, g = run(); record("pairs", g, loss)
    loss, g = run(2.0 ** -24, 1); record("pairs img 2^-24 s1", g, loss)
    loss, g = run(2.0 ** -24, 2); record("pairs img 2^-24 s2", g, loss)
else:
    cases = [("pairs fdw, planes", "fdw", True), ("pairs fdw", "fdw", False), ("pairs f only", "f", False), ("pairs d only", "d", False),
             ("pairs w only", "w", False), ("bf16 triples everywhere", "", False), ("pairs fd", "fd", False)]
    for name, kinds, planes in cases:
        rcf_amd.config.SCHED.h2_kinds = kinds
        layers.SCHED.planes = planes
        ops.weights_changed()
        loss, g = run()
        record(name, g, loss)
print("gradient norm / reference - 1:")
print(f"{'':46s}" + "".join(f"{n[:22]:>24s}" for n in names))
for r in rows:
    print(f"{r:46s}" + "".join(f"{table[n][r]:+24.2e}" for n in names))
