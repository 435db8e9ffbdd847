"""A/B of the training step's schedule and operand format in ONE process (boxes differ by more than the effects): the weight
gradients on the main stream / on a second stream beside the layer's data gradient / started after it (layers.SCHED.late_wgrad), with
the fp16 pair planes (layers.SCHED.planes) on and off.  Interleaved rounds, median of 3 x 6 steps.
usage: python tools/ab_schedule.py [fp32|bf16] [bnsums|joins|defer|prio]   (defer: layers.SCHED.defer_residual on / off; joins: which join outputs are also written as planes; bnsums: the default schedule and the one-stream order with the batch-norm
backward sums from the data gradients' epilogues (layers.SCHED.fuse_bn_bwd) on and off)"""
import os, sys, time, types
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import config, synth, layers

H, W, B = 480, 854, 8
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda:0")
args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_bench", object_channel=None, eval_save=False, eval_export=False)
model = rcf_amd.RCFModel(args, **config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.1, norm="SyncBN"))
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
tr = rcf_amd.Trainer(model, device=dev, precision=prec)
nb = synth.make_batch(B, H, W, config_id=2)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
batch = {k: [t(x) for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}
configs = {}
for planes in ((True, False) if prec == "fp32" else (False,)):
    for name, ov, late in (("one stream", False, False), ("two streams, beside dgrad", True, False), ("two streams, LATE", True, True)):
        configs[f"{name}, planes {'on' if planes else 'off'}"] = (ov, late, planes, layers.SCHED.fuse_bn_bwd)
if len(sys.argv) > 2 and sys.argv[2] == "bnsums":
    configs = {}
    for name, ov, late in (("one stream", False, False), ("two streams, LATE", True, True)):
        for fuse in (True, False):
            configs[f"{name}, bn sums {'from the dgrad epilogue' if fuse else 'by the reduction pass'}"] = (ov, late, prec == "fp32", fuse)
if len(sys.argv) > 2 and sys.argv[2] == "defer":
    # the identity branches' gradients added in conv1's data-gradient epilogue (layers.SCHED.defer_residual) or written by the join's
    # batch-norm backward and accumulated onto
    configs = {f"{name}, {'deferred residual gradients' if d else 'identity gradients written'}": (ov, late, prec == "fp32", layers.SCHED.fuse_bn_bwd, d)
               for name, ov, late in (("one stream", False, False), ("two streams, LATE", True, True)) for d in (True, False)}
prio = len(sys.argv) > 2 and sys.argv[2] == "prio"
if prio:
    # HIP priority of the second stream (the weight gradients): 0 = the default, 1 = low, -1 = high (layers._side_stream, SCHED.side_priority)
    configs = {f"two streams, LATE, second stream priority {p}": (True, True, prec == "fp32", layers.SCHED.fuse_bn_bwd, p) for p in (0, 1, -1)}
joins = len(sys.argv) > 2 and sys.argv[2] == "joins"
if joins:
    # which tensors exist as planes beside their fp32 copy: the bottleneck joins (all / only in front of a stage's first block) and
    # the last stage's output for the decode heads; default schedule
    configs = {f"joins {j}, heads {'on' if h else 'off'}": (True, True, True, layers.SCHED.fuse_bn_bwd, j, h)
               for j in ("all", "stage") for h in (True, False)}
for _ in range(4):
    tr.step(batch)
res = {k: [] for k in configs}
for r in range(int(os.environ.get("AB_ROUNDS", "3"))):
    for name, cfg in configs.items():
        ov, late, planes, fuse = cfg[:4]
        layers.SCHED.overlap_wgrad, layers.SCHED.late_wgrad, layers.SCHED.planes, layers.SCHED.fuse_bn_bwd = ov, late, planes, fuse
        if prio:
            torch.cuda.synchronize()
            layers.SCHED.side_priority = int(cfg[4])
            layers._side_streams.clear()
        elif joins:
            layers.SCHED.join_planes, model.backbone2.heads_take_planes = cfg[4], cfg[5]
        elif len(cfg) == 5:
            layers.SCHED.defer_residual = cfg[4]
        tr.step(batch)
        tr.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            tr.step(batch)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 6 * 1e3)
for name, v in res.items():
    print(f"{name:60s}: " + " ".join(f"{x:7.2f}" for x in v) + f"   median {np.median(v):7.2f} ms/step", flush=True)
