"""Two ranks on ONE GPU (gloo) against the single-process step on the global batch, per PARAMETER: where do the gradients part?
usage: python tools/two_rank_probe.py <fp32|bf16> [schedule_field=value ...]"""
import copy
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, B = 64, 96, 2


def build(prec, sched):
    sys.path.insert(0, ROOT)
    import rcf_amd
    from rcf_amd import config, synth
    config.SCHED.parse(sched)
    kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, norm="SyncBN")
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_dist", object_channel=None)
    m = rcf_amd.RCFModel(args, **copy.deepcopy(kw))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()})
    nb = synth.make_batch(B, H, W, config_id=1)
    return rcf_amd, m, nb


def batch(nb, sl):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a[sl])).to("cuda:0")
    return {k: [t(x) for x in nb[k]] for k in ("imgs", "gt_fw_flows", "gt_bw_flows")}


def worker(rank, world, port, q, prec, sched):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rcf_amd, m, nb = build(prec, sched)
    tr = rcf_amd.Trainer(m, device="cuda:0", precision=prec if prec == "bf16" else None)
    per = B // world
    tr.step(batch(nb, slice(rank * per, (rank + 1) * per)))
    torch.cuda.synchronize()
    if rank == 0:
        q.put({n: (p.grad.double() / world).cpu().contiguous().numpy() for n, p in m.named_parameters() if p.grad is not None})
    dist.barrier()
    dist.destroy_process_group()


def single(prec, sched, sl):
    rcf_amd, m, nb = build(prec, sched)
    tr = rcf_amd.Trainer(m, device="cuda:0", precision=prec if prec == "bf16" else None)
    losses = tr.step(batch(nb, sl))
    return float(losses["loss"]), {n: p.grad.double().cpu().contiguous() for n, p in m.named_parameters() if p.grad is not None}


def vec_err(a, b):
    num = sum(float((a[n] - b[n]).pow(2).sum()) for n in b) ** 0.5
    return num / sum(float(b[n].pow(2).sum()) for n in b) ** 0.5


if __name__ == "__main__":
    prec, sched = sys.argv[1], sys.argv[2:]
    if sched and sched[0] == "--selfcheck":
        # is the single-process step reproducible, and how far apart are the steps on pair 0, pair 1 and both pairs?
        sched = sched[1:]
        l0, g0 = single(prec, sched, slice(0, B))
        l1, g1 = single(prec, sched, slice(0, B))
        la, ga = single(prec, sched, slice(0, 1))
        lb, gb = single(prec, sched, slice(1, 2))
        avg = {n: 0.5 * (ga[n] + gb[n]) for n in g0}
        print(f"{prec} {sched}: same step twice: losses {l0} {l1}, gradient vector difference {vec_err(g1, g0):.2e}; "
              f"mean of the two one-pair steps (LOCAL statistics) against the two-pair step: loss {(0.5 * (la + lb) - l0) / l0:.2e}, vector {vec_err(avg, g0):.2e}")
        sys.exit(0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29999, q, prec, sched)) for r in range(2)]
    for p in procs:
        p.start()
    g2 = {k: torch.from_numpy(v) for k, v in q.get(timeout=600).items()}
    for p in procs:
        p.join(timeout=120)
    rcf_amd, m, nb = build(prec, sched)
    tr = rcf_amd.Trainer(m, device="cuda:0", precision=prec if prec == "bf16" else None)
    tr.step(batch(nb, slice(0, B)))
    rows = []
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        a, b = g2[n], p.grad.double().cpu().contiguous()
        rows.append((float((a - b).norm() / (b.norm() + 1e-30)), float(b.norm()), n))
    rows.sort(reverse=True)
    tot2 = sum(float(v.pow(2).sum()) for v in g2.values()) ** 0.5
    tot1 = sum(float(p.grad.double().pow(2).sum()) for p in m.parameters() if p.grad is not None) ** 0.5
    print(f"{prec} {sched}: total gradient norm two ranks {tot2:.6e} single {tot1:.6e} rel {abs(tot2 - tot1) / tot1:.2e}")
    by = {n: e for e, _, n in rows}
    order = [n for n, _ in m.named_parameters() if n in by and n.startswith("backbone2.") and n.endswith((".weight",)) and ".bn" in n]
    print("  bn weights in forward order: " + " ".join(f"{n.replace('backbone2.', '').replace('.weight', '')}:{by[n]:.1e}" for n in order))
    heads = [n for n in by if not n.startswith("backbone2.")]
    print("  worst outside the backbone: " + " ".join(f"{n}:{by[n]:.1e}" for n in sorted(heads, key=lambda k: -by[k])[:4]))
