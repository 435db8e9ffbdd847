#!/usr/bin/env python3
"""A REFERENCE-derived gradient fixture at the real batch size (the 8 x 480x854 gradient fixture,
oracle_b8_selfdev.json, is made by the ORACLE and cannot catch an error the oracle and the kernels share).

The reference model itself (imported from /root/reference with make_golden.py's stand-in modules) takes one stage-1 training step
on 8 pairs -- 16 frames through the backbone, batch-norm statistics over all of them, configs/rcf/rcf_stage1.yaml:4 -- at 192x320,
the largest geometry at which the float64 truth of 8 pairs fits this container's memory (480x854 needs ~200 GB).  Stored: the
reference's fp32 losses, per-module gradient norms and `synth.grad_sketch` fingerprints (32 signed sums per parameter tensor) of
  * the float64 truth (the oracle in double, asserted equal to the reference in fp32 right here),
  * the reference's fp32 gradients, evaluated five ways (8 threads, 1 thread, channels_last convolutions, parameters moved by one
    unit in the last place with two seeds -- make_golden.py's set): the worst of their distances from the truth is the yardstick
    tests/test_model_gpu.py::test_b8_gradients_vs_reference holds the HIP step to.

Run in the build container only:  python tests/golden/make_golden_b8.py
"""
import copy
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402

H, W, B, K = 192, 320, 8, 32


def main():
    mg.install_standins()
    sys.path.insert(0, mg.REF)
    import models as ref_models                            # noqa: the reference itself
    sys.path.insert(0, mg.ROOT)
    sys.path.insert(0, os.path.join(mg.ROOT, "oracle"))
    import rcf_torch as orc
    import rcf_amd                                         # noqa
    from rcf_amd import config, synth
    torch.manual_seed(0)
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_golden", object_channel=None, eval_save=False, eval_export=False)
    kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, norm="BN")
    kw.update(log_interval=10 ** 9, train_iter=1)
    probe = ref_models.RCFModel(args, **copy.deepcopy(kw))
    shapes = {k: tuple(v.shape) for k, v in probe.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()}
    del probe
    nb = synth.make_batch(B, H, W, config_id=1)

    def run(cls, double, nthreads, cl=False, ulp_seed=0):
        torch.set_num_threads(nthreads)
        m = cls(args, **copy.deepcopy(kw))
        sdl = sd
        if ulp_seed:                                       # parameters moved by one unit in the last place (make_golden.py): what any
            gp = torch.Generator().manual_seed(1000 + ulp_seed)      # backward-stable fp32 implementation is allowed to return
            sdl = {k: (v * (1 + (torch.randint(0, 2, v.shape, generator=gp).float() * 2 - 1) * 2.0 ** -23)
                       if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")) else v)
                   for k, v in sd.items()}
        m.load_state_dict(sdl)
        if cl:
            m = m.to(memory_format=torch.channels_last)
        b = mg.torch_batch(nb)
        if double:
            m = m.double()
            b = {k: ([t.double() for t in v] if k in ("imgs", "gt_fw_flows", "gt_bw_flows") else v) for k, v in b.items()}
        m.train()
        l = m(b)
        l["loss"].backward()
        torch.set_num_threads(8)
        grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
        return {k: float(v) for k, v in l.items() if "loss" in k}, mg.grad_norms(m), synth.grad_sketch(grads, k=K)

    l_ref, g_ref, s_ref = run(ref_models.RCFModel, False, 8)
    print("reference fp32:", l_ref, g_ref, flush=True)
    l_ora, g_ora, s_ora = run(orc.RCFModel, False, 8)
    chk = {k: mg.rel(l_ora[k], l_ref[k]) for k in l_ref}
    chk.update({"gradnorm." + k: mg.rel(g_ora[k], g_ref[k]) for k in g_ref})
    chk["sketch"] = synth.sketch_error(s_ora, s_ref)
    print("oracle vs reference (fp32):", json.dumps(chk), flush=True)
    assert max(v for k, v in chk.items() if k.startswith("loss")) < 2e-5 and chk["sketch"] < 2e-2, chk
    l64, g64, s64 = run(orc.RCFModel, True, 8)
    print("float64:", l64, g64, flush=True)
    mods = sorted(g_ref)
    # the reference's own fp32 evaluations: as is, one thread, channels_last convolutions, parameters moved by one ulp (two seeds)
    evals = [("8 threads", g_ref, s_ref)]
    for tag, a in (("1 thread", dict(nthreads=1)), ("channels_last", dict(nthreads=8, cl=True)), ("ulp 1", dict(nthreads=8, ulp_seed=1)),
                   ("ulp 2", dict(nthreads=8, ulp_seed=2))):
        _, g_v, s_v = run(ref_models.RCFModel, False, **a)
        evals.append((tag, g_v, s_v))
    per = {tag: {k: synth.sketch_error(s_v, s64, k + ".") for k in mods} for tag, _, s_v in evals}
    print("reference fp32 evaluations, vector error vs float64:", json.dumps(per), flush=True)
    vec = {k: max(per[tag][k] for tag in per) for k in mods}
    nrm = {k: max(mg.rel(g_v[k], g64[k]) for _, g_v, _ in evals) for k in mods}
    print("reference fp32 (worst of its five evaluations) vs float64: vector", vec, "norm", nrm, flush=True)
    out = dict(H=H, W=W, B=B, weight_seed=7, config_id=1, sketch_k=K, loss_ref_fp32=l_ref, loss_f64=l64, gradnorm_ref_fp32=g_ref,
               gradnorm_f64=g64, ref_fp32_vector_err=vec, ref_fp32_norm_err=nrm, ref_fp32_vector_err_by_evaluation=per,
               oracle_vs_reference=chk, sketch_f64=s64, sketch_ref_fp32=s_ref)
    json.dump(out, open(os.path.join(HERE, "b8_reference.json"), "w"))
    print("b8_reference.json written")


if __name__ == "__main__":
    main()
