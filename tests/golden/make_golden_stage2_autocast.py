#!/usr/bin/env python3
"""The yardstick for the stage-2.1 step in mixed precision (bench.py `stage2_bf16_ms_per_step`; BASELINE configs[3] is the step
that trains with the CRF in the loop, models/rcf_model.py:490-529): the REFERENCE's own stage-2.1 step (make_golden_stage2.py's
set-up: torchcrf_cpp.crf_soft bound to oracle/crf_ref.c) under torch.autocast(bf16) against its fp32 run on the same weights
and batch -- how far 16-bit storage moves the losses, the module gradient norms and the CRF targets of the reference itself.
tests/test_stage2_gpu.py::test_stage21_bf16_step_vs_reference_autocast holds the HIP bf16 step to a multiple of these.

Run in the build container only:  python tests/golden/make_golden_stage2_autocast.py
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


def main():
    mg.install_standins()
    sys.path.insert(0, mg.REF)
    import models as ref_models                            # noqa: the reference itself
    sys.path.insert(0, mg.ROOT)
    sys.path.insert(0, os.path.join(mg.ROOT, "oracle"))
    import crf_oracle
    import rcf_amd                                         # noqa
    from rcf_amd import config, synth
    torch.manual_seed(0)
    torch.set_num_threads(8)
    H, W, B = 64, 96, 2
    name = "stage21"
    kw, oc = config.variant_model_kwargs(name, H, W)
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_golden", object_channel=oc, eval_save=False, eval_export=False)
    sys.modules["torchcrf_cpp"].crf_soft = crf_oracle.crf_soft_torch
    probe = ref_models.RCFModel(args, **copy.deepcopy(kw))
    shapes = {k: tuple(v.shape) for k, v in probe.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()}
    nb = synth.make_batch(B, H, W, config_id=1)

    def run(amp):
        m = ref_models.RCFModel(args, **copy.deepcopy(kw))
        m.load_state_dict(sd)
        m.train()
        tgt = {}
        orig = m.get_crf_loss

        def wrapped(p, t):
            tgt["crf"] = t.detach().float().clone()
            return orig(p, t)
        m.get_crf_loss = wrapped
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
            l = m(mg.torch_batch(nb))
        l["loss"].backward()
        return {k: float(v) for k, v in l.items() if "loss" in k}, mg.grad_norms(m), tgt["crf"]
    l32, g32, t32 = run(False)
    l16, g16, t16 = run(True)
    fx32 = json.load(open(os.path.join(HERE, "stage2.json")))[name]
    assert max(mg.rel(l32[k], fx32["loss"][k]) for k in l32) < 1e-6, "the fp32 run is not the one in stage2.json"
    dev = {"loss": {k: mg.rel(l16[k], l32[k]) for k in l32}, "gradnorm": {k: mg.rel(g16[k], g32[k]) for k in g32},
           "crf_target_differing_frac": float(((t16 - t32).abs() > 1e-5).float().mean())}
    print(name, "reference autocast-bf16 vs reference fp32:", json.dumps(dev))
    out = {name: dict(H=H, W=W, B=B, weight_seed=7, config_id=1, object_channel=oc, loss_fp32=l32, loss_bf16=l16,
                      gradnorm_fp32=g32, gradnorm_bf16=g16, ref_bf16_vs_fp32=dev)}
    json.dump(out, open(os.path.join(HERE, "stage2_autocast.json"), "w"), indent=1)
    print("stage2_autocast.json written")


if __name__ == "__main__":
    main()
