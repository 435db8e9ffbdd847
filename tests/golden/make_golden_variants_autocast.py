#!/usr/bin/env python3
"""How far is the REFERENCE's own 16-bit autocast step from its fp32 step on the variants it trains with Lightning
`precision: 16` (configs/rcf_stv2/rcf_stage1.yaml:57-60, configs/rcf_fbms59/rcf_stage1.yaml:61)?  The reference model
(imported from /root/reference with the stand-in modules of make_golden.py) runs one training step on the seeded weights
and batch of variants.json three times: fp32, torch.autocast("cpu", bfloat16) -- the 16-bit type the MI355X path stores --
and torch.autocast("cpu", float16) with the loss scaled by 2^14 as GradScaler would (what `precision: 16` means on a GPU).
Stored per variant and 16-bit type: relative deviation of every loss term and of every module's gradient norm from the fp32
run -- the yardstick of tests/test_bf16_gpu.py::test_stv2_variant_under_autocast_precision.

Run in the build container only:  python tests/golden/make_golden_variants_autocast.py
"""
import copy
import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402


def main():
    mg.install_standins()
    sys.path.insert(0, mg.REF)
    import models as ref_models                            # noqa: the reference itself
    sys.path.insert(0, mg.ROOT)
    import rcf_amd                                         # noqa
    from rcf_amd import config, synth
    torch.manual_seed(0)
    torch.set_num_threads(8)
    H, W, B = 64, 96, 2
    out = {}
    for name in ("stv2", "fbms"):
        if name not in config.VARIANTS:
            continue
        kw, oc = config.variant_model_kwargs(name, H, W)
        args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_golden", object_channel=oc, eval_save=False, eval_export=False)
        probe = ref_models.RCFModel(args, **copy.deepcopy(kw))
        shapes = {k: tuple(v.shape) for k, v in probe.state_dict().items()}
        sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()}
        nb = synth.make_batch(B, H, W, config_id=1)

        def step(dtype, scale=1.0):
            m = ref_models.RCFModel(args, **copy.deepcopy(kw))
            m.load_state_dict(sd)
            m.train()
            with torch.autocast("cpu", dtype=dtype or torch.bfloat16, enabled=dtype is not None):
                l = m(mg.torch_batch(nb))
            (l["loss"] * scale).backward()
            g = {k: v / scale for k, v in mg.grad_norms(m).items()}
            return {k: float(v) for k, v in l.items() if "loss" in k}, g
        l32, g32 = step(None)
        rec = {}
        for tag, dt, sc in (("bf16", torch.bfloat16, 1.0), ("fp16", torch.float16, 2.0 ** 14)):
            try:
                l16, g16 = step(dt, sc)
            except Exception as e:                          # an op without a CPU kernel in this 16-bit type
                rec[tag] = {"error": repr(e)[:200]}
                continue
            rec[tag] = {"loss": {k: mg.rel(l16[k], l32[k]) for k in l32}, "gradnorm": {k: mg.rel(g16[k], g32[k]) for k in g32}}
        print(name, json.dumps(rec))
        out[name] = dict(H=H, W=W, B=B, weight_seed=7, config_id=1, loss_fp32=l32, gradnorm_fp32=g32, ref_autocast_vs_fp32=rec)
    json.dump(out, open(os.path.join(HERE, "variants_autocast.json"), "w"), indent=1)
    print("variants_autocast.json written")


if __name__ == "__main__":
    main()
