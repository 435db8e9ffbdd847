#!/usr/bin/env python3
"""Pins the oracle (and through it the HIP path) on the configuration variants of the training step that
the main fixtures do not exercise: FBMS / STv2 configs, both sharpen-loss branches, the joint residual
head and the object-channel compactness loss.  Same procedure as make_golden.py: the REFERENCE model
(imported from /root/reference with the stand-in modules) and the oracle restatement run one step on the
same seeded weights and batch; the reference's numbers are stored in variants.json together with the
float64 ground truth (oracle in double) and the reference's own fp32 deviation from it.

Run in the build container only:  python tests/golden/make_golden_variants.py
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
    import rcf_torch as orc
    import rcf_amd                                         # noqa
    from rcf_amd import config, synth
    torch.manual_seed(0)
    torch.set_num_threads(8)
    H, W, B = 64, 96, 2
    out = {}
    for name in config.VARIANTS:
        kw, oc = config.variant_model_kwargs(name, H, W)
        args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_golden", object_channel=oc, eval_save=False,
                                     eval_export=False)
        ref = ref_models.RCFModel(args, **copy.deepcopy(kw))
        ora = orc.RCFModel(args, **copy.deepcopy(kw))
        shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        assert list(shapes) == list(ora.state_dict().keys()), "state-dict schema differs"
        sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()}
        ref.load_state_dict(sd)
        ora.load_state_dict(sd)
        nb = synth.make_batch(B, H, W, config_id=1)

        def step(m, dbl=False):
            m.train()
            b = mg.torch_batch(nb)
            if dbl:
                b = {k: ([t.double() for t in v] if k in ("imgs", "gt_fw_flows", "gt_bw_flows") else v) for k, v in b.items()}
            l = m(b)
            l["loss"].backward()
            return {k: float(v) for k, v in l.items() if "loss" in k}, mg.grad_norms(m)
        l_ref, g_ref = step(ref)
        l_ora, g_ora = step(ora)
        o64 = orc.RCFModel(args, **copy.deepcopy(kw))
        o64.load_state_dict(sd)
        l64, g64 = step(o64.double(), dbl=True)
        # reference fp32 under other (equally valid) reduction orders: 1 thread, channels_last
        e_l = [{k: mg.rel(l_ref[k], l64[k]) for k in l_ref}]
        e_g = [{k: mg.rel(g_ref[k], g64[k]) for k in g_ref}]
        for nthreads, cl in ((1, False), (8, True)):
            torch.set_num_threads(nthreads)
            rv = ref_models.RCFModel(args, **copy.deepcopy(kw))
            rv.load_state_dict(sd)
            if cl:
                rv = rv.to(memory_format=torch.channels_last)
            lv, gv = step(rv)
            e_l.append({k: mg.rel(lv[k], l64[k]) for k in l_ref})
            e_g.append({k: mg.rel(gv[k], g64[k]) for k in g_ref})
            torch.set_num_threads(8)
        chk = {k: mg.rel(l_ora[k], l_ref[k]) for k in l_ref}
        chk.update({"gradnorm." + k: mg.rel(g_ora[k], g_ref[k]) for k in g_ref})
        assert sorted(l_ora) == sorted(l_ref), (sorted(l_ora), sorted(l_ref))
        print(name, json.dumps(chk))
        assert max(v for k, v in chk.items() if "loss" in k and "gradnorm" not in k) < 1e-5, name
        out[name] = dict(H=H, W=W, B=B, weight_seed=7, config_id=1, object_channel=oc,
                         loss=l_ref, gradnorm=g_ref, truth_loss=l64, truth_gradnorm=g64,
                         ref32_err_loss={k: max(e[k] for e in e_l) for k in l_ref},
                         ref32_err_gradnorm={k: max(e[k] for e in e_g) for k in g_ref},
                         oracle_vs_reference=chk)
    json.dump(out, open(os.path.join(HERE, "variants.json"), "w"), indent=1)
    print("variants.json written")


if __name__ == "__main__":
    main()
