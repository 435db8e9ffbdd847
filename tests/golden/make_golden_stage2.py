#!/usr/bin/env python3
"""Pins the stage-2.1 / 2.2 branches of the oracle (and through it the HIP path) to the REFERENCE:
`get_crf_loss` / `get_pl_loss` and the EMA-teacher -> object channel -> image size -> CRFHead -> mask size
orchestration (models/rcf_model.py:380-408,490-529; configs/rcf/rcf_stage2.1.yaml, rcf_stage2.2.yaml).

Same procedure as make_golden_variants.py: the reference model (imported from /root/reference with the stand-in
modules of make_golden.py) and the oracle restatement run one training step on the same seeded weights and batch.
The reference's only native dependency on this path, `torchcrf_cpp.crf_soft` (CUDA, cannot be built here), is bound
on BOTH sides to the C restatement oracle/crf_ref.c -- so everything around the FFI call is the
reference's own code, and what is stored are the reference's numbers:
  losses, per-module gradient norms, the u8 image / unary energies it handed to the FFI, the MAPs it got back and
  the CRF / pseudo-label targets at mask size it fed to its loss.

Run in the build container only:  python tests/golden/make_golden_stage2.py
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
    import rcf_torch as orc
    import rcf_amd                                         # noqa
    from rcf_amd import config, synth
    torch.manual_seed(0)
    torch.set_num_threads(8)
    H, W, B = 64, 96, 2
    out, arrays = {}, {}
    for name in config.STAGE2_VARIANTS:
        kw, oc = config.variant_model_kwargs(name, H, W)
        args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_golden", object_channel=oc, eval_save=False,
                                     eval_export=False)
        calls = []

        def crf_soft(img, UU, W_, H_, *rest):             # the reference's FFI call site, models/crf_head.py:57-58
            m = crf_oracle.crf_soft_torch(img, UU, W_, H_, *rest)
            calls.append((img.clone(), UU.clone(), m.clone(), [float(v) for v in rest]))
            return m
        sys.modules["torchcrf_cpp"].crf_soft = crf_soft
        ref = ref_models.RCFModel(args, **copy.deepcopy(kw))
        okw = copy.deepcopy(kw)
        if "crf_head" in okw:
            okw["crf_head"]["crf_soft"] = crf_oracle.crf_soft_torch
        ora = orc.RCFModel(args, **copy.deepcopy(okw))
        shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        assert list(shapes) == list(ora.state_dict().keys()), "state-dict schema differs"
        sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()}
        ref.load_state_dict(sd)
        ora.load_state_dict(sd)
        nb = synth.make_batch(B, H, W, config_id=1)
        pl = synth.make_pl_masks(B, H, W, config_id=1)
        targets = {}
        for meth in ("get_crf_loss", "get_pl_loss"):       # what the reference feeds its own loss
            orig = getattr(ref, meth)

            def wrapped(p, t, _orig=orig, _meth=meth):
                targets[_meth] = t.detach().clone()
                return _orig(p, t)
            setattr(ref, meth, wrapped)

        extras = {}

        def step(m, dbl=False, tag="x"):
            m.train()
            b = mg.torch_batch(nb)
            b["pl_masks"] = [torch.from_numpy(a) for a in pl]
            if dbl:
                b = {k: ([t.double() for t in v] if k in ("imgs", "gt_fw_flows", "gt_bw_flows", "pl_masks") else v)
                     for k, v in b.items()}
            l = m(b)
            l["loss"].backward()
            extras[tag] = {k: v.detach() for k, v in l.items() if "loss" not in k}
            return {k: float(v) for k, v in l.items() if "loss" in k}, mg.grad_norms(m)
        l_ref, g_ref = step(ref, tag="ref")
        ref_calls = list(calls)
        ref_targets = dict(targets)
        l_ora, g_ora = step(ora, tag="ora")
        o64 = orc.RCFModel(args, **copy.deepcopy(okw))
        o64.load_state_dict(sd)
        if o64.crf_head is not None:                       # CRFHead constants are plain tensors, not buffers
            o64.crf_head.mean, o64.crf_head.std = o64.crf_head.mean.double(), o64.crf_head.std.double()
        l64, g64 = step(o64.double(), dbl=True)
        # the yardstick of make_golden_variants.py: the reference's own fp32 deviation from the float64 truth under other,
        # equally valid reduction orders (1 thread; channels_last convolutions)
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
        # the EMA copies after the momentum update (incl. the int64 num_batches_tracked truncation)
        rs, os_ = ref.state_dict(), ora.state_dict()
        chk["ema"] = max(float((rs[k].double() - os_[k].double()).abs().max()) for k in rs if "_ema." in k)
        assert sorted(l_ora) == sorted(l_ref), (sorted(l_ora), sorted(l_ref))
        print(name, json.dumps(chk))
        assert max(v for k, v in chk.items() if "gradnorm" not in k) < 1e-5, name
        assert max(v for k, v in chk.items() if "gradnorm" in k) < 2e-3, name
        out[name] = dict(H=H, W=W, B=B, weight_seed=7, config_id=1, object_channel=oc,
                         loss=l_ref, gradnorm=g_ref, truth_loss=l64, truth_gradnorm=g64,
                         ref32_err_loss={k: max(e[k] for e in e_l) for k in l_ref},
                         ref32_err_gradnorm={k: max(e[k] for e in e_g) for k in g_ref},
                         oracle_vs_reference=chk, crf_calls=len(ref_calls))
        if ref_calls:
            arrays[name + "_crf_img_u8"] = np.stack([c[0].numpy() for c in ref_calls])
            arrays[name + "_crf_unary"] = np.stack([c[1].numpy() for c in ref_calls])
            arrays[name + "_crf_map"] = np.stack([c[2].numpy().astype(np.uint8) for c in ref_calls])
            out[name]["crf_params"] = ref_calls[0][3]
            assert torch.equal(ref_targets["get_crf_loss"], extras["ora"]["_crf_masks"]), "oracle's CRF targets differ"
            arrays[name + "_crf_target"] = ref_targets["get_crf_loss"].numpy()
        if "get_pl_loss" in ref_targets:
            arrays[name + "_pl_target"] = ref_targets["get_pl_loss"].numpy()
        ema_keys = ["backbone2_ema.layer1.0.conv1.weight", "backbone2_ema.bn1.running_mean",
                    "backbone2_ema.bn1.num_batches_tracked", "decode_head2_ema.conv_seg.bias"]
        for k in ema_keys:
            arrays[name + "_ema_" + k.replace(".", "_")] = rs[k].numpy()
        out[name]["ema_keys"] = ema_keys
    json.dump(out, open(os.path.join(HERE, "stage2.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "stage2.npz"), **arrays)
    print("stage2.json / stage2.npz written")


if __name__ == "__main__":
    main()
