#!/usr/bin/env python3
"""Fixtures for the configuration `bench.py` runs: Dropout2d 0.1 in both FCN heads (configs/rcf/rcf_stage1.yaml:118,139),
with the random draw taken out.

The REFERENCE model (imported from /root/reference with the stand-in modules of make_golden.py; its SyncBN is the
stand-in's BatchNorm2d: one process, same arithmetic) is built with dropout_ratio 0.1 and the `dropout` module of
decode_head2 / decode_head3 (models/decode_head.py:84-87, applied at models/fcn_head.py:144-145) is replaced by
oracle.FixedDropout2d holding a draw from synth.dropout_scale (seeds below) -- decode_head2 sees 2B frames, decode_head3 B
concatenated pairs.  The same draw goes to the oracle (checked against the reference right here) and, in the GPU tests,
to rcf_amd's FCNHead.keep_mask.  Stored per case: the reference's fp32 losses / gradient norms / logits (argmax + margin
at 480x854), the float64 truth, the reference's own fp32 spread around it (the yardstick of tests/test_model_gpu.py), and
the reference under torch.autocast(bf16) with the same draw (the yardstick of tests/test_bf16_gpu.py).

Run in the build container only:  python tests/golden/make_golden_dropout.py [--skip-large]
"""
import argparse
import copy
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402

P_DROP = 0.1
SEED_HEAD2, SEED_HEAD3 = 21, 22


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-large", action="store_true")
    opts = ap.parse_args()
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
    args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_golden", object_channel=None, eval_save=False, eval_export=False)
    meta, arrays = {}, {}
    cases = [("small", 96, 160, 2)] + ([] if opts.skip_large else [("480x854", 480, 854, 1)])
    for tag, H, W, B in cases:
        kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=P_DROP, norm="SyncBN")
        kw.update(log_interval=10 ** 9, train_iter=1)
        probe = ref_models.RCFModel(args, **copy.deepcopy(kw))
        assert isinstance(probe.decode_head2.dropout, torch.nn.Dropout2d) and isinstance(probe.decode_head3.dropout, torch.nn.Dropout2d)
        shapes = {k: tuple(v.shape) for k, v in probe.state_dict().items()}
        sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()}
        nb = synth.make_batch(B, H, W, config_id=1)
        s2 = synth.dropout_scale(2 * B, kw["decode_head2"]["channels"], P_DROP, SEED_HEAD2)
        s3 = synth.dropout_scale(B, kw["decode_head3"]["channels"], P_DROP, SEED_HEAD3)
        assert (s2 == 0).any() and (s3 == 0).any()

        def run(cls, mode, nthreads=8, cl=False, ulp_seed=0):
            """mode: "fp32" | "bf16" (torch.autocast) | "f64" """
            torch.set_num_threads(nthreads)
            m = cls(args, **copy.deepcopy(kw))
            sdl = sd
            if ulp_seed:                                   # parameters moved by one unit in the last place (make_golden.py)
                gp = torch.Generator().manual_seed(1000 + ulp_seed)
                sdl = {k: (v * (1 + (torch.randint(0, 2, v.shape, generator=gp).float() * 2 - 1) * 2.0 ** -23)
                           if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")) else v)
                       for k, v in sd.items()}
            m.load_state_dict(sdl)
            m.decode_head2.dropout = orc.FixedDropout2d(s2)
            m.decode_head3.dropout = orc.FixedDropout2d(s3)
            b = mg.torch_batch(nb)
            if mode == "f64":
                m = m.double()
                b = {k: ([t.double() for t in v] if k in ("imgs", "gt_fw_flows", "gt_bw_flows") else v) for k, v in b.items()}
            if cl:
                m = m.to(memory_format=torch.channels_last)
            m.train()
            cap = {}
            h = m.decode_head2.conv_seg.register_forward_hook(lambda mod, i, o: cap.__setitem__("logits", o.detach().double()))
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=mode == "bf16"):
                l = m(b)
            l["loss"].backward()
            h.remove()
            torch.set_num_threads(8)
            grads = {n: m.get_parameter(n).grad.detach().double().numpy().ravel()[:256].copy() for n in mg.SAMPLED}
            return {k: float(v) for k, v in l.items() if "loss" in k}, mg.grad_norms(m), cap["logits"], grads

        l32, g32, z32, s32 = run(ref_models.RCFModel, "fp32")
        lo, go, zo, so = run(orc.RCFModel, "fp32")
        chk = {k: mg.rel(lo[k], l32[k]) for k in l32}
        chk.update({"gradnorm." + k: mg.rel(go[k], g32[k]) for k in g32})
        chk.update({"grad." + k: mg.rel(so[k], s32[k]) for k in mg.SAMPLED})
        chk["logits"] = mg.rel(zo.numpy(), z32.numpy())
        print(tag, "oracle vs reference, same Dropout2d draw:", json.dumps(chk))
        assert max(chk.values()) < 2e-4, "the oracle disagrees with the reference under the injected draw"
        # without the draw the result is a different one: the fixture exercises what it claims to
        l_nodrop = json.load(open(os.path.join(HERE, "bf16.json")))[tag]["loss_fp32"]["loss"]
        assert abs(l_nodrop - l32["loss"]) / abs(l_nodrop) > 1e-4, "the injected draw did not change the loss"
        l64, g64, z64, s64 = run(orc.RCFModel, "f64")
        ref32 = dict(loss=[mg.rel(l32["loss"], l64["loss"])], logits=[mg.rel(z32.numpy(), z64.numpy())],
                     gradnorm=[[mg.rel(g32[k], g64[k]) for k in sorted(g32)]], grad=[[mg.rel(s32[k], s64[k]) for k in mg.SAMPLED]])
        for nthreads, cl, ulp in ((1, False, 0), (8, True, 0), (8, False, 1), (8, False, 2)):
            lv, gv, zv, sv = run(ref_models.RCFModel, "fp32", nthreads, cl, ulp)
            ref32["loss"].append(mg.rel(lv["loss"], l64["loss"]))
            ref32["logits"].append(mg.rel(zv.numpy(), z64.numpy()))
            ref32["gradnorm"].append([mg.rel(gv[k], g64[k]) for k in sorted(g32)])
            ref32["grad"].append([mg.rel(sv[k], s64[k]) for k in mg.SAMPLED])
        print(tag, "reference fp32 variants vs float64:", json.dumps(ref32))
        l16, g16, z16, _ = run(ref_models.RCFModel, "bf16")
        lo16, go16, zo16, _ = run(orc.RCFModel, "bf16")
        chk16 = {k: mg.rel(lo16[k], l16[k]) for k in l16}
        chk16.update({"gradnorm." + k: mg.rel(go16[k], g16[k]) for k in g16})
        chk16["logits"] = mg.rel(zo16.numpy(), z16.numpy())
        print(tag, "oracle(autocast) vs reference(autocast):", json.dumps(chk16))
        assert chk16["logits"] < 1e-6 and max(v for k, v in chk16.items() if k.startswith("loss")) < 1e-3
        top2 = torch.topk(z32, 2, dim=1).values
        margin = (top2[:, 0] - top2[:, 1]) / z32.abs().max()                # relative to the logits' range, as in bf16.npz
        mism = z32.argmax(1) != z16.argmax(1)
        rmargin = margin
        dev = {"loss": {k: mg.rel(l16[k], l32[k]) for k in l32}, "gradnorm": {k: mg.rel(g16[k], g32[k]) for k in g32},
               "logits": mg.rel(z16.numpy(), z32.numpy()), "argmax_mismatch_frac": float(mism.float().mean()),
               "argmax_sure_margin": float(rmargin[mism].max()) if bool(mism.any()) else 0.0}
        print(tag, "reference autocast-bf16 vs reference fp32:", json.dumps(dev))
        keys = sorted(g32)
        meta[tag] = dict(H=H, W=W, B=B, C=kw["mask_layer"], weight_seed=7, config_id=1, p=P_DROP, seed_head2=SEED_HEAD2,
                         seed_head3=SEED_HEAD3, dropped_head2=int((s2 == 0).sum()), dropped_head3=int((s3 == 0).sum()),
                         loss_fp32=l32, loss_bf16=l16, loss_f64=l64, gradnorm_fp32=g32, gradnorm_bf16=g16, gradnorm_f64=g64,
                         logit_absmax=float(z32.abs().max()), ref32_err_loss=max(ref32["loss"]),
                         ref32_err_logits=max(ref32["logits"]),
                         ref32_err_gradnorm=dict(zip(keys, np.array(ref32["gradnorm"]).max(axis=0).tolist())),
                         ref32_err_grad=dict(zip(mg.SAMPLED, np.array(ref32["grad"]).max(axis=0).tolist())),
                         ref_bf16_vs_fp32=dev, oracle_vs_reference=chk, oracle_vs_reference_autocast=chk16)
        arrays[tag + "_argmax_fp32"] = z32.argmax(1).numpy().astype(np.uint8)
        arrays[tag + "_argmax_bf16"] = z16.argmax(1).numpy().astype(np.uint8)
        arrays[tag + "_margin_fp32"] = margin.numpy().astype(np.float16)
        for i, n in enumerate(mg.SAMPLED):
            arrays[f"{tag}_truth_grad_{i}"] = s64[n]
        if tag == "small":
            arrays[tag + "_logits_fp32"] = z32.numpy().astype(np.float32)
            arrays[tag + "_logits_f64"] = z64.numpy()
            arrays[tag + "_masks_fp32"] = F.softmax(z32, dim=1).numpy().astype(np.float32)
            arrays[tag + "_masks_bf16"] = F.softmax(z16, dim=1).numpy().astype(np.float32)
        else:
            arrays[tag + "_logits_f64_0"] = z64[:1].numpy()
            arrays[tag + "_mask_mean_fp32"] = F.softmax(z32, dim=1).mean(dim=(2, 3)).numpy()
    out = os.path.join(HERE, "dropout.json")
    if opts.skip_large and os.path.exists(out):               # keep the large case of an earlier full run
        old_meta = json.load(open(out))
        old_arr = dict(np.load(os.path.join(HERE, "dropout.npz")))
        for k, v in old_meta.items():
            meta.setdefault(k, v)
        for k, v in old_arr.items():
            arrays.setdefault(k, v)
    json.dump(meta, open(out, "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "dropout.npz"), **arrays)
    print("dropout.json / dropout.npz written")


if __name__ == "__main__":
    main()
