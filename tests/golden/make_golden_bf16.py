#!/usr/bin/env python3
"""Fixtures for the mixed-precision step (BASELINE configs[2]: bf16 forward / fp32 gradients).

The reference trains STv2 / FBMS with Lightning `precision: 16` (configs/rcf_stv2/rcf_stage1.yaml:57-60,
configs/rcf_fbms59/rcf_stage1.yaml:61), i.e. inside torch autocast.  Here the REFERENCE model itself (imported from
/root/reference with the stand-in modules of make_golden.py) runs one stage-1 training step twice on the same seeded
weights and batch: in fp32, and under torch.autocast("cpu", dtype=torch.bfloat16) -- bf16 being the 16-bit type the
MI355X path uses (fp32's exponent range: no loss scaling).  Stored: both loss dicts, both sets of gradient norms, the
softmax masks of both runs (small case) or the argmax map + top-2 logit margin (480x854), and how far the reference's own
autocast run is from its fp32 run -- the yardstick the HIP bf16 step is held to (tests/test_bf16_gpu.py).
The oracle restatement is run under the same autocast and must reproduce the reference's autocast numbers.

Run in the build container only:  python tests/golden/make_golden_bf16.py [--skip-large]
"""
import argparse
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
    meta, arrays = {}, {}
    cases = [("small", 96, 160, 2)] + ([] if opts.skip_large else [("480x854", 480, 854, 1)])
    for tag, H, W, B in cases:
        kw = config.stage1_model_kwargs(config.mask_size_for(H, W), dropout=0.0, norm="BN")
        kw.update(log_interval=10 ** 9, train_iter=1)
        args = types.SimpleNamespace(checkpoints_dir="/tmp/rcf_golden", object_channel=None, eval_save=False,
                                     eval_export=False)
        probe = ref_models.RCFModel(args, **copy.deepcopy(kw))
        shapes = {k: tuple(v.shape) for k, v in probe.state_dict().items()}
        sd = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, seed=7).items()}
        nb = synth.make_batch(B, H, W, config_id=1)

        def run(cls, amp):
            m = cls(args, **copy.deepcopy(kw))
            m.load_state_dict(sd)
            m.train()
            cap = {}
            # the logits of decode_head2 (the reference keeps no handle on them): forward hook on its last conv
            h = m.decode_head2.conv_seg.register_forward_hook(lambda mod, i, o: cap.__setitem__("logits", o.detach().float()))
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
                l = m(mg.torch_batch(nb))
            l["loss"].backward()
            h.remove()
            return {k: float(v) for k, v in l.items() if "loss" in k}, mg.grad_norms(m), cap["logits"]
        l32, g32, z32 = run(ref_models.RCFModel, False)
        l16, g16, z16 = run(ref_models.RCFModel, True)
        lo16, go16, zo16 = run(orc.RCFModel, True)
        chk = {k: mg.rel(lo16[k], l16[k]) for k in l16}
        chk.update({"gradnorm." + k: mg.rel(go16[k], g16[k]) for k in g16})
        chk["logits"] = mg.rel(zo16.numpy(), z16.numpy())
        print(tag, "oracle(autocast) vs reference(autocast)", json.dumps(chk))
        # conv / BN / resize modules are the same torch modules on both sides (identical logits); the flow head is RESTATED
        # (einsum instead of broadcast-multiply-sum), so autocast routes a few of its ops to different precisions: the
        # two agree far below bf16's own noise (1e-2), not to fp32 rounding
        assert chk["logits"] < 1e-6 and max(v for k, v in chk.items() if k.startswith("loss")) < 1e-3
        assert max(v for k, v in chk.items() if k.startswith("gradnorm")) < 1e-2
        C = kw["mask_layer"]
        p32 = torch.softmax(z32, dim=1)
        p16 = torch.softmax(z16, dim=1)
        top2 = torch.topk(z32, 2, dim=1).values
        margin = (top2[:, 0] - top2[:, 1]) / z32.abs().max()                 # relative to the logits' range
        mism = z32.argmax(1) != z16.argmax(1)
        # smallest relative margin above which the reference's own bf16 run decides every pixel like its fp32 run
        sure = float(margin[mism].max()) if bool(mism.any()) else 0.0
        dev = {"loss": {k: mg.rel(l16[k], l32[k]) for k in l32}, "gradnorm": {k: mg.rel(g16[k], g32[k]) for k in g32},
               "logits": mg.rel(z16.numpy(), z32.numpy()), "masks_max_abs": float((p16 - p32).abs().max()),
               "argmax_mismatch_frac": float(mism.float().mean()), "argmax_sure_margin": sure}
        print(tag, "reference autocast-bf16 vs reference fp32:", json.dumps(dev))
        meta[tag] = dict(H=H, W=W, B=B, C=C, weight_seed=7, config_id=1, loss_fp32=l32, loss_bf16=l16, gradnorm_fp32=g32,
                         gradnorm_bf16=g16, ref_bf16_vs_fp32=dev, oracle_vs_reference_autocast=chk)
        arrays[tag + "_argmax_fp32"] = z32.argmax(1).numpy().astype(np.uint8)
        # the reference's own autocast decisions, pixel by pixel: the flip RATE as a function of the fp32 margin is the
        # yardstick that covers every pixel (the `sure` threshold above is a max statistic: one far-out flip of the
        # reference leaves 8-23 % of the pixels to check).  Scaling conv_seg x10 as SURVEY section 7 proposed does NOT
        # separate the logits: signal and bf16 noise scale together (measured: identical flip maps, 56 % gradient-norm
        # deviation from the saturated softmax) -- tried and dropped.
        arrays[tag + "_argmax_bf16"] = z16.argmax(1).numpy().astype(np.uint8)
        arrays[tag + "_margin_fp32"] = margin.numpy().astype(np.float16)
        if tag == "small":
            arrays[tag + "_masks_fp32"] = p32.numpy()
            arrays[tag + "_masks_bf16"] = p16.numpy()
    json.dump(meta, open(os.path.join(HERE, "bf16.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "bf16.npz"), **arrays)
    print("bf16.json / bf16.npz written")


if __name__ == "__main__":
    main()
