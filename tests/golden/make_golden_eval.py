#!/usr/bin/env python3
"""Fixtures for the evaluation metric: the REFERENCE's own code decides the numbers.

main.py cannot be imported here (pytorch_lightning, wandb, the dataset package), so -- as make_golden.py does for
`get_lr` -- the methods `on_test_start`, `test_step` and `test_epoch_end` of its `Model` class (main.py:193-292) are
cut out of the source with `ast` at generation time and executed on a stand-in `self`; `utils.eval_utils._resize` /
`utils.iou` are the reference's own (imported from /root/reference; mmseg.ops.resize is the F.interpolate stand-in of
make_golden.py).  Nothing of the reference's text is stored: only the synthetic inputs' seeds and the outputs.

Inputs: seeded smooth soft masks [N, C, h, w] (softmax of low-frequency logits) and annotations [N, H, W] in
{0, 128, 255} for three "sequences"; the generator picks the first seed for which no resized mask value lies within 3e-7
(10 ulp) of the threshold and no arg-max margin is below 3e-7, so that last-bit differences between two correct bilinear
resizes cannot flip a pixel and the integer counts are comparable bit for bit.
Cases: eval_pos_th 0.35 and -1 (hard arg-max), object channel unknown (vote) and given.

Run in the build container only:  python tests/golden/make_golden_eval.py
"""
import ast
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


def main():
    mg.install_standins()
    sys.path.insert(0, mg.REF)
    import utils as ref_utils                              # noqa: the reference's own utils (eval_utils, iou)
    sys.path.insert(0, mg.ROOT)
    import rcf_amd                                         # noqa
    from rcf_amd.synth import eval_inputs as synth_eval_inputs
    tree = ast.parse(open(os.path.join(mg.REF, "main.py")).read())
    want = {"on_test_start", "test_step", "test_epoch_end"}
    fns = [f for c in tree.body if isinstance(c, ast.ClassDef) and c.name == "Model"
           for f in c.body if isinstance(f, ast.FunctionDef) and f.name in want]
    assert {f.name for f in fns} == want
    for f in fns:
        f.decorator_list = []                              # @rank_zero_only: single process here
    logs = {}
    ns = {"utils": ref_utils, "np": np, "torch": torch, "F": F, "dist": None,
          "logger": types.SimpleNamespace(info=lambda *a, **k: None)}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "main.py", "exec"), ns)

    # the margin that makes the fixture insensitive to the last bit of the resize
    for seed in range(31, 200):
        masks, ann, names = synth_eval_inputs(seed=seed)
        rs = F.interpolate(torch.from_numpy(masks), size=ann.shape[1:3], mode="bilinear", align_corners=True)
        top2 = torch.topk(rs, 2, dim=1).values
        if float((rs - 0.35).abs().min()) > 3e-7 and float((top2[:, 0] - top2[:, 1]).min()) > 3e-7:
            break
    else:
        raise SystemExit("no seed with a safe margin")
    print("seed", seed, "threshold margin", float((rs - 0.35).abs().min()), "argmax margin", float((top2[:, 0] - top2[:, 1]).min()))
    out = {}
    for case, (pos_th, oc_given) in {"th035_vote": (0.35, None), "argmax_vote": (-1, None), "th035_oc2": (0.35, 2)}.items():
        C = masks.shape[1]
        args = types.SimpleNamespace(eval_pos_th=pos_th, model_kwargs={"mask_layer": C}, rank=-1, object_channel=oc_given,
                                     set_object_channel_after_epoch=1)
        pos = {"i": 0}

        def forward(x, pos=pos):
            n = len(x["ann"])
            m = torch.from_numpy(masks[pos["i"]:pos["i"] + n])
            pos["i"] += n
            return m
        me = types.SimpleNamespace(args=args, object_channel=oc_given, forward=forward,
                                   log=lambda k, v, **kw: logs.__setitem__(k, float(v)), current_epoch=0,
                                   trainer=types.SimpleNamespace(sanity_checking=False, testing=True))
        ns["on_test_start"](me)
        for i in range(0, len(ann), 2):                    # batches of 2 frames
            batch = {"ann": torch.from_numpy(ann[i:i + 2]), "seq_names": names[i:i + 2]}
            ns["test_step"](me, batch, 0)
        per_frame = {k: [float(v) for v in seq] for k, seq in me.iou_all_sequences.items()}
        logs.clear()
        ns["test_epoch_end"](me, None, name="test_miou", display_all=False)
        out[case] = dict(eval_pos_th=pos_th, object_channel_given=oc_given, seed=seed, per_frame_iou=per_frame,
                         max_channel_freq=[int(v) for v in me.max_channel_freq], object_channel_after=int(me.object_channel),
                         test_miou=logs["test_miou"], test_miou_frame_avg=logs["test_miou_frame_avg"],
                         per_sequence={k[len("test_miou_"):]: v for k, v in logs.items()
                                       if k.startswith("test_miou_") and k != "test_miou_frame_avg"})
        print(case, json.dumps(out[case])[:400])
    json.dump(out, open(os.path.join(HERE, "eval.json"), "w"), indent=1)
    print("eval.json written")


if __name__ == "__main__":
    main()
