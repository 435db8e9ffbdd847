#!/usr/bin/env python3
"""Fixtures for the data transform: the REFERENCE's own `Transform`
(dataset/transforms.py:884-924) decides the numbers.

dataset/transforms.py is imported from /root/reference as it is.  Its third-party operators are absent here (mmcv-full
1.6.2 = cv2, torchvision): `mmcv.imrescale` / `imflip` / `rgb2hsv` / `hsv2rgb` are served by oracle/transforms_np.py's
restatement of OpenCV's published 8-bit algorithms (PARITY-UNPINNED: no cv2 here, no vectors in the reference), and
torchvision's `Compose` / `functional.normalize` by the three-line stand-ins below.  What the fixtures therefore pin is
everything the reference itself does: which random calls it makes and in which order, the scale rule, crop, flip, the
photometric chain with its u8 round trips and its mode switch, FlowTransform (incl. scale_flow), PLTransform,
NumpyToTensor, TorchNormalize and the seg_fields handling of dataset/data.py:122-151.

Inputs are regenerated from seeds (rcf_amd.synth.loader_sample); stored per output tensor: its SHA-256 (data_pipeline.json) and a
16-strided subsample (data_pipeline.npz), plus the `scale` / `flip` the reference recorded.

Run in the build container only:  python tests/golden/make_golden_data.py
"""
import hashlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402

STRIDE = 16
SUBS = {}
CASES = {   # name: (training, transform kwargs, sample seeds (numpy seed = sample seed), frame size)
    "train_strong_pl": (True, dict(strong_aug=True, has_pl=True), [11, 12, 13, 14, 15, 16], (480, 854)),
    "train_weak_scaleflow": (True, dict(strong_aug=False, scale_flow=True), [21, 22], (480, 854)),
    "train_strong_small": (True, dict(strong_aug=True), [31, 32, 33, 34], (400, 500)),
    "eval": (False, dict(strong_aug=False), [41], (480, 854)),
}


def digest(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return hashlib.sha256(a.tobytes()).hexdigest()


def pack(a, key):
    a = np.asarray(a, dtype=np.float32)
    SUBS[key] = np.ascontiguousarray(a[..., ::STRIDE, ::STRIDE])
    return dict(shape=list(a.shape), sha256=digest(a), sub=key)


def load_reference_transforms():
    mg.install_standins()
    sys.path.insert(0, mg.ROOT)
    from oracle import transforms_np as T

    mmcv = sys.modules["mmcv"]
    mmcv.is_list_of = lambda seq, t: isinstance(seq, list) and all(isinstance(s, t) for s in seq)
    mmcv.imrescale, mmcv.imflip = T.imrescale, T.imflip
    img = types.ModuleType("mmcv.image")
    cs = types.ModuleType("mmcv.image.colorspace")
    cs.convert_color_factory = lambda src, dst: {("rgb", "hsv"): T.rgb2hsv, ("hsv", "rgb"): T.hsv2rgb}[(src, dst)]
    sys.modules["mmcv.image"], sys.modules["mmcv.image.colorspace"] = img, cs

    class Compose:                                         # torchvision.transforms.Compose
        def __init__(self, ts):
            self.transforms = ts

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    def normalize(tensor, mean, std, inplace=False):       # torchvision.transforms.functional.normalize
        if not inplace:
            tensor = tensor.clone()
        mean = torch.as_tensor(mean, dtype=tensor.dtype)
        std = torch.as_tensor(std, dtype=tensor.dtype)
        return tensor.sub_(mean.view(-1, 1, 1)).div_(std.view(-1, 1, 1))
    tv = sys.modules["torchvision"]
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvt.Compose, tvf.normalize, tvt.functional, tv.transforms = Compose, normalize, tvf, tvt
    sys.modules["torchvision.transforms"], sys.modules["torchvision.transforms.functional"] = tvt, tvf
    spec = importlib.util.spec_from_file_location("ref_transforms", os.path.join(mg.REF, "dataset", "transforms.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def main():
    ref = load_reference_transforms()
    import rcf_amd                                          # noqa
    from rcf_amd.synth import loader_sample
    out = {"stride": STRIDE, "cases": {}}
    for name, (training, kw, seeds, (H, W)) in CASES.items():
        tf = ref.Transform(training=training, **kw)
        recs = []
        for seed in seeds:
            s = loader_sample(seed, H, W)
            data = {"imgs": [f for f in s["frames"]], "seg_fields": []}     # dataset/data.py:93-151
            if training and kw.get("has_flow", True):
                data["gt_fw_flows"], data["gt_bw_flows"] = [s["fw"].copy()], [s["bw"].copy()]
                data["seg_fields"].extend(["gt_fw_flows", "gt_bw_flows"])
            if training and kw.get("has_pl", False):
                data["pl_masks"] = [m for m in s["pl"]]
                data["seg_fields"].append("pl_masks")
            if not training:                                # dataset/data.py:101-114; AnnotationTransform keeps channel 0
                data["ann"] = np.repeat(s["pl"][0][..., None], 3, axis=2)
            np.random.seed(seed)
            r = tf(data)
            if not training:
                assert np.array_equal(r["ann"], s["pl"][0])
            rec = dict(seed=seed, scale=[int(v) for v in r["scale"]], flip=bool(r.get("flip", False)),
                       imgs=[pack(t.numpy(), f"{name}.{seed}.img{i}") for i, t in enumerate(r["imgs"])])
            if "gt_fw_flows" in r:
                rec["fw"] = pack(r["gt_fw_flows"][0].numpy(), f"{name}.{seed}.fw")
                rec["bw"] = pack(r["gt_bw_flows"][0].numpy(), f"{name}.{seed}.bw")
            if "pl_masks" in r:
                rec["pl"] = [pack(t.numpy(), f"{name}.{seed}.pl{i}") for i, t in enumerate(r["pl_masks"])]
            recs.append(rec)
            print(name, seed, rec["scale"], rec["flip"], rec["imgs"][0]["shape"], rec["imgs"][0]["sha256"][:12])
        out["cases"][name] = dict(training=training, kwargs=kw, H=H, W=W, samples=recs)
    json.dump(out, open(os.path.join(HERE, "data_pipeline.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "data_pipeline.npz"), **SUBS)
    print("data_pipeline.json / .npz written", os.path.getsize(os.path.join(HERE, "data_pipeline.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
