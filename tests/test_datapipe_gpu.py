"""Data transform, GPU half: the device gather kernels (csrc/datapipe.hip) behind
rcf_amd.data_pipeline.Transform against (a) the fixtures the reference's own `Transform` produced -- bit for bit, compared by
SHA-256 of every output tensor -- and (b) the oracle's pipeline on further seeds / shapes, also bit for bit (tolerance 0:
the whole path is u8 / integer arithmetic followed by two correctly rounded fp32 divisions)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import transforms_np as T
from rcf_amd import _lib
from rcf_amd.data_pipeline import PARAMS_DTYPE, Transform
from rcf_amd.synth import loader_sample
from test_datapipe_cpu import GOLD, check, sample_for

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def to_batch(samples, training, has_flow, has_pl):
    d = {"imgs": torch.from_numpy(np.stack([s["frames"] for s in samples])).to(DEV)}
    if training and has_flow:
        d["gt_fw_flows"] = torch.from_numpy(np.stack([s["fw"] for s in samples])).to(DEV)
        d["gt_bw_flows"] = torch.from_numpy(np.stack([s["bw"] for s in samples])).to(DEV)
    if training and has_pl:
        d["pl_masks"] = torch.from_numpy(np.stack([s["pl"] for s in samples])).to(DEV)
    return d


@pytest.mark.parametrize("case", sorted(GOLD["cases"]))
def test_device_transform_reproduces_the_reference_fixtures(case):
    c = GOLD["cases"][case]
    kw = c["kwargs"]
    tf = Transform(training=c["training"], **kw)
    params, samples = [], []
    for rec in c["samples"]:                                   # one batch, every sample with its own decisions
        np.random.seed(rec["seed"])
        params.append(tf.sample_params(c["H"], c["W"]))
        samples.append(loader_sample(rec["seed"], c["H"], c["W"]))
    out = tf(to_batch(samples, c["training"], kw.get("has_flow", True), kw.get("has_pl", False)), params=np.stack(params))
    assert len(out["imgs"]) == 2 and out["imgs"][0].shape[0] == len(samples)
    for b, rec in enumerate(c["samples"]):
        for i, r in enumerate(rec["imgs"]):
            check(out["imgs"][i][b].cpu().numpy(), r, f"{case} seed {rec['seed']} frame {i} ops {int(params[b]['ops']):05b}")
        if "fw" in rec:
            check(out["gt_fw_flows"][0][b].cpu().numpy(), rec["fw"], f"{case} seed {rec['seed']} fw")
            check(out["gt_bw_flows"][0][b].cpu().numpy(), rec["bw"], f"{case} seed {rec['seed']} bw")
        for i, r in enumerate(rec.get("pl", [])):
            check(out["pl_masks"][i][b].cpu().numpy(), r, f"{case} seed {rec['seed']} pl {i}")


@pytest.mark.parametrize("H,W,seeds", [(480, 854, range(100, 108)), (400, 401, range(200, 204)), (720, 1280, range(300, 302))])
def test_device_transform_equals_oracle_on_more_seeds(H, W, seeds):
    tf = Transform(training=True, strong_aug=True, has_pl=True, scale_flow=True)
    rng = np.random.RandomState(H + W)
    samples = [loader_sample(s, H, W) for s in seeds]
    params = np.stack([tf.sample_params(H, W, rng) for _ in seeds])
    out = tf(to_batch(samples, True, True, True), params=params)
    for b, s in enumerate(samples):
        ref = T.apply_params(s, params[b], 384, 384)
        for i in range(2):
            assert np.array_equal(out["imgs"][i][b].cpu().numpy(), ref["imgs"][i]), f"seed {seeds[b]} frame {i} ops {int(params[b]['ops'])}"
            assert np.array_equal(out["pl_masks"][i][b].cpu().numpy(), ref["pl"][i])
        assert np.array_equal(out["gt_fw_flows"][0][b].cpu().numpy(), ref["fw"])
        assert np.array_equal(out["gt_bw_flows"][0][b].cpu().numpy(), ref["bw"])


def test_every_photometric_stage_alone_and_extreme_parameters():
    # single stages with hand-set extreme parameters, on pure noise frames (every hue sector, s = 0, v = 0 pixels)
    H, W = 96, 130
    g = np.random.default_rng(5)
    frames = g.integers(0, 256, size=(1, 2, H, W, 3), dtype=np.uint8)
    frames[0, 0, :8] = 0
    frames[0, 0, 8:16] = 255
    frames[0, 0, 16:24] = frames[0, 0, 16:24, :, :1]           # grey rows: s = 0
    tf = Transform(training=False)
    settings = [dict(ops=1, beta=-32.0), dict(ops=1, beta=31.99), dict(ops=2, alpha_c=0.5), dict(ops=2 | 16, alpha_c=1.5),
                dict(ops=4, alpha_s=0.5), dict(ops=4, alpha_s=1.5), dict(ops=8, hue_delta=-17.999), dict(ops=8, hue_delta=17.3),
                dict(ops=8, hue_delta=0.0), dict(ops=31, beta=7.5, alpha_c=1.31, alpha_s=0.77, hue_delta=-3.25),
                dict(ops=15, beta=-20.0, alpha_c=0.61, alpha_s=1.49, hue_delta=11.0)]
    for flip in (0, 1):
        for st in settings:
            p = np.zeros((1,), dtype=PARAMS_DTYPE)
            p["rw"], p["rh"], p["flip"], p["flow_sx"], p["flow_sy"] = 117, 86, flip, 1, 1     # 0.9 x: a real resize
            for k, v in st.items():
                p[k] = v
            out = tf({"imgs": torch.from_numpy(frames).to(DEV)}, params=p)
            ref = T.apply_params(dict(frames=frames[0]), p[0], 86, 117)
            for i in range(2):
                got = out["imgs"][i][0].cpu().numpy()
                assert np.array_equal(got, ref["imgs"][i]), f"{st} flip {flip}: {int((got != ref['imgs'][i]).sum())} values differ"


def test_upscaling_resize_border_rules():
    # frames smaller than the target: the border taps of cv2's horizontal / vertical tables differ (see oracle)
    H, W = 50, 70
    g = np.random.default_rng(9)
    frames = g.integers(0, 256, size=(2, 1, H, W, 3), dtype=np.uint8)
    flows = g.normal(size=(2, H, W, 2)).astype(np.float32)
    tf = Transform(training=True, strong_aug=False)
    tf.crop_size = (120, 150)
    p = np.zeros((2,), dtype=PARAMS_DTYPE)
    p["rw"], p["rh"], p["flow_sx"], p["flow_sy"] = (170, 155), (121, 133), 1, 1
    p["crop_x"], p["crop_y"] = (20, 0), (1, 13)
    out = tf({"imgs": torch.from_numpy(frames).to(DEV), "gt_fw_flows": torch.from_numpy(flows).to(DEV),
              "gt_bw_flows": torch.from_numpy(-flows).to(DEV)}, params=p)
    for b in range(2):
        ref = T.apply_params(dict(frames=frames[b], fw=flows[b], bw=-flows[b]), p[b], 120, 150)
        assert np.array_equal(out["imgs"][0][b].cpu().numpy(), ref["imgs"][0])
        assert np.array_equal(out["gt_fw_flows"][0][b].cpu().numpy(), ref["fw"])
        assert np.array_equal(out["gt_bw_flows"][0][b].cpu().numpy(), ref["bw"])


def test_output_feeds_the_model_batch_layout_and_refuses_host_tensors():
    tf = Transform(training=True, strong_aug=True)
    s = [loader_sample(7, 400, 520), loader_sample(8, 400, 520)]
    out = tf(to_batch(s, True, True, False), rng=np.random.RandomState(3))
    assert [t.shape for t in out["imgs"]] == [torch.Size([2, 3, 384, 384])] * 2
    assert out["gt_fw_flows"][0].shape == (2, 2, 384, 384) and out["imgs"][0].dtype == torch.float32
    assert out["imgs"][0].is_contiguous() and out["gt_bw_flows"][0].is_contiguous()
    with pytest.raises(_lib.RcfHipError):
        tf({"imgs": torch.zeros((1, 2, 400, 520, 3), dtype=torch.uint8)})
    tf.crop_size = (500, 500)                                  # cannot happen with the reference's 400 / 384 constants
    with pytest.raises(ValueError):                            # the reference rescales too-small frames again; refused here
        tf.sample_params(300, 520, np.random.RandomState(0))


@pytest.mark.parametrize("backed_up", [False, True])
def test_batch_uploader_pinned_double_buffer(tmp_path, backed_up):
    """decoded batch -> pinned staging (flows np.load'ed straight into it) -> copy stream -> Transform: the same tensors as
    the plain `torch.from_numpy(...).to(device)` path, over more batches than staging sets (buffers are recycled).
    `backed_up`: the compute stream is busy for ~1 s while the host runs all five batches ahead of it, so the copies of the
    recycled staging sets are still QUEUED when the loader asks for those sets again -- `stage()` itself has to wait for
    them (the loader never calls `wait_host`); a loader that got the set back early would overwrite a batch in flight."""
    import torch
    from rcf_amd.data_pipeline import BatchUploader, Transform, load_flow_npy_into
    from rcf_amd import synth
    H, W, B = 120, 214, 3
    tf = Transform(training=True, strong_aug=True, has_pl=True)
    up = BatchUploader(B, 2, H, W, has_flow=True, has_pl=True, device="cuda:0")
    torch.cuda.synchronize()
    if backed_up:
        torch.cuda._sleep(2_000_000_000)                       # ~1 s of device time ahead of everything below
    runs = []
    for it in range(5):
        samples = [synth.loader_sample(100 * it + b, H, W) for b in range(B)]
        rng = np.random.RandomState(it)
        params = np.stack([tf.sample_params(H, W, rng) for _ in range(B)])
        st = up.stage()
        for b, smp in enumerate(samples):
            st["imgs"][b] = smp["frames"]
            st["pl_masks"][b] = smp["pl"]
            for key, name in (("fw", "gt_fw_flows"), ("bw", "gt_bw_flows")):
                path = str(tmp_path / f"{key}_{it}_{b}.npy")
                np.save(path, smp[key])
                load_flow_npy_into(path, st[name][b])
        got = tf(up.upload(), params=params)
        assert "_release" not in got
        runs.append((samples, params, got))
    torch.cuda.synchronize()
    for it, (samples, params, got) in enumerate(runs):         # the reference path synchronises: after the loop
        ref = tf({"imgs": torch.from_numpy(np.stack([s["frames"] for s in samples])).cuda(),
                  "gt_fw_flows": torch.from_numpy(np.stack([s["fw"] for s in samples])).cuda(),
                  "gt_bw_flows": torch.from_numpy(np.stack([s["bw"] for s in samples])).cuda(),
                  "pl_masks": torch.from_numpy(np.stack([s["pl"] for s in samples])).cuda()}, params=params)
        for k in ("imgs", "gt_fw_flows", "gt_bw_flows", "pl_masks"):
            for a, b_ in zip(got[k], ref[k]):
                assert torch.equal(a, b_), (it, k)
