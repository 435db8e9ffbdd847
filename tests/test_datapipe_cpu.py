"""Data transform, CPU half: the host logic that draws the random decisions
(rcf_amd.data_pipeline.Transform.sample_params) and the oracle's restatement of the pipeline (oracle/transforms_np.py) against
the fixtures the reference's own `Transform` produced (tests/golden/make_golden_data.py).  The cv2 operators under the
reference (bilinear / nearest resize, RGB<->HSV) are restated from OpenCV's published algorithms and are parity-unpinned;
what is pinned here is the reference's own code around them (order of the random draws, scale rule, crop, flip, the
photometric chain, flow / pseudo-label handling, /255, normalisation)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import transforms_np as T
from rcf_amd.data_pipeline import PARAMS_DTYPE, Transform, rescale_size
from rcf_amd.synth import loader_sample

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "data_pipeline.json")))
SUBS = np.load(os.path.join(HERE, "golden", "data_pipeline.npz"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def check(got, rec, what):
    """exact: the digest of the whole tensor; the subsample only serves the failure message"""
    assert list(got.shape) == rec["shape"], what
    if sha(got) != rec["sha256"]:
        sub = got[..., ::GOLD["stride"], ::GOLD["stride"]]
        d = np.abs(sub - SUBS[rec["sub"]])
        raise AssertionError(f"{what}: digest differs; on the subsample {int((d > 0).sum())} of {d.size} values differ, max {d.max():.3e}")


def sample_for(case, seed):
    c = GOLD["cases"][case]
    s = loader_sample(seed, c["H"], c["W"])
    if not (c["training"] and c["kwargs"].get("has_flow", True)):
        s["fw"] = s["bw"] = None
    if not (c["training"] and c["kwargs"].get("has_pl", False)):
        s["pl"] = None
    return s


def test_params_struct_matches_the_c_layout():
    # include/rcf_hip.h `struct rcf_aug_params`: 6 x int32, 5 x float, pad, 1 x double
    assert PARAMS_DTYPE.itemsize == 56
    assert PARAMS_DTYPE.fields["hue_delta"][1] == 48 and PARAMS_DTYPE.fields["beta"][1] == 24


@pytest.mark.parametrize("case", sorted(GOLD["cases"]))
def test_host_decisions_and_oracle_pipeline_match_the_reference(case):
    c = GOLD["cases"][case]
    tf = Transform(training=c["training"], **c["kwargs"])
    for rec in c["samples"]:
        np.random.seed(rec["seed"])
        p = tf.sample_params(c["H"], c["W"])                       # draws from numpy's global generator, as the reference
        assert (int(p["rw"]), int(p["rh"])) == rescale_size(c["W"], c["H"], tuple(rec["scale"])), "scale draw"
        assert bool(p["flip"]) == rec["flip"], "flip draw"
        oh, ow = tf.output_size([p])
        out = T.apply_params(sample_for(case, rec["seed"]), p, oh, ow)
        for i, r in enumerate(rec["imgs"]):
            check(out["imgs"][i], r, f"{case} seed {rec['seed']} frame {i} ops {int(p['ops']):05b}")
        for k in ("fw", "bw"):
            if k in rec:
                check(out[k], rec[k], f"{case} seed {rec['seed']} {k}")
        for i, r in enumerate(rec.get("pl", [])):
            check(out["pl"][i], r, f"{case} seed {rec['seed']} pl {i}")


def test_fixture_covers_every_photometric_branch():
    tf = Transform(training=True, strong_aug=True)
    seen, flips = 0, set()
    for case in ("train_strong_pl", "train_strong_small"):
        c = GOLD["cases"][case]
        for rec in c["samples"]:
            np.random.seed(rec["seed"])
            p = tf.sample_params(c["H"], c["W"])
            ops = int(p["ops"])
            seen |= ops if ops & 2 else ops & ~16             # "contrast last" only counts when contrast is applied
            seen |= 32 if (ops & 2) and not (ops & 16) else 0    # contrast first
            flips.add(bool(p["flip"]))
    assert seen == 63 and flips == {True, False}


def test_resize_restatement_properties():
    # constant images stay constant, identity size is the identity, nearest picks source pixels
    g = np.random.default_rng(0)
    img = g.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    assert np.array_equal(T.resize_linear_u8(img, (53, 37)), img)
    const = np.full((40, 60, 3), 77, dtype=np.uint8)
    assert np.all(T.resize_linear_u8(const, (45, 31)) == 77) and np.all(T.resize_linear_u8(const, (90, 70)) == 77)
    nn = T.resize_nearest(img, (31, 20))
    assert nn.shape == (20, 31, 3) and np.array_equal(nn[0, 0], img[0, 0])
    assert T.rescale_size((854, 480), (9799, 392)) == ((697, 392), 392 / 480)


def test_hsv_restatement_round_trip():
    # RGB -> HSV -> RGB on 8-bit data comes back within the quantisation of H (2 degrees) and S
    g = np.random.default_rng(1)
    img = g.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)
    back = T.hsv2rgb(T.rgb2hsv(img))
    assert np.abs(back.astype(int) - img.astype(int)).max() <= 6
    gray = np.repeat(g.integers(0, 256, size=(8, 8, 1), dtype=np.uint8), 3, axis=2)
    hsv = T.rgb2hsv(gray)
    assert np.all(hsv[..., 0] == 0) and np.all(hsv[..., 1] == 0) and np.array_equal(T.hsv2rgb(hsv), gray)
    prim = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255]]], dtype=np.uint8)
    assert T.rgb2hsv(prim)[0, :, 0].tolist() == [0, 60, 120]


def test_npy_flow_reader_fills_a_given_buffer(tmp_path):
    """dataset/data.py:122-128 `np.load(flow_path)`: the payload of a v1 / v2 `.npy` file lands in the caller's (pinned)
    buffer without a temporary; files of another dtype / order take numpy's general path"""
    from rcf_amd.data_pipeline import load_flow_npy_into
    g = np.random.RandomState(3)
    a = g.randn(13, 17, 2).astype(np.float32)
    for name, arr in (("v1.npy", a), ("f64.npy", a.astype(np.float64)), ("fortran.npy", np.asfortranarray(a))):
        np.save(tmp_path / name, arr)
        out = np.full(a.shape, 7, np.float32)
        load_flow_npy_into(str(tmp_path / name), out)
        assert np.array_equal(out, a), name
    with open(tmp_path / "v2.npy", "wb") as f:
        np.lib.format.write_array(f, a, version=(2, 0))
    out = np.empty_like(a)
    load_flow_npy_into(str(tmp_path / "v2.npy"), out)
    assert np.array_equal(out, a)
    np.save(tmp_path / "short.npy", a)
    data = open(tmp_path / "short.npy", "rb").read()
    open(tmp_path / "short.npy", "wb").write(data[:-100])
    import pytest
    with pytest.raises((IOError, ValueError)):
        load_flow_npy_into(str(tmp_path / "short.npy"), np.empty_like(a))
