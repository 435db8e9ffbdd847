"""CPU: invariants that pin the C restatement of the reference's dense CRF (oracle/crf_ref.c).
The reference's CUDA source cannot run here and holds no golden vectors: PARITY UNPINNED, so the
restatement is held to the algorithm's own invariants and to an independent brute-force bound."""
import numpy as np

import crf_oracle
from rcf_amd import synth


def _unary(mask, crf_scale=0.7):
    q = (mask * 255.0 / crf_scale).clip(0, 255).astype(np.uint8)
    U = np.clip(q.astype(np.float32) / (np.float32(q.max()) + np.float32(1e-8)), 1e-6, 1 - 1e-6).astype(np.float32)
    return (-np.log(np.stack([1 - U, U], 0))).reshape(2, -1).T.copy().astype(np.float32)


def test_lattice_weights_and_keys():
    g = np.random.Generator(np.random.PCG64(3))
    for pd in (2, 5):
        feat = (g.random((2000, pd)) * 12).astype(np.float32)
        nv, keys, w = crf_oracle.lattice_np(feat)
        assert np.abs(w.sum(1) - 1).max() < 1e-5 and w.min() > -1e-5     # barycentric weights
        assert 0 < nv <= feat.shape[0] * (pd + 1)
        assert nv == len({tuple(k) for k in keys.reshape(-1, pd)})        # vertex count == distinct keys
        # the pd+1 vertices of a pixel are distinct lattice points whose coordinates differ by remainder
        assert all(len({tuple(k) for k in keys[i]}) == pd + 1 for i in range(0, 2000, 97))
        # identical features -> identical simplex
        nv2, keys2, _ = crf_oracle.lattice_np(np.repeat(feat[:1], 5, 0))
        assert nv2 == pd + 1 and (keys2 == keys2[0]).all()


def test_filter_preserves_constants_and_rows_sum_to_one():
    H, W = 40, 56
    rgb = synth.smooth_rgb(H, W, 11)
    const = np.ones((H * W, 2), np.float32) * np.array([0.3, 0.7], np.float32)
    out = crf_oracle.filter_np(rgb, W, H, 60., 5., const)
    assert np.abs(out - const).max() < 1e-5
    m, q, nv = crf_oracle.crf_soft_np(rgb, _unary(synth.soft_blob_mask(H, W, 11)), W, H, 0, 0, 5, 60, 5, 5)
    assert np.abs(q.sum(1) - 1).max() < 1e-6 and q.min() >= 0 and nv[0] == 0 and nv[1] > 0
    assert set(np.unique(m)) <= {0, 1}


def test_zero_weight_and_zero_iterations_give_unary_argmax():
    H, W = 32, 48
    rgb, un = synth.noise_rgb(H, W, 5), _unary(synth.soft_blob_mask(H, W, 5))
    want = (un[:, 1] < un[:, 0]).astype(np.int16).reshape(H, W)
    assert np.array_equal(crf_oracle.crf_soft_np(rgb, un, W, H, 0, 0, 0, 60, 5, 5)[0], want)
    assert np.array_equal(crf_oracle.crf_soft_np(rgb, un, W, H, 0, 0, 5, 60, 5, 0)[0], want)
    assert np.array_equal(crf_oracle.crf_soft_np(rgb, un, W, H, 0, 0, 5, 0, 5, 5)[0], want)     # sigma 0 disables


def test_against_bruteforce_bilateral_meanfield():
    """O(N^2) dense bilateral mean-field as an independent sanity bound (the lattice is an
    approximation of the Gaussian, so only approximate agreement is expected)."""
    H, W, T = 20, 28, 5
    rgb = synth.smooth_rgb(H, W, 21)
    un = _unary(synth.soft_blob_mask(H, W, 21))
    sxy, srgb, w = 8.0, 20.0, 5.0
    m, q, _ = crf_oracle.crf_soft_np(rgb, un, W, H, 0, 0, w, sxy, srgb, T)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    f = np.concatenate([np.stack([xx, yy], -1).reshape(-1, 2) / sxy, rgb.reshape(-1, 3).astype(np.float64) / srgb], 1)
    d2 = ((f[:, None] - f[None]) ** 2).sum(-1)
    K = np.exp(-0.5 * d2)
    K /= K.sum(1, keepdims=True)                     # the lattice normalises by the filtered constant
    Q = np.exp(-un - (-un).max(1, keepdims=True)); Q /= Q.sum(1, keepdims=True)
    for _ in range(T):
        nx = -un + w * (K @ Q)
        Q = np.exp(nx - nx.max(1, keepdims=True)); Q /= Q.sum(1, keepdims=True)
    agree = (Q.argmax(1).reshape(H, W) == m).mean()
    assert agree > 0.93, agree
    assert np.abs(Q[:, 1] - q[:, 1]).mean() < 0.05


def test_hard_labels():
    H, W = 24, 32
    rgb = synth.smooth_rgb(H, W, 9)
    lab = (synth.soft_blob_mask(H, W, 9) > 0.5).astype(np.int16)
    m, q, _ = crf_oracle.crf_hard_np(rgb, lab, W, H, 0, 0, 0, 60, 5, 0.7, 3)
    assert np.array_equal(m, lab)                     # no pairwise term: labels come back
    lab2 = lab.copy(); lab2[:] = -1
    m2, q2, _ = crf_oracle.crf_hard_np(rgb, lab2, W, H, 0, 0, 0, 60, 5, 0.7, 3)
    assert np.abs(q2 - 0.5).max() < 1e-6 and (m2 == 0).all()       # unknown everywhere: ties -> label 0


def test_symmetric_normalisation_against_bruteforce_and_invariants():
    """DenseCRF2D semantics (pydensecrf default, NORMALIZE_SYMMETRIC): Ksym = D^-1/2 K D^-1/2 with D = K 1.
    Parity-unpinned restatement of the published algorithm; held to (i) an O(N^2) dense mean-field with the same
    symmetric normalisation, (ii) symmetry of the operator <a, Ksym b> = <Ksym a, b>, which the row-normalised torchCRF
    filter does not have, (iii) marginals on the simplex, (iv) w = 0 gives the unary arg-max."""
    import ctypes
    H, W, T = 20, 28, 5
    rgb = synth.smooth_rgb(H, W, 21)
    un = _unary(synth.soft_blob_mask(H, W, 21))
    sxy, srgb, w = 8.0, 20.0, 5.0
    m, q, _ = crf_oracle.dcrf_soft_np(rgb, un, W, H, 0, 0, w, sxy, srgb, T)
    assert np.abs(q.sum(1) - 1).max() < 1e-6 and q.min() >= 0
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    f = np.concatenate([np.stack([xx, yy], -1).reshape(-1, 2) / sxy, rgb.reshape(-1, 3).astype(np.float64) / srgb], 1)
    K = np.exp(-0.5 * ((f[:, None] - f[None]) ** 2).sum(-1))
    n = 1.0 / np.sqrt(K.sum(1) + 1e-20)
    Ks = n[:, None] * K * n[None]
    Q = np.exp(-un - (-un).max(1, keepdims=True)); Q /= Q.sum(1, keepdims=True)
    for _ in range(T):
        nx = -un + w * (Ks @ Q)
        Q = np.exp(nx - nx.max(1, keepdims=True)); Q /= Q.sum(1, keepdims=True)
    agree = (Q.argmax(1).reshape(H, W) == m).mean()
    assert agree > 0.93, agree
    assert np.abs(Q[:, 1] - q[:, 1]).mean() < 0.05
    # it is a different operator from the row-normalised one of tools/torchCRF
    m_row, q_row, _ = crf_oracle.crf_soft_np(rgb, un, W, H, 0, 0, w, sxy, srgb, T)
    assert np.abs(q - q_row).max() > 1e-3
    # w = 0: unary arg-max
    want = (un[:, 1] < un[:, 0]).astype(np.int16).reshape(H, W)
    assert np.array_equal(crf_oracle.dcrf_soft_np(rgb, un, W, H, 0, 0, 0, sxy, srgb, 3)[0], want)
