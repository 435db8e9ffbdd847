"""GPU parity of the HIP dense-CRF (csrc/crf.hip) against the C restatement of the reference
(oracle/crf_ref.c) through the C ABI, plus invariants at the full 480x854 size.

Bars: lattice vertex count L identical; marginals Q within 1e-4 absolute (the only intended
differences are __expf vs expf and fp32 summation order); MAP identical on every pixel whose
marginal margin |Q1-Q0| exceeds 1e-3 and >= 99.9 % overall; run-to-run bit-identical output."""
import os

import numpy as np
import pytest
import torch

import crf_oracle
import rcf_amd
from rcf_amd import synth
from rcf_amd.crf import crf_soft_batched

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _unary(mask, crf_scale=0.7):
    q = (mask * 255.0 / crf_scale).clip(0, 255).astype(np.uint8)
    U = np.clip(q.astype(np.float32) / (np.float32(q.max()) + np.float32(1e-8)), 1e-6, 1 - 1e-6).astype(np.float32)
    return (-np.log(np.stack([1 - U, U], 0))).reshape(2, -1).T.copy().astype(np.float32)


def _run_hip(rgbs, unaries, W, H, params, iters):
    rgb = torch.from_numpy(np.stack(rgbs)).to(DEV)
    un = torch.from_numpy(np.stack(unaries)).to(DEV)
    m, q, nv = crf_soft_batched(rgb, un, W, H, *params, iters, want_q=True, want_nvert=True)
    return m.cpu().numpy(), q.cpu().numpy(), nv.cpu().numpy()


@pytest.mark.parametrize("H,W,kind,iters,params", [
    (64, 96, "smooth", 1, (0., 0., 5., 60., 5.)),
    (64, 96, "smooth", 5, (0., 0., 5., 60., 5.)),
    (64, 96, "noise", 5, (0., 0., 5., 60., 5.)),
    (48, 70, "smooth", 5, (3., 3., 5., 60., 5.)),        # smoothness kernel on as well
    (48, 70, "smooth", 3, (3., 3., 0., 60., 5.)),        # smoothness only
    (40, 52, "smooth", 50, (0., 0., 5., 60., 5.)),       # reference default iteration count
    (480, 854, "smooth", 5, (0., 0., 5., 60., 5.)),      # BASELINE config 4
])
def test_crf_soft_vs_oracle(H, W, kind, iters, params, report):
    F = 2
    gen = synth.smooth_rgb if kind == "smooth" else synth.noise_rgb
    rgbs = [gen(H, W, 4100 + i) for i in range(F)]
    uns = [_unary(synth.soft_blob_mask(H, W, 4100 + i)) for i in range(F)]
    m, q, nv = _run_hip(rgbs, uns, W, H, params, iters)
    worst_q, agree_all, mism_sure = 0.0, 1.0, 0
    for f in range(F):
        mo, qo, nvo = crf_oracle.crf_soft_np(rgbs[f], uns[f], W, H, *params, iters)
        assert tuple(nv[f]) == nvo, f"lattice size differs: hip {tuple(nv[f])} oracle {nvo}"
        worst_q = max(worst_q, float(np.abs(q[f] - qo).max()))
        agree_all = min(agree_all, float((m[f] == mo).mean()))
        sure = np.abs(qo[:, 1] - qo[:, 0]).reshape(H, W) > 1e-3
        mism_sure += int((m[f] != mo)[sure].sum())
    report(f"crf {H}x{W} {kind} T={iters} params={params}: L={nv.tolist()} L/N={nv[:, 1].max() / (H * W):.3f} "
           f"max|dQ| {worst_q:.2e} MAP agreement {agree_all:.6f} mismatches on sure pixels {mism_sure}")
    assert worst_q < 1e-4 and agree_all >= 0.999 and mism_sure == 0


def test_crf_deterministic_and_batch_independent(report):
    H, W, F = 120, 214, 3
    rgbs = [synth.smooth_rgb(H, W, 4200 + i) for i in range(F)]
    uns = [_unary(synth.soft_blob_mask(H, W, 4200 + i)) for i in range(F)]
    p = (0., 0., 5., 60., 5.)
    m1, q1, _ = _run_hip(rgbs, uns, W, H, p, 5)
    m2, q2, _ = _run_hip(rgbs, uns, W, H, p, 5)
    assert np.array_equal(m1, m2) and np.array_equal(q1, q2), "CRF output is not run-to-run bit-identical"
    ms, qs, _ = _run_hip(rgbs[1:2], uns[1:2], W, H, p, 5)
    assert np.array_equal(ms[0], m1[1]) and np.array_equal(qs[0], q1[1]), "frame result depends on its batch"
    report("crf determinism: bit-identical across runs and batch compositions")


def test_crf_invariants_fullsize(report):
    H, W = 480, 854
    rgb = synth.noise_rgb(H, W, 4300)                      # worst-case lattice occupancy
    un = _unary(synth.soft_blob_mask(H, W, 4300))
    # zero pairwise weight -> MAP is the unary argmax whatever the iteration count
    m, q, nv = _run_hip([rgb], [un], W, H, (0., 0., 0., 60., 5.), 5)
    assert np.array_equal(m[0].ravel(), (un[:, 1] < un[:, 0]).astype(np.int16))
    assert tuple(nv[0]) == (0, 0)
    # marginals are distributions
    m, q, nv = _run_hip([rgb], [un], W, H, (0., 0., 5., 60., 5.), 5)
    assert np.abs(q[0].sum(1) - 1).max() < 1e-6 and q[0].min() >= 0
    report(f"crf invariants 480x854 noise: L={int(nv[0, 1])} L/N={nv[0, 1] / (H * W):.3f}")
    assert 0 < nv[0, 1] <= 6 * H * W


def test_crf_build_variants_identical(report):
    """the packed 64-bit-key lattice build (default; small table first, all buckets after an overflow) and the
    array-of-keys build give the same MAP, Q and vertex count"""
    from rcf_amd import _lib, synth
    from rcf_amd.crf import crf_soft_batched
    H, W = 120, 214
    head = rcf_amd.CRFHead(None, refine_iters=5)
    imgs = torch.from_numpy(np.stack([synth.normalize_rgb(synth.smooth_rgb(H, W, 4100 + i)) for i in range(3)])).to(DEV)
    masks = torch.from_numpy(np.stack([synth.soft_blob_mask(H, W, 4100 + i) for i in range(3)])).to(DEV)
    rgb, unary = head.prepare(imgs, masks)
    out = {}
    for v in (0, 1, 2):                          # 2: a 1024-bucket first attempt, so every frame takes the overflow path
        out[v] = crf_soft_batched(rgb, unary, W, H, 0.0, 0.0, 5.0, 60.0, 5.0, 5, want_q=True, want_nvert=True, build=v)
    same_map = bool(torch.equal(out[0][0], out[1][0])) and bool(torch.equal(out[0][0], out[2][0]))
    dq = max(float((out[0][1] - out[1][1]).abs().max()), float((out[0][1] - out[2][1]).abs().max()))
    report(f"crf build variants: MAP identical {same_map}, max |dQ| {dq:.2e}, vertices {out[0][2][:, 1].tolist()} vs "
           f"{out[1][2][:, 1].tolist()} vs {out[2][2][:, 1].tolist()} (packed / array-of-keys / packed after a table overflow)")
    assert same_map and dq == 0.0 and torch.equal(out[0][2], out[1][2]) and torch.equal(out[0][2], out[2][2])


@pytest.mark.parametrize("kind,H,W,n", [("smooth", 120, 214, 3), ("noise", 120, 214, 3), ("smooth", 480, 854, 2), ("noise", 480, 854, 2),
                                        ("mixed", 96, 160, 9)])
def test_crf_sort_build_identical(kind, H, W, n, report):
    """RCF_CRF_BUILD_SORT (radix sort / run heads / scan instead of the hash table; vertices numbered in key order, neighbours by
    merging) against the default build: identical MAP, Q to the last bit (the splat's fixed-point sums do not depend on the list
    order, the blur reads the same neighbours), identical vertex counts -- on smooth frames, noise frames, a batch that mixes
    them (9 frames: 4 frame bits), both potentials on"""
    from rcf_amd import synth
    from rcf_amd.crf import crf_soft_batched
    head = rcf_amd.CRFHead(None, refine_iters=5)
    mk = lambda i: (synth.noise_rgb if (kind == "noise" or (kind == "mixed" and i % 2)) else synth.smooth_rgb)(H, W, 4300 + i)
    imgs = torch.from_numpy(np.stack([synth.normalize_rgb(mk(i)) for i in range(n)])).to(DEV)
    masks = torch.from_numpy(np.stack([synth.soft_blob_mask(H, W, 4300 + i) for i in range(n)])).to(DEV)
    rgb, unary = head.prepare(imgs, masks)
    res = {}
    for pots in ((0.0, 0.0, 5.0, 60.0, 5.0), (3.0, 3.0, 5.0, 60.0, 5.0)):
        a = crf_soft_batched(rgb, unary, W, H, *pots, 5, want_q=True, want_nvert=True, build=0)
        b = crf_soft_batched(rgb, unary, W, H, *pots, 5, want_q=True, want_nvert=True, build=3)
        res[pots[0]] = (bool(torch.equal(a[0], b[0])), float((a[1] - b[1]).abs().max()), bool(torch.equal(a[2], b[2])), a[2][:, 1].tolist())
    report(f"crf sort build vs packed build [{kind} {H}x{W} x{n}]: appearance only: MAP equal {res[0.0][0]}, max |dQ| {res[0.0][1]:.1e}, "
           f"vertex counts equal {res[0.0][2]} {res[0.0][3]}; both potentials: {res[3.0][:3]}")
    for r in res.values():
        assert r[0] and r[1] == 0.0 and r[2]


def test_crf_head_picks_the_sort_build_for_noise_like_content(report):
    """CRFHead chooses the lattice build from the vertex counts of its PREVIOUS call (an asynchronous copy, no host wait): noise-like
    frames switch it to the sort build, natural ones back; the masks do not depend on the choice"""
    from rcf_amd import synth
    H, W, n = 480, 854, 2            # (the rule is in vertices per entry: 3 % on natural frames of this size, ~70-90 % on noise)
    head = rcf_amd.CRFHead(None, refine_iters=5)
    ref = rcf_amd.CRFHead(None, refine_iters=5)
    ref.sort_build = False
    seq = []
    for step, make in enumerate([synth.noise_rgb, synth.noise_rgb, synth.noise_rgb, synth.smooth_rgb, synth.smooth_rgb, synth.smooth_rgb]):
        imgs = torch.from_numpy(np.stack([synth.normalize_rgb(make(H, W, 4400 + 10 * step + i)) for i in range(n)])).to(DEV)
        masks = torch.from_numpy(np.stack([synth.soft_blob_mask(H, W, 4400 + i) for i in range(n)])).to(DEV)
        out = head(imgs, masks)
        torch.cuda.synchronize()                  # the test waits so that the next call sees this call's counts
        seq.append(head.last_build)
        assert torch.equal(out, ref(imgs, masks))
    report(f"CRFHead build per call (3 noise calls, then 3 smooth ones): {seq}")
    assert seq[0] == 0 and seq[1] == 3 and seq[2] == 3 and seq[3] == 3 and seq[4] == 0 and seq[5] == 0


def test_offline_callers(report):
    """the offline 480x854 CRF callers (rcf_amd.offline): the pydenseCRF-style `refine` against the C restatement run
    on the same unary, and the double-CRF merge of semantic_constraints.py (unstandardize=False takes NHWC [0,1])"""
    from rcf_amd import offline
    H, W = 96, 130
    img = synth.smooth_rgb(H, W, 4200)                                     # u8 [H,W,3]
    soft = synth.soft_blob_mask(H, W, 4200)
    mask_u8 = (np.asarray(soft) * 255 / 0.8).clip(0, 255).astype(np.uint8)   # tools/pydenseCRF/crf.py:174
    got = offline.refine(mask_u8, img, 0.1, 60.0, 5.0, 5.0, None, iters=10)
    unary = offline._unary_from_u8(mask_u8, 0.1)
    # `refine` runs pydensecrf's symmetric kernel normalisation: checked against the restatement of that algorithm
    ref = crf_oracle.dcrf_soft_np(img, unary, W, H, 0.0, 0.0, 5.0, 60.0, 5.0, 10)[0]
    agree = float((got == ref.astype(np.float32)).mean())
    got_row = offline.refine(mask_u8, img, 0.1, 60.0, 5.0, 5.0, None, iters=10, symmetric=False)
    ref_row = crf_oracle.crf_soft_np(img, unary, W, H, 0.0, 0.0, 5.0, 60.0, 5.0, 10)[0]
    assert float((got_row == ref_row.astype(np.float32)).mean()) > 0.999
    new_mask, iou = offline.refine(mask_u8, img, 0.1, 60.0, 5.0, 5.0, (np.asarray(soft) > 0.5).astype(np.float32), iters=10)
    # double CRF merge: equals the product of the two heads run separately
    head_a = rcf_amd.CRFHead(None, refine_iters=5, crf_scale=0.7)
    head_b = rcf_amd.CRFHead(None, refine_iters=5, crf_scale=0.5)
    imgs = torch.from_numpy(img[None].astype(np.float32) / 255.0).to(DEV)                  # [1,H,W,3] in [0,1]
    m = torch.from_numpy(np.asarray(soft)[None].astype(np.float32)).to(DEV)
    merged = offline.double_crf_merge(head_a, head_b, imgs, m, m)
    rgb, _ = head_a.prepare(imgs, m, unstandardize=False)                  # models/crf_head.py:97-98: *255, clamp, truncate
    same_u8 = bool(torch.equal(rgb, (imgs * 255.0).clamp(0.0, 255.0).to(torch.uint8)))
    prod = head_a(imgs, m, unstandardize=False) * head_b(imgs, m, unstandardize=False)
    e_merge = float((merged - prod).abs().max()) + (0.0 if same_u8 else 1.0)
    report(f"offline refine vs C restatement: pixel agreement {agree:.5f}, iou vs blob {iou:.3f}; double-CRF merge |d| {e_merge}")
    assert agree > 0.999 and 0.0 < iou <= 1.0 and e_merge == 0.0


@pytest.mark.parametrize("H,W,iters", [(64, 96, 5), (96, 130, 50), (480, 854, 5)])
def test_crf_symmetric_normalisation_vs_oracle(H, W, iters, report):
    """rcf_crf_soft_ex(normalization=1) -- DenseCRF2D's symmetric kernel normalisation (pydensecrf default; the CPU
    post-processor tools/pydenseCRF/crf.py:58-89) -- against oracle/crf_ref.c's restatement of it (parity-unpinned)"""
    from rcf_amd.crf import crf_soft_batched
    rgb = synth.smooth_rgb(H, W, 4300)
    soft = np.clip(synth.soft_blob_mask(H, W, 4300), 1e-6, 1 - 1e-6)
    unary = np.stack([-np.log(1 - soft), -np.log(soft)], -1).reshape(-1, 2).astype(np.float32)
    m, q, nv = crf_soft_batched(torch.from_numpy(rgb)[None].to(DEV), torch.from_numpy(unary)[None].to(DEV), W, H, 0.0, 0.0,
                                5.0, 60.0, 5.0, iters, want_q=True, want_nvert=True, symmetric=True)
    mo, qo, nvo = crf_oracle.dcrf_soft_np(rgb, unary, W, H, 0.0, 0.0, 5.0, 60.0, 5.0, iters)
    agree = float((m[0].cpu().numpy() == mo).mean())
    dq = float(np.abs(q[0].cpu().numpy() - qo).max())
    m_row = crf_soft_batched(torch.from_numpy(rgb)[None].to(DEV), torch.from_numpy(unary)[None].to(DEV), W, H, 0.0, 0.0, 5.0,
                             60.0, 5.0, iters)
    differs = float((m_row[0].cpu().numpy() != m[0].cpu().numpy()).mean())
    report(f"crf symmetric normalisation {H}x{W} T={iters}: MAP agreement with the oracle {agree:.6f}, max |dQ| {dq:.2e}, "
           f"vertices {int(nv[0, 1])} vs {nvo[1]}; differs from the row-normalised MAP on {differs:.4f} of the pixels")
    assert agree > 0.9995 and dq < 5e-4 and int(nv[0, 1]) == nvo[1]


def test_crf_hard_vs_oracle(report):
    H, W = 48, 64
    rgb = synth.smooth_rgb(H, W, 4400)
    lab = (synth.soft_blob_mask(H, W, 4400) > 0.5).astype(np.int16)
    lab[::7, ::5] = -1
    out = rcf_amd.crf_hard(torch.from_numpy(rgb).to(DEV), torch.from_numpy(lab).to(DEV), W, H, 3., 3., 5., 60., 5., 0.7, 5)
    mo, _, _ = crf_oracle.crf_hard_np(rgb, lab, W, H, 3., 3., 5., 60., 5., 0.7, 5)
    agree = float((out.cpu().numpy() == mo).mean())
    report(f"crf_hard agreement {agree:.6f}")
    assert agree >= 0.999


def test_crf_head_prepare_vs_reference_golden(golden_dir, report):
    """CRFHead pre-FFI products (u8 image, unary) against what the reference's CRFHead handed to
    torchcrf_cpp.crf_soft (captured in tests/golden/crf_pre.npz)."""
    fx = np.load(os.path.join(golden_dir, "crf_pre.npz"))
    H, W, seed = int(fx["H"]), int(fx["W"]), int(fx["seed"])
    img = torch.from_numpy(synth.normalize_rgb(synth.smooth_rgb(H, W, seed)))[None].to(DEV)
    msk = torch.from_numpy(synth.soft_blob_mask(H, W, seed))[None].to(DEV)
    head = rcf_amd.CRFHead(None)
    rgb, un = head.prepare(img, msk)
    mism = int((rgb[0].cpu().numpy() != fx["img_u8"]).sum())
    e = float(np.abs(un[0].cpu().numpy() - fx["unary"]).max())
    report(f"crf prepare: u8 mismatches {mism}, unary max abs diff {e:.2e}")
    assert mism == 0 and e < 1e-6
    out = head(img, msk)
    assert tuple(out.shape) == (1, H, W) and set(np.unique(out.cpu().numpy())) <= {0.0, 1.0}


def test_crf_soft_float_features_vs_oracle(report):
    """torchcrf_cpp.crf_soft accepts rgbFeat of any dtype and converts it to float UNROUNDED (tools/torchCRF/src/torchcrf.cu:84-85).
    rcf_amd.crf_soft does the same through rcf_crf_soft_f32 (it used to round to u8 silently): a float
    image with non-integer values -- and a few outside [0, 255] -- against the oracle on the same floats; a float image that
    holds integers gives what the uint8 image gives, bit for bit; crf_hard refuses non-integer floats."""
    H, W, iters, p = 64, 96, 5, (0., 0., 5., 60., 5.)
    g = np.random.Generator(np.random.PCG64(77))
    rgb8 = synth.smooth_rgb(H, W, 4400)
    rgbf = rgb8.astype(np.float32) + g.uniform(-0.45, 0.45, size=rgb8.shape).astype(np.float32)
    rgbf[::7, ::5] += 30.0                                         # leaves [0, 255] at the bright end: no clamp either
    un = _unary(synth.soft_blob_mask(H, W, 4400))
    t_un = torch.from_numpy(un).to(DEV)
    m, q, nv = crf_soft_batched(torch.from_numpy(rgbf[None]).to(DEV), t_un[None], W, H, *p, iters, want_q=True, want_nvert=True)
    mo, qo, nvo = crf_oracle.crf_soft_np(rgbf, un, W, H, *p, iters)
    m8o, q8o, nv8o = crf_oracle.crf_soft_np(rgb8, un, W, H, *p, iters)
    dq = float(np.abs(q[0].cpu().numpy() - qo).max())
    sure = np.abs(qo[:, 1] - qo[:, 0]).reshape(H, W) > 1e-3
    mism = int((m[0].cpu().numpy() != mo)[sure].sum())
    moved = float(np.abs(qo - q8o).max())                          # what rounding the features would have changed
    # the module surface (torchcrf_cpp.crf_soft signature): float64 in, converted to float32 unrounded
    m_mod = rcf_amd.crf_soft(torch.from_numpy(rgbf.astype(np.float64)).to(DEV), t_un, W, H, *p, iters)
    same_mod = bool(torch.equal(m_mod, m[0]))
    # integers held in floats == the uint8 image
    m_i = rcf_amd.crf_soft(torch.from_numpy(rgb8.astype(np.float32)).to(DEV), t_un, W, H, *p, iters)
    m_u = rcf_amd.crf_soft(torch.from_numpy(rgb8).to(DEV), t_un, W, H, *p, iters)
    report(f"crf_soft on float features {H}x{W}: vertices {tuple(nv[0].tolist())} vs oracle {nvo} (u8-rounded image: {nv8o}); "
           f"max|dQ| {dq:.2e}, MAP mismatches on sure px {mism}; rounding the features would move Q by {moved:.2e}; "
           f"module surface identical {same_mod}; integer-valued floats == uint8: {bool(torch.equal(m_i, m_u))}")
    assert tuple(nv[0].tolist()) == nvo and nvo != nv8o
    assert dq < 1e-4 and mism == 0 and moved > 10 * dq and same_mod and torch.equal(m_i, m_u)
    lab = torch.from_numpy((synth.soft_blob_mask(H, W, 4400) > 0.5).astype(np.int16)).to(DEV)
    with pytest.raises(RuntimeError, match="non-integer"):
        rcf_amd.crf_hard(torch.from_numpy(rgbf).to(DEV), lab, W, H, *p, 0.5, iters)
    h_i = rcf_amd.crf_hard(torch.from_numpy(rgb8.astype(np.float32)).to(DEV), lab, W, H, *p, 0.5, iters)
    h_u = rcf_amd.crf_hard(torch.from_numpy(rgb8).to(DEV), lab, W, H, *p, 0.5, iters)
    assert torch.equal(h_i, h_u)


@pytest.mark.parametrize("kind,params,iters", [("smooth", (0., 0., 5., 60., 5.), 5), ("noise", (0., 0., 5., 60., 5.), 3),
                                               ("smooth", (3., 3., 5., 60., 5.), 2), ("smooth", (3., 3., 0., 60., 5.), 3)])
def test_crf_blur_pairs_identical(kind, params, iters, report):
    """round 6: a filter's six blur passes run as three launches of two axes each (blur_pair_kernel: the first axis's values are
    recomputed on the fly for the three vertices the second axis reads -- the same float operations on the same values), the
    position lattice's three as a pair + one.  Against one launch per axis (RCF_CRF_BLUR_SEQUENTIAL): MAP, marginals and vertex
    counts bit for bit, on natural and noise frames, with the smoothness kernel on, and through the first pass that carries the
    homogeneous channel along."""
    H, W, F = 120, 214, 3
    gen = synth.smooth_rgb if kind == "smooth" else synth.noise_rgb
    rgb = torch.from_numpy(np.stack([gen(H, W, 4500 + i) for i in range(F)])).to(DEV)
    un = torch.from_numpy(np.stack([_unary(synth.soft_blob_mask(H, W, 4500 + i)) for i in range(F)])).to(DEV)
    out = {}
    for seq in (0, 4):                                   # build bits: 4 = RCF_CRF_BLUR_SEQUENTIAL >> 8
        out[seq] = crf_soft_batched(rgb, un, W, H, *params, iters, want_q=True, want_nvert=True, build=seq)
    same = all(bool(torch.equal(a, b)) for a, b in zip(out[0], out[4]))
    report(f"crf blur pairs vs one launch per axis ({kind}, params {params}, T={iters}): MAP / Q / vertex counts identical: {same}; "
           f"vertices {out[0][2].tolist()}")
    assert same


@pytest.mark.parametrize("kind,size,params,iters,sym", [
    ("smooth", (120, 214), (0., 0., 5., 60., 5.), 5, False), ("smooth", (97, 131), (3., 3., 5., 60., 5.), 3, False),
    ("noise", (64, 80), (0., 0., 5., 60., 5.), 2, False), ("smooth", (120, 214), (0., 0., 5., 60., 5.), 3, True),
    ("smooth", (480, 854), (0., 0., 10., 60., 20.), 5, False)])
def test_crf_tile_splat_identical(kind, size, params, iters, sym, report):
    """round 6: frames whose 16 x 16 pixel tiles share most of their lattice vertices take the splat as a scatter into the tile's
    vertex list in LDS (splat_tile_kernel: 64-bit fixed-point atomics, one flush per distinct vertex of the tile) and build no CSR
    list.  Against the gather over the CSR lists (RCF_CRF_SPLAT_GATHER): MAP, marginals and vertex counts bit for bit -- by the
    default rule, with the tile splat forced on every frame (noise: ~1 536 distinct vertices per tile), on sizes with partial tiles,
    with the position kernel on, under the symmetric normalisation (the stand-alone homogeneous-channel pass), through the
    overflow path of the packed build (small table), and with the slice handing its marginals straight to the next pass's per-tile
    sums (one potential: the default) or not (RCF_CRF_SLICE_SPLAT_SEPARATE)."""
    H, W = size
    F = 3 if H < 400 else 2
    gen = synth.smooth_rgb if kind == "smooth" else synth.noise_rgb
    rgb = torch.from_numpy(np.stack([gen(H, W, 4700 + i) for i in range(F)])).to(DEV)
    un = torch.from_numpy(np.stack([_unary(synth.soft_blob_mask(H, W, 4700 + i)) for i in range(F)])).to(DEV)
    GATHER, TILES, SEPARATE, SMALL = 0x4000 >> 8, 0x8000 >> 8, 0x10000 >> 8, 2
    out = {}
    for name, flags in (("gather", GATHER), ("default", 0), ("tiles", TILES), ("tiles+overflow", TILES | SMALL),
                        ("tiles, slice and sums apart", TILES | SEPARATE)):
        out[name] = crf_soft_batched(rgb, un, W, H, *params, iters, want_q=True, want_nvert=True, symmetric=sym, build=flags)
    same = {k: all(bool(torch.equal(a, b)) for a, b in zip(out["gather"], v)) for k, v in out.items() if k != "gather"}
    report(f"crf tile splat vs gather ({kind} {H}x{W}, params {params}, T={iters}, symmetric {sym}): MAP / Q / vertex counts "
           f"identical: {same}; vertices {out['gather'][2].tolist()}")
    assert all(same.values()), same


def test_crf_forms_agree_on_random_calls(report):
    """40 random calls (sizes 5 ... 200, 1 ... 4 frames, clean / textured / noisy content, one or two potentials, both normalisations, T = 0 ... 3):
    the list walk, the tile splat (default rule and forced, fused and separate slice, small-table overflow) and the sort build give the
    same MAP, marginals and vertex counts bit for bit (tools/fuzz_crf.py runs hundreds of these)."""
    rng = np.random.default_rng(5)
    GATHER, TILES, SEPARATE, SMALL, SORT = 0x4000 >> 8, 0x8000 >> 8, 0x10000 >> 8, 2, 3
    bad = []
    for c in range(40):
        H, W, F, T = int(rng.integers(5, 201)), int(rng.integers(5, 201)), int(rng.integers(1, 5)), int(rng.integers(0, 4))
        amp = int(rng.choice([0, 0, 3, 10, 40, 255]))
        frames = [np.clip(synth.smooth_rgb(H, W, 9500 + 10 * c + i).astype(np.int32) + rng.integers(-amp, amp + 1, (H, W, 3)), 0, 255).astype(np.uint8)
                  for i in range(F)]
        rgb = torch.from_numpy(np.stack(frames)).to(DEV)
        un = torch.from_numpy(np.stack([_unary(synth.soft_blob_mask(H, W, 9500 + 10 * c + i)) for i in range(F)])).to(DEV)
        two = bool(rng.integers(0, 3) == 0)
        sym = bool(not two and rng.integers(0, 3) == 0)
        params = (3.0, 3.0, 5.0, 60.0, 5.0) if two else (0.0, 0.0, 5.0, 60.0, 5.0)
        ref = None
        for name, fl in (("gather", GATHER), ("default", 0), ("tiles", TILES), ("tiles separate", TILES | SEPARATE), ("tiles overflow", TILES | SMALL), ("sort", SORT)):
            r = crf_soft_batched(rgb, un, W, H, *params, T, want_q=True, want_nvert=True, symmetric=sym, build=fl)
            if ref is None:
                ref = r
            elif not all(bool(torch.equal(a, b)) for a, b in zip(ref, r)):
                bad.append((c, H, W, F, T, amp, two, sym, name))
    report(f"CRF forms on 40 random calls: mismatches {bad}")
    assert not bad
