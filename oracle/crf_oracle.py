"""ORACLE (test infrastructure): ctypes front-end of oracle/crf_ref.c (see its header).
Signature mirrors `torchcrf_cpp.crf_soft/crf_hard` (tools/torchCRF/src/torchcrf.cu:106-149)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libcrf_ref.so")
        src = os.path.join(_HERE, "crf_ref.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libcrf_ref.so"])
        _lib = ctypes.CDLL(so)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def crf_soft_np(rgb, unary, W, H, scomp_smooth, sxy_smooth, scomp_app, sxy_app, srgb_app, iters):
    """rgb [H,W,3] any dtype, unary f32 [H*W,2] -> (map int16 [H,W], Q f32 [H*W,2], (Ls, La))."""
    rgbf = np.ascontiguousarray(rgb, dtype=np.float32)
    un = np.ascontiguousarray(unary, dtype=np.float32)
    assert rgbf.shape == (H, W, 3) and un.shape == (H * W, 2)
    out = np.empty((H, W), dtype=np.int16)
    q = np.empty((H * W, 2), dtype=np.float32)
    nv = np.zeros(2, dtype=np.int32)
    rc = lib().crf_ref_soft(_p(rgbf, ctypes.c_float), _p(un, ctypes.c_float), int(W), int(H),
                            ctypes.c_float(scomp_smooth), ctypes.c_float(sxy_smooth), ctypes.c_float(scomp_app),
                            ctypes.c_float(sxy_app), ctypes.c_float(srgb_app), int(iters),
                            _p(out, ctypes.c_int16), _p(q, ctypes.c_float), _p(nv, ctypes.c_int32))
    assert rc == 0
    return out, q, (int(nv[0]), int(nv[1]))


def dcrf_soft_np(rgb, unary, W, H, scomp_smooth, sxy_smooth, scomp_app, sxy_app, srgb_app, iters):
    """DenseCRF2D semantics (symmetric kernel normalisation, pydensecrf's default): see crf_ref.c lattice_filter.
    Parity-unpinned: restated from the published algorithm, pydensecrf itself is not available here."""
    rgbf = np.ascontiguousarray(rgb, dtype=np.float32)
    un = np.ascontiguousarray(unary, dtype=np.float32)
    assert rgbf.shape == (H, W, 3) and un.shape == (H * W, 2)
    out = np.empty((H, W), dtype=np.int16)
    q = np.empty((H * W, 2), dtype=np.float32)
    nv = np.zeros(2, dtype=np.int32)
    rc = lib().crf_ref_soft_symmetric(_p(rgbf, ctypes.c_float), _p(un, ctypes.c_float), int(W), int(H),
                                      ctypes.c_float(scomp_smooth), ctypes.c_float(sxy_smooth), ctypes.c_float(scomp_app),
                                      ctypes.c_float(sxy_app), ctypes.c_float(srgb_app), int(iters),
                                      _p(out, ctypes.c_int16), _p(q, ctypes.c_float), _p(nv, ctypes.c_int32))
    assert rc == 0
    return out, q, (int(nv[0]), int(nv[1]))


def crf_hard_np(rgb, label, W, H, scomp_smooth, sxy_smooth, scomp_app, sxy_app, srgb_app, confidence, iters):
    rgbf = np.ascontiguousarray(rgb, dtype=np.float32)
    lab = np.ascontiguousarray(label, dtype=np.int16)
    out = np.empty((H, W), dtype=np.int16)
    q = np.empty((H * W, 2), dtype=np.float32)
    nv = np.zeros(2, dtype=np.int32)
    rc = lib().crf_ref_hard(_p(rgbf, ctypes.c_float), _p(lab, ctypes.c_int16), int(W), int(H),
                            ctypes.c_float(scomp_smooth), ctypes.c_float(sxy_smooth), ctypes.c_float(scomp_app),
                            ctypes.c_float(sxy_app), ctypes.c_float(srgb_app), ctypes.c_float(confidence),
                            int(iters), _p(out, ctypes.c_int16), _p(q, ctypes.c_float), _p(nv, ctypes.c_int32))
    assert rc == 0
    return out, q, (int(nv[0]), int(nv[1]))


def lattice_np(feat):
    """feat f32 [n,pd] -> (#vertices, keys int16 [n,pd+1,pd], weights f32 [n,pd+1])."""
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    n, pd = feat.shape
    keys = np.empty((n, pd + 1, pd), dtype=np.int16)
    w = np.empty((n, pd + 1), dtype=np.float32)
    nv = lib().crf_ref_lattice(_p(feat, ctypes.c_float), n, pd, _p(keys, ctypes.c_int16), _p(w, ctypes.c_float))
    assert nv >= 0
    return nv, keys, w


def filter_np(rgb, W, H, sxy, srgb, values):
    rgbf = np.ascontiguousarray(rgb, dtype=np.float32)
    v = np.ascontiguousarray(values, dtype=np.float32)
    out = np.empty_like(v)
    rc = lib().crf_ref_filter(_p(rgbf, ctypes.c_float), int(W), int(H), ctypes.c_float(sxy), ctypes.c_float(srgb),
                              _p(v, ctypes.c_float), _p(out, ctypes.c_float))
    assert rc == 0
    return out


def crf_soft_torch(img, UU, W, H, scomp_smooth, sxy_smooth, scomp, sxy, srgb, iters):
    """Drop-in for torchcrf_cpp.crf_soft on CPU tensors (used by oracle CRFHead)."""
    import torch
    m, _, _ = crf_soft_np(img.numpy(), UU.numpy(), W, H, scomp_smooth, sxy_smooth, scomp, sxy, srgb, iters)
    return torch.from_numpy(m)
