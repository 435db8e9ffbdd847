"""Dense-CRF refinement: `CRFHead` (models/crf_head.py:12-109) and a `torchcrf_cpp`-compatible
module surface (`crf_soft` / `crf_hard`, tools/torchCRF/src/torchcrf.cu:106-149) over the HIP
permutohedral mean-field kernels (csrc/crf.hip).  Frames are batched into ONE library call with a
persistent workspace (the reference loops over frames in Python and cudaMallocs 12 buffers each).
"""
import torch
import torch.nn as nn

from . import _lib
from .ops import _p, _stream, workspace


def _check(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")          # CHECK_CUDA, torchcrf.cu:18
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")             # CHECK_CONTIGUOUS, torchcrf.cu:19


def crf_soft_batched(rgb_u8, unary, W, H, scomp_smooth, sxy_smooth, scomp_app, sxy_app, srgb_app, iters,
                     want_q=False, want_nvert=False, symmetric=False, build=0):
    """rgb_u8 [n,H,W,3] uint8 (or float32: the features as they are, rcf_crf_soft_f32), unary [n,H*W,2] f32 -> MAP int16 [n,H,W]
    (+ Q [n,H*W,2], nvert [n,2]).
    symmetric: pydensecrf's DenseCRF2D kernel normalisation (NORMALIZE_SYMMETRIC) instead of tools/torchCRF's.
    build: 0 default, 1 RCF_CRF_BUILD_ARRAY, 2 RCF_CRF_BUILD_SMALL_TABLE, 3 RCF_CRF_BUILD_SORT (identical results)"""
    _check(rgb_u8, "rgbFeat")
    _check(unary, "unaryEnergy")
    n = rgb_u8.shape[0]
    if tuple(rgb_u8.shape) != (n, H, W, 3) or tuple(unary.shape) != (n, H * W, 2):
        raise RuntimeError("shape check not satisfied")               # CHECK_COND, torchcrf.cu:55-82
    if rgb_u8.dtype not in (torch.uint8, torch.float32) or unary.dtype != torch.float32:
        raise RuntimeError("rgbFeat must be uint8 or float32, unaryEnergy float32")
    dev = rgb_u8.device
    out = torch.empty((n, H, W), dtype=torch.int16, device=dev)
    q = torch.empty((n, H * W, 2), dtype=torch.float32, device=dev) if want_q else None
    nv = torch.empty((n, 2), dtype=torch.int32, device=dev) if want_nvert else None
    need = _lib.load().rcf_crf_workspace_bytes(W, H, n)
    ws = workspace(need, dev)
    entry = "rcf_crf_soft_ex" if rgb_u8.dtype == torch.uint8 else "rcf_crf_soft_f32"
    _lib.call(entry, _p(rgb_u8), _p(unary), W, H, n, scomp_smooth, sxy_smooth, scomp_app, sxy_app, srgb_app,
              int(iters), int(bool(symmetric)) | (int(build) << 8), _p(out), _p(q), _p(nv), _p(ws), need, _stream())
    res = (out,)
    if want_q:
        res += (q,)
    if want_nvert:
        res += (nv,)
    return res if len(res) > 1 else out


def crf_soft(rgbFeat, unaryEnergy, W, H, scompSmooth=3.0, sxySmooth=3.0, scompApp=10.0, sxyApp=60.0, srgbApp=20.0,
             iters=10):
    """torchcrf_cpp.crf_soft: rgbFeat [H,W,3] (any dtype; uint8 on the RCF path), unary [H*W,2] f32.  A non-uint8 rgbFeat is
    converted to float UNROUNDED, as tools/torchCRF/src/torchcrf.cu:84-85 does (`toType(Float)`), and filtered as it is."""
    img = rgbFeat if rgbFeat.dtype == torch.uint8 else rgbFeat.float()
    return crf_soft_batched(img.contiguous()[None], unaryEnergy.float().contiguous()[None], W, H, scompSmooth, sxySmooth, scompApp,
                            sxyApp, srgbApp, iters)[0]


def crf_hard(rgbFeat, label, W, H, scompSmooth=3.0, sxySmooth=3.0, scompApp=10.0, sxyApp=60.0, srgbApp=20.0,
             confidence=0.5, iters=10):
    """torchcrf_cpp.crf_hard: label int16 [H,W] (-1 = unknown).  The C ABI of this entry takes the uint8 image only: a float
    rgbFeat must hold integers in [0, 255] (converted exactly) -- anything else raises instead of being rounded silently
    (the reference would filter the unrounded floats, torchcrf.cu:84-85; use crf_soft for such features)."""
    _check(rgbFeat, "rgbFeat")
    _check(label, "label")
    if rgbFeat.dtype == torch.uint8:
        img = rgbFeat
    else:
        img = rgbFeat.round().clamp(0, 255).to(torch.uint8)
        if not bool((img.to(rgbFeat.dtype) == rgbFeat).all()):
            raise RuntimeError("crf_hard takes uint8 colours (or floats holding integers in [0, 255]); got non-integer features")
    dev = img.device
    out = torch.empty((1, H, W), dtype=torch.int16, device=dev)
    need = _lib.load().rcf_crf_workspace_bytes(W, H, 1)
    ws = workspace(need, dev)
    _lib.call("rcf_crf_hard", _p(img.contiguous()), _p(label.to(torch.int16).contiguous()), W, H, 1, scompSmooth,
              sxySmooth, scompApp, sxyApp, srgbApp, confidence, int(iters), _p(out), None, None, _p(ws), need, _stream())
    return out[0]


class CRFHead(nn.Module):
    def __init__(self, args=None, srgb=5., scomp=5., sxy=60., scomp_smooth=0., sxy_smooth=0., refine_iters=50,
                 crf_scale=0.7, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        super().__init__()
        self.args = args
        self.srgb, self.scomp, self.sxy = srgb, scomp, sxy
        self.scomp_smooth, self.sxy_smooth = scomp_smooth, sxy_smooth
        self.refine_iters, self.crf_scale = refine_iters, crf_scale
        self._mean, self._std = tuple(mean), tuple(std)
        self._consts = None
        # lattice build of the next call: "auto" = from the vertex counts of the PREVIOUS call (noise-like content -- more than
        # SORT_ABOVE x (6 entries per pixel) vertices per frame in the appearance lattice -- takes the sort build,
        # RCF_CRF_BUILD_SORT, until a call comes in below SORT_BELOW); True / False = always / never.  Measured with the round-6
        # tile splat (tools/crf_texture_sweep.py, profiles/r06_crf_texture_sweep.txt: 480x854 frames with more and more pixel
        # noise): at 10 % vertices per entry the packed build is ahead (0.50 vs 0.53 ms per frame, T = 5; per pass equal), at 14 %
        # the sort build (0.60 vs 0.65; per pass 439 vs 503 us); clean natural frames sit at 3 % (0.15 vs 0.41).  The counts travel
        # to the host by an asynchronous copy that is never waited for, and the masks do not depend on the choice.
        # The choice is a heuristic on counts that are one or two calls old and read without synchronisation: it may differ from
        # run to run and rank to rank BY DESIGN -- every build gives bit-identical masks (tests/test_crf_gpu.py).
        self.sort_build = "auto"
        self.last_build = 0
        self._nv_host = None
        self._sorting = False

    SORT_ABOVE, SORT_BELOW = 0.12, 0.09          # vertices per entry (6 entries per pixel)

    def _pick_build(self, npix):
        if self.sort_build != "auto":
            return 3 if self.sort_build else 0
        if self._nv_host is not None:
            # whatever the last completed copy left there (zeros before the first one): a count that is one or two calls old
            # serves a heuristic as well as a fresh one, and no event / query is needed (a recorded event per call cost
            # 0.3 ms per frame here: its release flushes the caches the next call's kernels were about to hit)
            verts = float(self._nv_host[:, 1].float().mean())
            if verts > 0:
                self._sorting = verts > (self.SORT_BELOW if self._sorting else self.SORT_ABOVE) * 6 * npix
        return 3 if self._sorting else 0

    def _note_counts(self, nv):
        if self._nv_host is None or self._nv_host.shape[0] != nv.shape[0]:
            self._nv_host = torch.zeros(tuple(nv.shape), dtype=nv.dtype).pin_memory()
        self._nv_host.copy_(nv, non_blocking=True)

    def _mean_std(self, device):
        if self._consts is None or self._consts[0].device != device:
            self._consts = (torch.tensor(self._mean, dtype=torch.float32, device=device),
                            torch.tensor(self._std, dtype=torch.float32, device=device))
        return self._consts

    def prepare(self, imgs, masks, unstandardize=True):
        """models/crf_head.py:33-37,43-55,95-98 on the GPU: -> (u8 image [N,H,W,3], unary [N,HW,2])."""
        if not unstandardize:
            # models/crf_head.py:93-98: without `unnormalize` there is no permute either -- the caller passes
            # [N,H,W,3] images already in [0,1] (tools/SemanticConstraintsAndMAA/semantic_constraints.py:293-306)
            if imgs.dim() != 4 or imgs.shape[-1] != 3:
                raise RuntimeError("CRFHead(unstandardize=False) takes [N,H,W,3] images in [0,1]")
            imgs = imgs.permute(0, 3, 1, 2)
        imgs, masks = imgs.contiguous().float(), masks.contiguous().float()
        N, _, H, W = imgs.shape
        mean, std = self._mean_std(imgs.device)
        rgb = torch.empty((N, H, W, 3), dtype=torch.uint8, device=imgs.device)
        unary = torch.empty((N, H * W, 2), dtype=torch.float32, device=imgs.device)
        scratch = torch.empty(N, dtype=torch.int32, device=imgs.device)
        _lib.call("rcf_crf_prepare", _p(imgs), _p(masks), _p(mean), _p(std), int(unstandardize), self.crf_scale,
                  _p(rgb), _p(unary), _p(scratch), N, H, W, _stream())
        return rgb, unary

    @torch.no_grad()
    def forward(self, imgs, masks, unstandardize=True, symmetric=False):
        """imgs [N,3,H,W] normalised, masks [N,H,W] in [0,1] -> refined masks [N,H,W] float 0/1.
        symmetric=True: the arithmetic of the reference's `crf_cpu` (models/crf_head.py:62-91: pydensecrf DenseCRF2D,
        symmetric kernel normalisation) instead of `crf` (torchcrf_cpp), batched on the GPU all the same."""
        rgb, unary = self.prepare(imgs, masks, unstandardize)
        N, H, W, _ = rgb.shape
        auto = self.sort_build == "auto"
        self.last_build = self._pick_build(H * W)
        if self.last_build == 3 and N > 16:
            self.last_build = 0              # csrc/crf.hip build_lattice: the sort build takes at most 16 frames per call (4 frame
                                             # bits beside the 60 key bits) -- report the build that runs, not the one asked for
        m = crf_soft_batched(rgb, unary, W, H, self.scomp_smooth, self.sxy_smooth, self.scomp, self.sxy, self.srgb,
                             self.refine_iters, symmetric=symmetric, want_nvert=auto, build=self.last_build)
        if auto:
            m, nv = m
            self._note_counts(nv)
        return m.float()
