"""Tensor-level wrappers over the C ABI (include/rcf_hip.h).

torch is used here only as the owner of device memory and of the HIP stream: every function
takes/returns torch CUDA tensors, pulls raw device pointers out of them and launches the
hand-written kernels in librcf_hip.so on torch's current stream.  No torch arithmetic.

Activation convention: NHWC fp32 tensors of shape [N,H,W,C] whose last dim is contiguous; a
tensor may be a channel slice of a wider buffer (pitch = stride(2) > C).
Conv weights: torch shape [Cout,Cin,R,S] in channels_last memory format, i.e. [Cout][R][S][Cin]
in memory.
"""
import ctypes
import os
from ctypes import c_void_p, byref

import torch

from . import _lib
from ._lib import ConvShape, call


def _p(t):
    return None if t is None else c_void_p(t.data_ptr())


H16 = (torch.bfloat16, torch.float16)          # the 16-bit storage types (one per library build)


def half():
    """the 16-bit storage type of the library calls go to right now (torch.bfloat16, or torch.float16 inside half_storage(float16))"""
    return torch.float16 if _lib.ACTIVE == "f16" else torch.bfloat16


class half_storage:
    """with half_storage(torch.float16): every library call goes to librcf_hip_f16.so (IEEE fp16 as the 16-bit storage type --
    Lightning's `precision: 16`, configs/rcf_stv2/rcf_stage1.yaml:57-60); torch.bfloat16 -> librcf_hip.so (the default); any
    other dtype (fp32 models) leaves the choice alone.  RCFModel wraps its forward and its backward in it."""

    def __init__(self, dtype):
        self.want = "f16" if dtype == torch.float16 else ("bf16" if dtype == torch.bfloat16 else None)

    def __enter__(self):
        self.old = _lib.ACTIVE
        if self.want is not None:
            _lib.ACTIVE = self.want
        return self

    def __exit__(self, *exc):
        _lib.ACTIVE = self.old
        return False


def _dt(t):
    """storage type code of an activation tensor (RCF_F32 / RCF_BF16 of include/rcf_hip.h; RCF_BF16 = "the 16-bit type of the
    library the call goes to": a float16 tensor must not reach the bf16 build, nor the other way round)"""
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype in H16:
        if t.dtype != half():
            raise _lib.RcfHipError(f"a {t.dtype} tensor was handed to the library that stores {half()}: wrap the call in "
                                   "rcf_amd.ops.half_storage(dtype)")
        return _lib.BF16
    raise _lib.RcfHipError(f"unsupported activation dtype {t.dtype}")


def _same_dt(*ts):
    d = _dt(ts[0])
    for t in ts[1:]:
        if t is not None and _dt(t) != d:
            raise _lib.RcfHipError("tensors of one call must share their storage type")
    return d


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.RcfHipError("rcf_amd ops need CUDA (HIP) tensors: there is no CPU fallback")


def pitch_of(x):
    """Pixel pitch of an NHWC activation view; validates the layout."""
    N, H, W, C = x.shape
    if C > 1 and x.stride(3) != 1:
        raise ValueError("NHWC activation must be channel-contiguous")
    if W > 1:
        p = x.stride(2)
    elif H > 1:
        p = x.stride(1)
    elif N > 1:
        p = x.stride(0)
    else:
        p = C
    ok = (H == 1 or x.stride(1) == W * p) and (N == 1 or x.stride(0) == H * W * p) and p >= C
    if not ok:
        raise ValueError(f"not a pitched NHWC view: shape {tuple(x.shape)} strides {x.stride()}")
    return p


class _Profile:
    """Live per-kernel timing with HIP events on the launch stream (bench.py's `roofline` legs): every launch of a
    selected kernel family is bracketed by two events recorded on the stream it is launched on (weight gradients: the
    second stream).  bench.py selects ONE family inside the timed region, so the timed steps are barely perturbed, and
    all of them in a short separate pass (`roofline_by_kernel`)."""

    def __init__(self):
        self.which, self.records = None, {}

    def start(self, which):
        """which: a family name, or a collection of names"""
        self.which = {which} if isinstance(which, str) else set(which)
        self.records = {w: [] for w in self.which}

    def bracket(self, which, flops, tag=None):
        if self.which is None or which not in self.which:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records[which].append((flops, e0, e1, tag))
        e0.record()
        return e1

    def relabel_last(self, which, new):
        """the launch just bracketed under `which` turned out to run another kernel (the C side picks it): file it under `new`
        -- or drop it when `new` is not being recorded"""
        if self.which is None or which not in self.which or not self.records[which]:
            return
        rec = self.records[which].pop()
        if new in self.which:
            self.records[new].append(rec)

    def stop_detail(self):
        """{(family, tag): {launches, flops, ms}}: the brackets of stop(), kept apart by the launch's shape tag
        (tools/layer_table.py)"""
        torch.cuda.synchronize()
        out = {}
        for w, r in self.records.items():
            for flops, e0, e1, tag in r:
                d = out.setdefault((w, tag), {"launches": 0, "flops": 0.0, "ms": 0.0})
                d["launches"] += 1
                d["flops"] += float(flops)
                d["ms"] += float(e0.elapsed_time(e1))
        self.which, self.records = None, {}
        return out

    def stop(self):
        """{family: {launches, flops, ms}}; a single selected family is returned directly"""
        torch.cuda.synchronize()
        out = {w: {"launches": len(r), "flops": float(sum(x[0] for x in r)),
                   "ms": float(sum(x[1].elapsed_time(x[2]) for x in r))} for w, r in self.records.items()}
        single = len(self.which) == 1
        self.which, self.records = None, {}
        return next(iter(out.values())) if single else out


PROFILE = _Profile()
_workspaces = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer per device (stream-ordered reuse on the current stream)."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def weight_rsck(w):
    """[Cout,Cin,R,S] parameter -> its [Cout,R,S,Cin] memory view (must be channels_last)."""
    v = w.permute(0, 2, 3, 1)
    if not v.is_contiguous():
        raise ValueError("conv weight must be in channels_last memory format ([Cout][R][S][Cin])")
    return v


def conv_out_size(n, k, stride, pad, dil):
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


def _conv_shape(xshape, x_pitch, w, stride, pad, dil, y_pitch=None, amax=None, w_pairs=None, w_pairs_t=None, w_pairs2_t=None,
                flags=0, amax_y=None, nt_cols=0):
    """amax = (amax_x, amax_w, amax_dy): int32 [1] device tensors from absmax() or None -- the operand ranges that
    select the fp16-pair kernels (rcf_conv_shape in include/rcf_hip.h).
    nt_cols: output columns of a forward / data-gradient launch -- from SCHED.nt_stores columns up the tile kernels write
    their output with streaming stores (RCF_CONV_NT_STORES)"""
    if SCHED.nt_stores and nt_cols >= SCHED.nt_stores:
        flags = int(flags) | _lib.CONV_NT_STORES
    N, H, W, Cin = xshape
    Cout, Cin_w, R, S = w.shape
    assert Cin == Cin_w, f"channel mismatch {Cin} vs {Cin_w}"
    Ho, Wo = conv_out_size(H, R, stride, pad, dil), conv_out_size(W, S, stride, pad, dil)
    ax, aw, ady = amax if amax is not None else (None, None, None)
    kind = "w" if aw is None else ("d" if ax is None else "f")
    if kind not in SCHED.h2_kinds:                      # debug knob: which launches may take the fp16-pair kernels
        ax = aw = ady = w_pairs = w_pairs2_t = None
    return ConvShape(N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, x_pitch, y_pitch or Cout,
                     _addr(ax), _addr(aw), _addr(ady), None, _addr(w_pairs_t), _addr(w_pairs), _addr(w_pairs2_t), _addr(amax_y),
                     int(flags) | CONV_FLAGS, ctypes.sizeof(ConvShape))


def _addr(t):
    return None if t is None else t.data_ptr()


# RCF_CONV_* bits OR-ed into every launch's rcf_conv_shape.flags: the A/B switches of tests and tools live HERE, in the
# Python process that drives the library, and travel with each call -- the C library keeps no mutable global state
CONV_FLAGS = 0


_amax_pool = {}
from .config import SCHED  # noqa: E402  (SCHED.h2_kinds: which conv directions may take the fp16-pair kernels)


def new_amax(device):
    """a zeroed int32 [1] device scalar for a tensor's range.  Slots come from a zero-filled pool and are never handed
    out twice, so no fill kernel runs per range."""
    key = torch.device(device).index or 0
    pool = _amax_pool.get(key)
    if pool is None or pool[1] >= pool[0].numel():
        pool = _amax_pool[key] = [torch.zeros(8192, dtype=torch.int32, device=device), 0]
    i = pool[1]
    pool[1] = i + 1
    return pool[0][i:i + 1]


def reserve_amax(device, n):
    """make sure the next `n` new_amax slots come from a pool that was zero-filled on the CURRENT stream (call before
    forking work to another stream: a pool refilled over there would be zeroed in that stream's order only)"""
    key = torch.device(device).index or 0
    pool = _amax_pool.get(key)
    if pool is None or pool[1] + n > pool[0].numel():
        _amax_pool[key] = [torch.zeros(8192, dtype=torch.int32, device=device), 0]


def absmax(x, out=None):
    """int32 [1] on the device: raw fp32 bits of max |x| (x: NHWC activation, possibly a channel slice, or any
    contiguous fp32 tensor).  `out` accumulates (max with its current content)."""
    _need_cuda(x)
    if out is None:
        out = new_amax(x.device)
    if x.dim() == 4 and x.stride(3) == 1 and x.shape[3] % 4 == 0 and not x.is_contiguous():
        rows, C, pitch = _rows(x), x.shape[3], pitch_of(x)
    else:
        assert x.is_contiguous() and x.numel() % 4 == 0
        rows, C, pitch = x.numel() // 4, 4, 4
    call("rcf_absmax_f32", _p(x), rows, C, pitch, _p(out), _stream())
    return out


# Operands derived from a weight (its range, fp16 planes, bf16 copies) are cached by the layers between weight updates.
# In-place updates made through torch on the parameter itself (optimizers, `p.copy_`, `p.mul_`) bump the tensor's
# `_version`; writes through `p.data` / `p.detach()` views do NOT, and neither do the kernels of this library (raw
# pointers) -- those writers bump this epoch instead: the library's own wrappers (Adam, EMA), `copy_param_and_buffer`,
# `RCFModel.train()/eval()` and its load_state_dict hook do; an external `.data` writer must call `weights_changed()`.
# RCF_DEBUG_WEIGHT_CACHE=1 re-validates every cache hit against a checksum of the weight (two reductions per launch).
WEIGHT_EPOCH = [0]
DEBUG_WEIGHT_CACHE = os.environ.get("RCF_DEBUG_WEIGHT_CACHE", "0") == "1"


def weights_changed(module=None):
    """weights were written behind torch's back.  Without an argument every cached derived operand of the process is
    dropped (the epoch moves); with a module only the caches of that module's conv layers (e.g. the EMA teacher after its
    momentum update: the student's operands, prepared in bulk after the optimizer step, stay valid)."""
    if module is None:
        WEIGHT_EPOCH[0] += 1
        return
    for m in module.modules():
        m.__dict__.pop("_wcache", None)


def weight_key(w):
    return (w.data_ptr(), w._version, WEIGHT_EPOCH[0])


def weight_checksum(w):
    """debug mode only: (max |w|, sum w) as Python floats -- a stale cache entry shows up as a changed checksum"""
    d = w.detach()
    return (float(d.abs().max()), float(d.double().sum()))


def weight_pairs_t(w, amax_w):
    """the transposed fp16-pair planes the data gradient contracts against (once per weight update, not per launch), in
    both reading orders (rcf_conv_weight_pairs2_f32, transpose = 1): conv2d_dgrad(w_pairs_t=...)"""
    _need_cuda(w)
    Cout, Cin, R, S = w.shape
    planes = torch.empty(_lib.load().rcf_conv_weight_pairs2_bytes(Cout, Cin, R, S, 1), dtype=torch.uint8, device=w.device)
    call("rcf_conv_weight_pairs2_f32", _p(weight_rsck(w)), Cout, Cin, R, S, 1, _p(amax_w), _p(planes), CONV_FLAGS, _stream())
    return planes


def weight_pairs(w, amax_w):
    """the fp16-pair split of a conv weight (channels_last [Cout,Cin,R,S]) for the forward launches that read it: a uint8
    buffer holding the two fp16 planes in both reading orders (rcf_conv_weight_pairs2_f32: the 128 x 256 kernel's and the
    persistent LDS-DMA kernel's)"""
    _need_cuda(w)
    Cout, Cin, R, S = w.shape
    planes = torch.empty(_lib.load().rcf_conv_weight_pairs2_bytes(Cout, Cin, R, S, 0), dtype=torch.uint8, device=w.device)
    call("rcf_conv_weight_pairs2_f32", _p(weight_rsck(w)), Cout, Cin, R, S, 0, _p(amax_w), _p(planes), CONV_FLAGS, _stream())
    return planes


def set_conv_flags(flags):
    """RCF_CONV_* bits (rcf_amd._lib.CONV_*) to OR into every conv launch of this process from now on -- the A/B switches of
    tests and tools.  Returns the previous value.  The K order is baked into the cached weight operands: they are dropped."""
    global CONV_FLAGS
    old, CONV_FLAGS = CONV_FLAGS, int(flags)
    if (old ^ CONV_FLAGS) & _lib.CONV_KORDER_NATURAL:
        weights_changed()
    return old


if os.environ.get("RCF_CONV_FLAGS"):               # e.g. RCF_CONV_FLAGS=0x8 keeps every conv off the persistent kernel
    CONV_FLAGS = int(os.environ["RCF_CONV_FLAGS"], 0)


def _relabel_conv(s, region, dgrad, family, h2p_family):
    """profiling only: a forward / data-gradient launch that takes the persistent kernel is filed under its own family
    (rcf_conv_kernel_of: a pure function of the launch's arguments)"""
    if _lib.load().rcf_conv_kernel_of(byref(s), region, int(dgrad)) == 2:
        PROFILE.relabel_last(family, h2p_family)


def fused_stats_available():
    """regions and the fused batch-norm statistics exist on the default kernels, not on the fp32-MFMA test path"""
    return not (CONV_FLAGS >> 12) & 7


def _shape_tag(s, region=None):
    return (f"{s.Cin}->{s.Cout} k{s.R} d{s.dil} s{s.stride} {s.N}x{s.Ho}x{s.Wo}" + ("" if region is None else f" region{tuple(int(v) for v in region)}"))


def _region_pixels(region, H, W):
    if region is None:
        return H * W
    r = [int(v) for v in region] + [0]
    return r[2] * r[3] if r[4] <= 0 else 2 * r[4] * r[3] + 2 * r[4] * (r[2] - 2 * r[4])


def _region(region):
    """(y0, x0, h, w[, band]) -> ctypes pointer (None = the whole tensor); band t > 0 = only the rectangle's border
    frame of thickness t"""
    if region is None:
        return None
    r = [int(v) for v in region]
    return byref(_lib.ConvRegion(*(r + [0] * (5 - len(r)))))


def conv2d_fwd(x, w, bias=None, stride=1, pad=0, dil=1, act=0, slope=0.0, out=None, beta=0, region=None, amax=None,
               w_pairs=None, x_planes=False, amax_y=None):
    """region = (y0, x0, h, w) in output coordinates: only those pixels of `out` are written.
    amax = (amax_x, amax_w): operand ranges (absmax) -> fp16-pair kernels; w_pairs: weight_pairs(w, amax_w).
    x_planes: `x` (an fp32-typed tensor of the activation's shape) holds fp16 pair planes (RCF_CONV_X_PLANES: bn_apply's
    `planes`); amax_y: new_amax() slot that receives the range of the output"""
    _need_cuda(x, w)
    s = _conv_shape(x.shape, pitch_of(x), w, stride, pad, dil, amax=None if amax is None else (amax[0], amax[1], None),
                    w_pairs=w_pairs, flags=_lib.CONV_X_PLANES if x_planes else 0, amax_y=amax_y, nt_cols=w.shape[0])
    if out is None:
        out = torch.empty((s.N, s.Ho, s.Wo, s.Cout), dtype=torch.float32, device=x.device)
    s.y_pitch = pitch_of(out)
    end = None
    if PROFILE.which is not None:       # forward launches of the 128x256-tile kernel instance / of the narrower tiles
        end = PROFILE.bracket(("conv_h2d_fwd" if s.Cout > 128 else "conv_h2d_fwd_narrow") if x_planes else "conv_x3_128x256" if s.Cout > 128 else "conv_fwd_narrow",
                              2.0 * s.N * _region_pixels(region, s.Ho, s.Wo) * s.Cout * s.R * s.S * s.Cin, _shape_tag(s, region))
    call("rcf_conv2d_fwd_region_f32", _p(x), _p(weight_rsck(w)), _p(bias), _p(out), byref(s), _region(region), act,
         slope, beta, _stream())
    if end is not None:
        end.record()
        if not x_planes:
            _relabel_conv(s, _region(region), 0, "conv_x3_128x256" if s.Cout > 128 else "conv_fwd_narrow", "conv_h2p_fwd")
    return out


def _bn_fin(bn, count, device):
    """(struct rcf_bn_finalize, mean, invstd) for a training-mode batch norm whose statistics the next launch produces:
    its final reduction also writes the normalisation constants, the running statistics and num_batches_tracked"""
    C = bn.num_features
    mean = torch.empty(C, dtype=torch.float32, device=device)
    invstd = torch.empty(C, dtype=torch.float32, device=device)
    fin = _lib.BnFinalize(float(count), bn.eps, bn.momentum, mean.data_ptr(), invstd.data_ptr(), bn.running_mean.data_ptr(),
                          bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr())
    return fin, mean, invstd


def conv2d_fwd_stats(x, w, stride=1, pad=0, dil=1, amax=None, w_pairs=None, bn=None, x_planes=False, amax_y=None):
    """conv (no bias) whose epilogue also yields the batch-norm statistics of the output: (y, fp64 [2*Cout] sums); with
    `bn` (a training-mode BatchNorm2d whose statistics are local) the same reduction finalizes it: (y, (mean, invstd, count)).
    x_planes / amax_y: as conv2d_fwd"""
    _need_cuda(x, w)
    s = _conv_shape(x.shape, pitch_of(x), w, stride, pad, dil, amax=None if amax is None else (amax[0], amax[1], None),
                    w_pairs=w_pairs, flags=_lib.CONV_X_PLANES if x_planes else 0, amax_y=amax_y, nt_cols=w.shape[0])
    out = torch.empty((s.N, s.Ho, s.Wo, s.Cout), dtype=torch.float32, device=x.device)
    s.y_pitch = pitch_of(out)
    sums = torch.empty(2 * s.Cout, dtype=torch.float64, device=x.device)
    need = _lib.load().rcf_conv2d_fwd_stats_workspace_bytes(byref(s))
    ws = workspace(need, x.device)
    end = None
    if PROFILE.which is not None:
        end = PROFILE.bracket(("conv_h2d_fwd" if s.Cout > 128 else "conv_h2d_fwd_narrow") if x_planes else "conv_x3_128x256" if s.Cout > 128 else "conv_fwd_narrow",
                              2.0 * s.N * s.Ho * s.Wo * s.Cout * s.R * s.S * s.Cin, _shape_tag(s))
    if bn is not None:
        count = s.N * s.Ho * s.Wo
        fin, mean, invstd = _bn_fin(bn, count, x.device)
        call("rcf_conv2d_fwd_bnstats_f32", _p(x), _p(weight_rsck(w)), _p(out), byref(s), None, byref(fin), _p(ws), need, _stream())
        sums = (mean, invstd, count)
    else:
        call("rcf_conv2d_fwd_stats_f32", _p(x), _p(weight_rsck(w)), _p(out), byref(s), _p(sums), _p(ws), need, _stream())
    if end is not None:
        end.record()
        if not x_planes:
            _relabel_conv(s, None, 0, "conv_x3_128x256" if s.Cout > 128 else "conv_fwd_narrow", "conv_h2p_fwd")
    return out, sums


def conv2d_dgrad(dy, w, xshape, stride=1, pad=0, dil=1, out=None, beta=0, region=None, amax=None, w_pairs_t=None,
                 dy_planes=False, amax_y=None, bn_bwd=None, addend=None):
    """region = (y0, x0, h, w) in INPUT coordinates: only those pixels of dx are written.  amax = (amax_dy, amax_w);
    w_pairs_t = weight_pairs_t(w, amax_w), prepared once per weight update.  dy_planes: `dy` holds fp16 pair planes
    (RCF_CONV_DY_PLANES: bn_bwd_apply's dx); amax_y: new_amax() slot for the range of dx (after the accumulation).
    bn_bwd = (x_bn, relu_mask, mean, invstd) of the batch norm + ReLU whose OUTPUT is this conv's input, given when this call is the
    last writer of dx: returns (dx, sums2) with the norm's backward sums from the kernel's epilogue (rcf_conv2d_dgrad_bnsums_f32),
    or (dx, None) when the launch this shape takes has no such epilogue -- the caller then runs bn_bwd_reduce.
    addend = (t, relu_mask): dx = data gradient + (relu_mask ? t : 0) (rcf_conv2d_dgrad_add_f32; needs dgrad_takes_addend(...), beta 0)"""
    _need_cuda(dy, w)
    if out is None:
        out = torch.empty(tuple(xshape), dtype=torch.float32, device=dy.device)
    if amax is None or amax[0] is None or amax[1] is None or "d" not in SCHED.h2_kinds:
        w_pairs_t = None
    s = _conv_shape(xshape, pitch_of(out), w, stride, pad, dil, pitch_of(dy),
                    amax=None if amax is None else (None, amax[1], amax[0]), w_pairs2_t=w_pairs_t,
                    flags=_lib.CONV_DY_PLANES if dy_planes else 0, amax_y=amax_y, nt_cols=xshape[3])
    assert tuple(dy.shape) == (s.N, s.Ho, s.Wo, s.Cout)
    fuse = (bn_bwd is not None and region is None and out.is_contiguous() and bn_bwd[0].dtype == torch.float32 and
            tuple(bn_bwd[0].shape) == tuple(out.shape) and _lib.load().rcf_conv2d_dgrad_bnsums_ok(byref(s)) == 1)
    if addend is not None:
        assert beta == 0 and region is None and tuple(addend[0].shape) == tuple(out.shape) and addend[0].dtype == torch.float32
        if _lib.load().rcf_conv2d_dgrad_bnsums_ok(byref(s)) != 1:
            raise _lib.RcfHipError("this data gradient cannot take a masked addend (ask ops.dgrad_takes_addend first)")
    need = 0 if w_pairs_t is not None else _lib.load().rcf_conv2d_dgrad_workspace_bytes(byref(s))
    if fuse:
        need = _lib.load().rcf_conv2d_dgrad_bnsums_workspace_bytes(byref(s))
    ws = workspace(need, dy.device) if need else None
    end = None
    if PROFILE.which is not None:
        end = PROFILE.bracket(("conv_h2d_dgrad" if (s.Cin > 128 and stride == 1) else "conv_h2d_dgrad_narrow") if dy_planes else
                              "conv_dgrad_wide" if (s.Cin > 128 and stride == 1) else "conv_dgrad_other",
                              2.0 * s.N * _region_pixels(region, s.H, s.W) * s.Cin * s.R * s.S * s.Cout, _shape_tag(s, region))
    sums2 = None
    if fuse or addend is not None:
        bn = None
        if fuse:
            xb, mask, mean, invstd = bn_bwd
            sums2 = torch.empty(2 * s.Cin, dtype=torch.float64, device=dy.device)
            bn = byref(_lib.BnBwdIn(xb.data_ptr(), pitch_of(xb), mask.data_ptr(), mean.data_ptr(), invstd.data_ptr()))
        ad, am = addend if addend is not None else (None, None)
        call("rcf_conv2d_dgrad_add_f32", _p(dy), _p(weight_rsck(w)), _p(out), byref(s), beta, _p(ad), pitch_of(ad) if ad is not None else 0,
             _p(am), bn, _p(sums2), _p(ws) if fuse else None, need if fuse else 0, _stream())
    else:
        call("rcf_conv2d_dgrad_region_f32", _p(dy), _p(weight_rsck(w)), _p(out), byref(s), _region(region), beta, _p(ws),
             need, _stream())
    if end is not None:
        end.record()
        if not dy_planes and not fuse and addend is None:
            _relabel_conv(s, _region(region), 1, "conv_dgrad_wide" if (s.Cin > 128 and stride == 1) else "conv_dgrad_other", "conv_h2p_dgrad")
    return (out, sums2) if bn_bwd is not None else out


def dgrad_takes_addend(w, xshape, stride, pad, dil, dy_pitch, amax, w_pairs_t, dy_planes):
    """can conv2d_dgrad(addend=...) run for this launch?  (rcf_conv2d_dgrad_bnsums_ok: fp16-pair kernels with prepared weights, Cin
    a whole column tile)"""
    if amax is None or amax[0] is None or amax[1] is None or w_pairs_t is None or "d" not in SCHED.h2_kinds:
        return False
    s = _conv_shape(xshape, xshape[3], w, stride, pad, dil, dy_pitch, amax=(None, amax[1], amax[0]), w_pairs2_t=w_pairs_t,
                    flags=_lib.CONV_DY_PLANES if dy_planes else 0)
    return _lib.load().rcf_conv2d_dgrad_bnsums_ok(byref(s)) == 1


def relu_mask_copy(dy, relu_mask, out=None, beta=0):
    """out (+)= relu_mask ? dy : 0 (rcf_relu_mask_copy_mp)"""
    _need_cuda(dy)
    if out is None:
        out = torch.empty(tuple(dy.shape), dtype=dy.dtype, device=dy.device)
    call("rcf_relu_mask_copy_mp", _p(dy), _dt(dy), pitch_of(dy), _p(relu_mask), _p(out), pitch_of(out), _rows(dy), dy.shape[3], int(beta), _stream())
    return out


def conv2d_wgrad(x, dy, w_like, dw, stride=1, pad=0, dil=1, beta=1, region=None, amax=None, small_tile=False, planes=False):
    """dw (same memory layout as the weight) (+)= wgrad.  region = (y0, x0, h, w) in OUTPUT coordinates: only those
    pixels of dy (and their input patches) contribute.  small_tile: RCF_CONV_WGRAD_TILE_128 (a launch that shares the chip
    with an HBM-bound kernel on another stream).  planes: x AND dy hold fp16 pair planes (RCF_CONV_X_PLANES | RCF_CONV_DY_PLANES:
    bn_apply's `planes`, bn_bwd_apply's dx)."""
    _need_cuda(x, dy, dw)
    s = _conv_shape(x.shape, pitch_of(x), w_like, stride, pad, dil, pitch_of(dy),
                    amax=None if amax is None else (amax[0], None, amax[1]),      # amax = (amax_x, amax_dy)
                    flags=(_lib.CONV_WGRAD_TILE_128 if small_tile else 0) |
                          ((_lib.CONV_X_PLANES | _lib.CONV_DY_PLANES) if planes else 0))
    reg = _region(region)
    need = _lib.load().rcf_conv2d_wgrad_region_workspace_bytes(byref(s), reg)
    ws = workspace(need, x.device) if need else None
    end = None
    if PROFILE.which is not None:
        ktot = s.R * s.S * s.Cin                                   # plan_wgrad (csrc/igemm_conv.hip): the 128 x 256 fp16-pair tile
        wide = amax is not None and s.Cin % 64 == 0 and ktot >= 256 and s.Cout >= 64
        end = PROFILE.bracket(("conv_wgrad_h2d" if ktot >= 256 else "conv_wgrad_h2d_narrow") if planes else ("conv_wgrad_h2t4" if wide else "conv_wgrad_other"),
                              2.0 * s.N * _region_pixels(region, s.Ho, s.Wo) * s.Cout * s.R * s.S * s.Cin, _shape_tag(s, region))
    call("rcf_conv2d_wgrad_region_f32", _p(x), _p(dy), _p(weight_rsck(dw)), byref(s), reg, beta, _p(ws), need, _stream())
    if end is not None:
        end.record()
    return dw


# ------------------------------------------------------------------------------- bf16-operand convs (csrc/igemm_bf16.hip)
def weight_bf16(w, transpose=False):
    """fp32 master weight (channels_last [Cout,Cin,R,S]) -> the bf16 operand of the bf16 conv kernels (uint8 buffer)"""
    _need_cuda(w)
    Cout, Cin, R, S = w.shape
    out = torch.empty(_lib.load().rcf_conv_weight_bf16_bytes(Cout, Cin, R, S, int(transpose)), dtype=torch.uint8,
                      device=w.device)
    call("rcf_conv_weight_bf16", _p(weight_rsck(w)), Cout, Cin, R, S, int(transpose), _p(out), CONV_FLAGS, _stream())
    return out


def conv2d_fwd_bf16(x, w, w_bf16=None, bias=None, stride=1, pad=0, dil=1, act=0, slope=0.0, out=None, beta=0, region=None,
                    out_dtype=None, stats=False, bn=None):
    """x: NHWC bf16; w: the fp32 master weight (shape only, unless w_bf16 is None); returns y (bf16 or fp32) and, with
    stats, the fp64 [2*Cout] batch-norm sums of the fp32 accumulators"""
    _need_cuda(x, w)
    assert x.dtype == half()
    out_dtype = out_dtype or half()
    if w_bf16 is None:
        w_bf16 = weight_bf16(w)
    s = _conv_shape(x.shape, pitch_of(x), w, stride, pad, dil, nt_cols=w.shape[0])
    if out is None:
        out = torch.empty((s.N, s.Ho, s.Wo, s.Cout), dtype=out_dtype, device=x.device)
    s.y_pitch = pitch_of(out)
    sums = ws = None
    need = 0
    if stats:
        sums = torch.empty(2 * s.Cout, dtype=torch.float64, device=x.device)
        need = _lib.load().rcf_conv2d_fwd_stats_bf16_workspace_bytes(byref(s))
        ws = workspace(need, x.device)
    end = None
    if PROFILE.which is not None:
        end = PROFILE.bracket("conv_bf16_fwd" if s.Cout > 128 else "conv_bf16_fwd_narrow",
                              2.0 * s.N * _region_pixels(region, s.Ho, s.Wo) * s.Cout * s.R * s.S * s.Cin, _shape_tag(s, region))
    if stats and bn is not None:                     # the statistics reduction also finalizes the batch norm
        count = s.N * s.Ho * s.Wo
        fin, mean, invstd = _bn_fin(bn, count, x.device)
        call("rcf_conv2d_fwd_bnstats_bf16", _p(x), _p(w_bf16), _p(out), _dt(out), byref(s), None, byref(fin), _p(ws), need,
             _stream())
        sums = (mean, invstd, count)
    else:
        call("rcf_conv2d_fwd_bf16", _p(x), _p(w_bf16), _p(bias), _p(out), _dt(out), byref(s), _region(region), act, slope, beta,
             _p(sums), _p(ws), need, _stream())
    if end is not None:
        end.record()
    return (out, sums) if stats else out


def conv2d_dgrad_bf16(dy, w, xshape, stride=1, pad=0, dil=1, out=None, beta=0, region=None, w_t_bf16=None):
    """dy: NHWC bf16, w: fp32 master weight -> dx bf16 (region in INPUT coordinates).  w_t_bf16 = weight_bf16(w, True),
    prepared once per weight update (otherwise the launch casts into its workspace)"""
    _need_cuda(dy, w)
    assert dy.dtype == half()
    if out is None:
        out = torch.empty(tuple(xshape), dtype=half(), device=dy.device)
    s = _conv_shape(xshape, pitch_of(out), w, stride, pad, dil, pitch_of(dy), w_pairs_t=w_t_bf16, nt_cols=xshape[3])
    assert tuple(dy.shape) == (s.N, s.Ho, s.Wo, s.Cout)
    need = 0 if w_t_bf16 is not None else _lib.load().rcf_conv2d_dgrad_bf16_workspace_bytes(byref(s))
    ws = workspace(need, dy.device) if need else None
    end = None
    if PROFILE.which is not None:
        end = PROFILE.bracket("conv_bf16_dgrad_wide" if (s.Cin > 128 and stride == 1) else "conv_bf16_dgrad_other",
                              2.0 * s.N * _region_pixels(region, s.H, s.W) * s.Cin * s.R * s.S * s.Cout, _shape_tag(s, region))
    call("rcf_conv2d_dgrad_bf16", _p(dy), _p(weight_rsck(w)), _p(out), byref(s), _region(region), beta, _p(ws), need,
         _stream())
    if end is not None:
        end.record()
    return out


def conv2d_wgrad_bf16(x, dy, w_like, dw, stride=1, pad=0, dil=1, beta=1, region=None):
    """dw (fp32, the weight's memory layout) (+)= wgrad of bf16 x / dy (region in OUTPUT coordinates)"""
    _need_cuda(x, dy, dw)
    assert x.dtype == half() and dy.dtype == half() and dw.dtype == torch.float32
    s = _conv_shape(x.shape, pitch_of(x), w_like, stride, pad, dil, pitch_of(dy))
    reg = _region(region)
    need = _lib.load().rcf_conv2d_wgrad_bf16_workspace_bytes(byref(s), reg)
    ws = workspace(need, x.device) if need else None
    end = None
    if PROFILE.which is not None:
        wide = region is None and s.Cout >= 64 and s.R * s.S * s.Cin >= 256                     # plan_wgrad (csrc/igemm_bf16.hip)
        end = PROFILE.bracket("conv_bf16_wgrad4" if wide else "conv_bf16_wgrad_other",
                              2.0 * s.N * _region_pixels(region, s.Ho, s.Wo) * s.Cout * s.R * s.S * s.Cin, _shape_tag(s, region))
    call("rcf_conv2d_wgrad_bf16", _p(x), _p(dy), _p(weight_rsck(dw)), byref(s), reg, beta, _p(ws), need, _stream())
    if end is not None:
        end.record()
    return dw


# ------------------------------------------------------- folded 1x1 conv + batch norm (csrc/foldbn.hip; include/rcf_hip.h)
def gram_bf16(x):
    """S = x^T x: [K,K,1,1] fp32 (the weight-gradient kernel with dy = x) of an NHWC bf16 activation [.., K]"""
    K = x.shape[3]
    S = torch.empty((K, K, 1, 1), dtype=torch.float32, device=x.device)
    conv2d_wgrad_bf16(x, x, S, S, 1, 0, 1, beta=0)
    return S


def _fold_fin(bn, count, device):
    C = bn.num_features
    mk = lambda: torch.empty(C, dtype=torch.float32, device=device)
    mean, invstd, scale, shift = mk(), mk(), mk(), mk()
    fin = _lib.FoldFinalize(float(count), bn.eps, bn.momentum, bn.weight.data_ptr(), bn.bias.data_ptr(), mean.data_ptr(),
                            invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), bn.running_mean.data_ptr(),
                            bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr())
    return fin, (mean, invstd, scale, shift)


def fold_fwd(S, A1, w, bn=None, count=None, rows=None):
    """The statistics of z = conv1x1(x, w) from the moments of x (A1: fp64 [K] column sums, S: Gram matrix, both over `rows` rows of
    this rank) and P: flat fp32 [N K + N] = the centred product W (S - A1 A1^T / rows) followed by the local means of z, kept for
    fold_bwd_prepare.  bn given (local statistics, count == rows): returns (P, (mean, invstd, scale, shift)), the norm finalized in
    the same launch; otherwise (P, sums fp64 [2N] = sum z | sum z^2) for fold_finalize after the all-reduce."""
    N, K = w.shape[0], w.shape[1]
    rows = float(count if rows is None else rows)
    P = torch.empty(N * K + N, dtype=torch.float32, device=w.device)
    need = _lib.load().rcf_fold_fwd_scratch_bytes(N, K)
    ws = workspace(need, w.device)
    if bn is not None:
        fin, outs = _fold_fin(bn, count, w.device)
        call("rcf_fold_fwd_f32", _p(S), _p(A1), _p(weight_rsck(w)), _p(P), None, byref(fin), _p(ws), need, rows, N, K, _stream())
        return P, outs
    sums = torch.empty(2 * N, dtype=torch.float64, device=w.device)
    call("rcf_fold_fwd_f32", _p(S), _p(A1), _p(weight_rsck(w)), _p(P), _p(sums), None, _p(ws), need, rows, N, K, _stream())
    return P, sums


def fold_finalize(sums, count, bn):
    """(mean, invstd, scale, shift) + the running statistics and num_batches_tracked of `bn` (a training-mode BatchNorm2d)"""
    fin, outs = _fold_fin(bn, count, sums.device)
    call("rcf_fold_finalize_f32", _p(sums), bn.num_features, byref(fin), _stream())
    return outs


def conv2d_fwd_affine_bf16(x, w, w_bf16, scale, shift, residual=None, relu=True, stride=1, pad=0, dil=1, want_bits=False):
    """y = [relu](conv(x, w) * scale[c] + shift[c] [+ residual]) in the conv's own epilogue (bf16 in, bf16 out).
    want_bits (with relu): also the ReLU's sign bits in tile order, (y, bits) -- for conv2d_dgrad_masked_bf16(mask_bits=)"""
    _need_cuda(x, w)
    assert x.dtype == half() and (residual is None or residual.dtype == half())
    s = _conv_shape(x.shape, pitch_of(x), w, stride, pad, dil, nt_cols=w.shape[0])
    out = torch.empty((s.N, s.Ho, s.Wo, s.Cout), dtype=half(), device=x.device)
    s.y_pitch = pitch_of(out)
    if residual is not None:
        assert tuple(residual.shape) == tuple(out.shape)
    end = None
    if PROFILE.which is not None:
        end = PROFILE.bracket("conv_bf16_fwd" if s.Cout > 128 else "conv_bf16_fwd_narrow",
                              2.0 * s.N * s.Ho * s.Wo * s.Cout * s.R * s.S * s.Cin, _shape_tag(s) + " +bn")
    bits = None
    if want_bits and relu:
        bits = torch.empty(_lib.load().rcf_conv_relu_bits_bytes(s.N * s.Ho * s.Wo, s.Cout), dtype=torch.uint8, device=x.device)
    call("rcf_conv2d_fwd_affine_bf16", _p(x), _p(w_bf16), _p(scale), _p(shift), _p(residual),
         pitch_of(residual) if residual is not None else 0, int(relu), _p(out), _p(bits), byref(s), _stream())
    if end is not None:
        end.record()
    return (out, bits) if want_bits else out


def relu_mask_colsum_ok(C):
    """does rcf_relu_mask_colsum_bf16 -- the fold's backward when no data gradient delivers the masked gradient -- take this
    width?  The library's own predicate (its workspace query returns 0 for a width it refuses: 64 ... 2048 in powers of two
    pass), so that layers.fold_ok cannot accept a conv whose backward would then fail."""
    return _lib.load().rcf_relu_mask_colsum_bf16_workspace_bytes(64, int(C)) > 0


def relu_mask_colsum(dy, y, out=None):
    """(g = y > 0 ? dy : 0, fp64 [2C] whose first half holds the column sums of g); out may be dy (in place)"""
    _need_cuda(dy, y)
    assert dy.dtype == half() and y.dtype == half() and tuple(dy.shape) == tuple(y.shape)
    if out is None:
        out = torch.empty(tuple(dy.shape), dtype=dy.dtype, device=dy.device)
    rows, C = _rows(dy), dy.shape[3]
    cs = torch.empty(2 * C, dtype=torch.float64, device=dy.device)
    need = _lib.load().rcf_relu_mask_colsum_bf16_workspace_bytes(rows, C)
    ws = workspace(need, dy.device)
    call("rcf_relu_mask_colsum_bf16", _p(dy), pitch_of(dy), _p(y), pitch_of(y), _p(out), pitch_of(out), rows, C, _p(cs), _p(ws),
         need, _stream())
    return out, cs


def conv2d_dgrad_masked_bf16(dy, w, xshape, w_t_bf16, mask_src, out, beta=0, stride=1, pad=0, dil=1, colsums=True, mask_bits=None):
    """dx = mask_src > 0 ? conv_transpose(dy, w) (+ dx) : 0 in the data gradient's epilogue, with the column sums of what it
    writes (fp64 [2 Cin], first half): the LAST writer of a ReLU output's gradient applies that ReLU's mask"""
    _need_cuda(dy, w)
    assert dy.dtype == half() and (mask_bits is not None or (mask_src.dtype == half() and tuple(mask_src.shape) == tuple(xshape)))
    if mask_bits is not None:
        mask_src = None                       # the sign bits the forward tile wrote (1/16 of the bytes) instead of the tensor
        assert mask_bits.numel() == _lib.load().rcf_conv_relu_bits_bytes(xshape[0] * xshape[1] * xshape[2], xshape[3])
    s = _conv_shape(xshape, pitch_of(out), w, stride, pad, dil, pitch_of(dy), nt_cols=xshape[3])
    cs = ws = None
    need = 0
    if colsums:
        cs = torch.empty(2 * s.Cin, dtype=torch.float64, device=dy.device)
        need = _lib.load().rcf_conv2d_dgrad_masked_bf16_workspace_bytes(byref(s))
        ws = workspace(need, dy.device)
    end = None
    if PROFILE.which is not None:
        end = PROFILE.bracket("conv_bf16_dgrad_wide" if s.Cin > 128 else "conv_bf16_dgrad_other",
                              2.0 * s.N * s.H * s.W * s.Cin * s.R * s.S * s.Cout, _shape_tag(s) + " +mask")
    call("rcf_conv2d_dgrad_masked_bf16", _p(dy), _p(w_t_bf16), _p(out), byref(s), int(beta), _p(mask_src),
         pitch_of(mask_src) if mask_src is not None else 0, _p(mask_bits), _p(cs), _p(ws), need, _stream())
    if end is not None:
        end.record()
    return out, cs


def fold_bwd_sums(G, w, colsums, mean, invstd):
    """fp64 [2N] = sum g | sum g zhat of the folded norm, from G = g^T x and the column sums of g"""
    N, K = w.shape[0], w.shape[1]
    sums2 = torch.empty(2 * N, dtype=torch.float64, device=w.device)
    call("rcf_fold_bwd_sums_f32", _p(G), _p(weight_rsck(w)), _p(colsums), _p(mean), _p(invstd), _p(sums2), N, K, _stream())
    return sums2


def fold_wg(w, scale):
    """(scale[c] W)^T as the bf16 operand of the folded data gradient g Wg^T (uint8 buffer, weight_bf16(.., True) layout)"""
    N, K = w.shape[0], w.shape[1]
    wg_t = torch.empty(_lib.load().rcf_conv_weight_bf16_bytes(N, K, 1, 1, 1), dtype=torch.uint8, device=w.device)
    call("rcf_fold_wg_bf16", _p(weight_rsck(w)), _p(scale), _p(wg_t), N, K, _stream())
    return wg_t


def fold_bwd_prepare(G, P, A1, w, sums2, sums2_local, count, mean, invstd, gamma, dW, dgamma, dbeta):
    """dW / dgamma / dbeta accumulate; returns (negT, c0): the K -> K weight operand and bias of the second half of the folded
    data gradient, dx += x (-T) + c0"""
    N, K = w.shape[0], w.shape[1]
    lib = _lib.load()
    negT = torch.empty(lib.rcf_conv_weight_bf16_bytes(K, K, 1, 1, 0), dtype=torch.uint8, device=w.device)
    c0 = torch.empty(K, dtype=torch.float32, device=w.device)
    need = lib.rcf_fold_bwd_scratch_bytes(N, K)
    ws = workspace(need, w.device)
    call("rcf_fold_bwd_prepare_f32", _p(G), _p(P), _p(A1), _p(weight_rsck(w)), _p(sums2), _p(sums2_local), float(count), _p(mean),
         _p(invstd), _p(gamma), _p(weight_rsck(dW)) if dW is not None else None, _p(dgamma), _p(dbeta), _p(negT), _p(c0),
         _p(ws), need, N, K, _stream())
    return negT, c0


def cast(x, dtype, out=None):
    """NHWC activation (possibly a channel slice) -> the same values in `dtype` (fp32 <-> bf16)"""
    _need_cuda(x)
    if out is None:
        out = torch.empty(tuple(x.shape), dtype=dtype, device=x.device)
    call("rcf_copy2d_mp", _p(x), _dt(x), pitch_of(x), _p(out), _dt(out), pitch_of(out), _rows(x), x.shape[3], 0, _stream())
    return out


def _rows(x):
    return x.shape[0] * x.shape[1] * x.shape[2]


def bn_stats(x):
    """fp64 [2C]: per-channel sum | sum of squares over all pixels."""
    _need_cuda(x)
    rows, C = _rows(x), x.shape[3]
    sums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
    need = _lib.load().rcf_bn_stats_workspace_bytes(rows, C)
    ws = workspace(need, x.device)
    call("rcf_bn_stats_mp", _p(x), _dt(x), rows, C, pitch_of(x), _p(sums), _p(ws), need, _stream())
    return sums


# RCF_BN_* bits OR-ed into every streaming batch-norm launch (tests: the row-order switches); Python-side state, sent per call
BN_FLAGS = 0


def bn_finalize(sums, count, eps, momentum, running_mean=None, running_var=None):
    C = sums.numel() // 2
    mean = torch.empty(C, dtype=torch.float32, device=sums.device)
    invstd = torch.empty(C, dtype=torch.float32, device=sums.device)
    call("rcf_bn_finalize_f32", _p(sums), float(count), C, eps, momentum, _p(mean), _p(invstd), _p(running_mean),
         _p(running_var), _stream())
    return mean, invstd


def bn_invstd_from_var(var, eps):
    out = torch.empty_like(var)
    call("rcf_bn_invstd_from_var_f32", _p(var), var.numel(), eps, _p(out), _stream())
    return out


def bn_apply(x, mean, invstd, gamma, beta, relu, residual=None, chan_scale=None, out=None, relu_mask=None,
             amax_out=None, out_dtype=None, planes=None, planes_only=False, amax_x=None, amax_res=None, res_norm=None):
    """relu_mask: uint8 [rows * C/4] to receive the sign bits of the pre-clamp output (for the backward pass).
    res_norm: (mean, invstd, gamma, beta) of a plain batch norm the RAW residual is normalised by on the fly
    (rcf_bn_apply_res_mp: a stage's downsample branch, whose own apply pass then never runs); amax_res = the raw residual's range.
    out_dtype: storage type of y (default: x's); an fp32 x may be normalised into bf16 activations.
    planes: an fp32-typed buffer of y's shape that receives y as fp16 pair planes (RCF_CONV_X_PLANES of the consuming convs),
    scaled by a bound derived from amax_x (the range of x) and amax_res (of the residual); the bound goes to amax_out.
    planes_only: y itself is not written (returns None)"""
    _need_cuda(x)
    if out is None and not (planes is not None and planes_only):
        out = torch.empty(tuple(x.shape), dtype=out_dtype or x.dtype, device=x.device)
    rows, C = _rows(x), x.shape[3]
    ydt = _same_dt(out, residual) if out is not None else _dt(x)
    flags = BN_FLAGS | (_lib.BN_Y_PLANES_ONLY if (planes is not None and planes_only) else 0)
    rn = None
    if res_norm is not None:
        rn = _lib.BnResNorm(*[t.data_ptr() for t in res_norm])
    call("rcf_bn_apply_res_mp", _p(x), _dt(x), pitch_of(x), _p(residual), pitch_of(residual) if residual is not None else 0,
         byref(rn) if rn is not None else None,
         _p(out), ydt, pitch_of(out) if out is not None else C, rows, C, _p(mean), _p(invstd), _p(gamma), _p(beta), int(relu),
         _p(chan_scale), x.shape[1] * x.shape[2], _p(relu_mask), _p(amax_out), _p(planes), _p(amax_x), _p(amax_res), flags,
         _stream())
    return out


def bn_bwd_reduce(dy, x, y, mean, invstd, relu, chan_scale=None, relu_mask=None):
    rows, C = _rows(x), x.shape[3]
    sums2 = torch.empty(2 * C, dtype=torch.float64, device=x.device)
    need = _lib.load().rcf_bn_stats_workspace_bytes(rows, C)
    ws = workspace(need, x.device)
    call("rcf_bn_bwd_reduce_mp", _p(dy), _same_dt(dy, y), pitch_of(dy), _p(x), _dt(x), pitch_of(x), _p(y),
         pitch_of(y) if y is not None else 0, rows, C, _p(mean), _p(invstd), int(relu), _p(relu_mask), _p(chan_scale), x.shape[1] * x.shape[2], _p(sums2),
         _p(ws), need, BN_FLAGS, _stream())
    return sums2


def bn_bwd_apply(dy, x, y, mean, invstd, gamma, relu, sums2, count, dgamma, dbeta, dx=None, dres=None, res_beta=0,
                 chan_scale=None, sums2_local=None, relu_mask=None, amax_out=None, dx_planes=False, amax_x=None, amax_dy=None):
    """dx_planes: dx (an fp32-typed buffer of x's shape) receives fp16 pair planes (RCF_CONV_DY_PLANES of the conv's data and
    weight gradient) scaled by a bound from amax_x / amax_dy (the ranges of x and dy); the bound goes to amax_out"""
    rows, C = _rows(x), x.shape[3]
    if dx is None:
        dx = torch.empty(tuple(x.shape), dtype=x.dtype, device=x.device)
    call("rcf_bn_bwd_apply_mp", _p(dy), _same_dt(dy, y, dres), pitch_of(dy), _p(x), _same_dt(x, dx), pitch_of(x), _p(y),
         pitch_of(y) if y is not None else 0, _p(dx), pitch_of(dx), _p(dres), pitch_of(dres) if dres is not None else 0, res_beta, rows, C, _p(mean),
         _p(invstd), _p(gamma), int(relu), _p(relu_mask), _p(chan_scale), x.shape[1] * x.shape[2], _p(sums2),
         _p(sums2_local),
         float(count), _p(dgamma), _p(dbeta), _p(amax_out), _p(amax_x), _p(amax_dy),
         BN_FLAGS | (_lib.BN_DX_PLANES if dx_planes else 0), _stream())
    return dx


def bn_bwd_reduce2(dy, x, x2, mean, invstd, mean2, invstd2, relu_mask):
    """[sum g | sum g xhat | sum g | sum g xhat2] (fp64 [4C]) of g = dy under relu_mask against two norm inputs in ONE pass
    (rcf_bn_bwd_reduce2_mp: a stage's first join + its downsample norm)"""
    rows, C = _rows(x), x.shape[3]
    sums4 = torch.empty(4 * C, dtype=torch.float64, device=x.device)
    need = 2 * _lib.load().rcf_bn_stats_workspace_bytes(rows, C)
    ws = workspace(need, x.device)
    call("rcf_bn_bwd_reduce2_mp", _p(dy), _same_dt(dy), pitch_of(dy), _p(x), _same_dt(x, x2), pitch_of(x), _p(x2), pitch_of(x2), rows, C,
         _p(mean), _p(invstd), _p(mean2), _p(invstd2), _p(relu_mask), _p(sums4), _p(ws), need, BN_FLAGS, _stream())
    return sums4


def bn_bwd_apply2(dy, x, mean, invstd, gamma, relu_mask, sums2, count, dgamma, dbeta, dx, x2, mean2, invstd2, gamma2, sums2_2,
                  dgamma2, dbeta2, dx2, sums2_local=None, sums2_2_local=None, amax_out=None, amax_out2=None, dx_planes=False,
                  amax_x=None, amax_x2=None, amax_dy=None):
    """both norms' input gradients (and parameter gradients) in ONE pass over dy and the sign bits (rcf_bn_bwd_apply2_mp)"""
    rows, C = _rows(x), x.shape[3]
    sec = _lib.BnBwdSecond(x2.data_ptr(), pitch_of(x2), dx2.data_ptr(), pitch_of(dx2), mean2.data_ptr(), invstd2.data_ptr(),
                           gamma2.data_ptr(), sums2_2.data_ptr(), sums2_2_local.data_ptr() if sums2_2_local is not None else None,
                           dgamma2.data_ptr() if dgamma2 is not None else None, dbeta2.data_ptr() if dbeta2 is not None else None,
                           amax_out2.data_ptr() if amax_out2 is not None else None, amax_x2.data_ptr() if amax_x2 is not None else None)
    call("rcf_bn_bwd_apply2_mp", _p(dy), _same_dt(dy), pitch_of(dy), _p(x), _same_dt(x, dx, x2, dx2), pitch_of(x), _p(dx), pitch_of(dx),
         rows, C, _p(mean), _p(invstd), _p(gamma), _p(relu_mask), _p(sums2), _p(sums2_local), float(count), _p(dgamma), _p(dbeta),
         _p(amax_out), _p(amax_x), _p(amax_dy), byref(sec), BN_FLAGS | (_lib.BN_DX_PLANES if dx_planes else 0), _stream())
    return dx, dx2


def maxpool_fwd(x):
    N, H, W, C = x.shape
    assert x.is_contiguous()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    am = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    call("rcf_maxpool3x3s2_fwd_mp", _p(x), _p(y), _dt(x), _p(am), N, H, W, C, Ho, Wo, _stream())
    return y, am


def maxpool_bwd(dy, am, xshape):
    N, H, W, C = xshape
    assert dy.is_contiguous()
    dx = torch.empty(tuple(xshape), dtype=dy.dtype, device=dy.device)
    call("rcf_maxpool3x3s2_bwd_mp", _p(dy), _p(am), _p(dx), _dt(dy), N, H, W, C, dy.shape[1], dy.shape[2], _stream())
    return dx


def resize_nhwc_fwd(x, size, align_corners=False, out=None, frame=0):
    """frame > 0: only the output pixels within `frame` of the border are written"""
    N, Hi, Wi, C = x.shape
    Ho, Wo = size
    if out is None:
        out = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    call("rcf_resize_bilinear_nhwc_fwd_mp", _p(x), pitch_of(x), _p(out), pitch_of(out), _same_dt(x, out), N, Hi, Wi, Ho, Wo,
         C, int(align_corners), int(frame), _stream())
    return out


def resize_nhwc_bwd(dy, in_size, align_corners=False, out=None, beta=0, frame=0):
    """frame > 0: dy counts as zero (and is not read) outside the border frame of that thickness"""
    N, Ho, Wo, C = dy.shape
    Hi, Wi = in_size
    if out is None:
        out = torch.empty((N, Hi, Wi, C), dtype=dy.dtype, device=dy.device)
    call("rcf_resize_bilinear_nhwc_bwd_mp", _p(dy), pitch_of(dy), _p(out), pitch_of(out), _same_dt(dy, out), beta, N, Hi, Wi,
         Ho, Wo, C, int(align_corners), int(frame), _stream())
    return out


def resize_nchw(x, size, align_corners=False):
    assert x.is_contiguous()
    Hi, Wi = x.shape[-2:]
    out = torch.empty(tuple(x.shape[:-2]) + tuple(size), dtype=torch.float32, device=x.device)
    planes = x.numel() // (Hi * Wi)
    call("rcf_resize_bilinear_nchw_f32", _p(x), _p(out), planes, Hi, Wi, size[0], size[1], int(align_corners), _stream())
    return out


def nchw_to_nhwc(x, cpad=None):
    assert x.is_contiguous()
    N, C, H, W = x.shape
    cpad = cpad or (C + 3) // 4 * 4
    out = torch.empty((N, H, W, cpad), dtype=torch.float32, device=x.device)
    call("rcf_nchw_to_nhwc_f32", _p(x), _p(out), N, C, H, W, cpad, _stream())
    return out


def nhwc_to_nchw(x, C=None):
    if x.dtype in H16:
        x = cast(x, torch.float32)
    N, H, W, Cx = x.shape
    C = C or Cx
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
    call("rcf_nhwc_to_nchw_f32", _p(x), pitch_of(x), _p(out), N, C, H, W, _stream())
    return out


def copy2d(src, spitch, dst, dpitch, rows, C, beta=0):
    call("rcf_copy2d_mp", _p(src), _dt(src), spitch, _p(dst), _dt(dst), dpitch, rows, C, beta, _stream())


def copy2d_batched(src, spitch, sstrides, dst, dpitch, dstrides, rows, C, batch, beta=0):
    """batch[0] x batch[1] copies in one launch; src / dst are the base tensors of copy (0, 0) (their data pointers may
    carry an offset), strides in elements (may be negative)"""
    call("rcf_copy2d_batched_mp", _p(src), int(spitch), int(sstrides[0]), int(sstrides[1]), _p(dst), int(dpitch),
         int(dstrides[0]), int(dstrides[1]), _same_dt(src, dst), int(rows), int(C), int(beta), int(batch[0]), int(batch[1]),
         _stream())


def split_rect(x, rect, want_inside=True, want_outside=True):
    """dense NHWC x -> (x on the rectangle (y0,x0,h,w) else 0, x off the rectangle else 0)"""
    _need_cuda(x)
    assert x.is_contiguous()
    N, H, W, C = x.shape
    ins = torch.empty_like(x) if want_inside else None
    outs = torch.empty_like(x) if want_outside else None
    call("rcf_split_rect_mp", _p(x), _p(ins), _p(outs), _dt(x), N, H, W, C, *[int(v) for v in rect], _stream())
    return ins, outs


def colsum(x, out, beta=1):
    rows, C = _rows(x), x.shape[3]
    need = _lib.load().rcf_bn_stats_workspace_bytes(rows, C)
    ws = workspace(need, x.device)
    call("rcf_colsum_mp", _p(x), _dt(x), rows, C, pitch_of(x), _p(out), beta, _p(ws), need, _stream())
    return out


# ------------------------------------------------------------------------------- warp family (NCHW planar)
PAD = {"border": 0, "zeros": 1, "border_per_pixel": 0 | _lib.WARP_PER_PIXEL}      # the last: RCF_WARP_PER_PIXEL (tests)


def flow_warp(x, flow12, pad="border"):
    _need_cuda(x, flow12)
    x, flow12 = x.contiguous(), flow12.contiguous()
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    call("rcf_flow_warp_f32", _p(x), _p(flow12), _p(out), B, C, H, W, PAD[pad], _stream())
    return out


def flow_warp_bwd(x, flow12, dout, pad="border", need_dx=True, need_dflow=True):
    x, flow12, dout = x.contiguous(), flow12.contiguous(), dout.contiguous()
    B, C, H, W = x.shape
    dx = torch.zeros_like(x) if need_dx else None
    dflow = torch.empty_like(flow12) if need_dflow else None
    call("rcf_flow_warp_bwd_f32", _p(x), _p(flow12), _p(dout), _p(dx), _p(dflow), B, C, H, W, PAD[pad], _stream())
    return dx, dflow


def occu_mask_backward(flow21, th=0.2):
    flow21 = flow21.contiguous()
    B, _, H, W = flow21.shape
    occ = torch.empty((B, 1, H, W), dtype=torch.float32, device=flow21.device)
    scratch = torch.empty((B, H, W), dtype=torch.float32, device=flow21.device)
    call("rcf_occu_mask_backward_f32", _p(flow21), _p(occ), th, _p(scratch), B, H, W, _stream())
    return occ


def occu_mask_bidirection(flow12, flow21, scale=0.01, bias=0.5):
    flow12, flow21 = flow12.contiguous(), flow21.contiguous()
    B, _, H, W = flow12.shape
    occ = torch.empty((B, 1, H, W), dtype=torch.float32, device=flow12.device)
    call("rcf_occu_mask_bidirection_f32", _p(flow12), _p(flow21), _p(occ), scale, bias, B, H, W, _stream())
    return occ


def warp_l1_residual(im1, im2, flow, occ=None, pad="border"):
    """fused |im1 - warp(im2, flow)| * occ: returns fp64 [2] = (sum of masked residual, sum of occ)."""
    im1, im2, flow = im1.contiguous(), im2.contiguous(), flow.contiguous()
    B, C, H, W = im1.shape
    out = torch.empty(2, dtype=torch.float64, device=im1.device)
    call("rcf_warp_l1_residual_f32", _p(im1), _p(im2), _p(flow), _p(occ.contiguous() if occ is not None else None),
         _p(out), B, C, H, W, PAD[pad], _stream())
    return out


def photometric_loss(im, recon, occ, w_l1=0.15, w_ssim=0.85):
    im, recon, occ = im.contiguous(), recon.contiguous(), occ.contiguous()
    B, C, H, W = im.shape
    out = torch.empty(1, dtype=torch.float32, device=im.device)
    scratch = torch.empty(4, dtype=torch.float64, device=im.device)
    call("rcf_photometric_loss_f32", _p(im), _p(recon), _p(occ), w_l1, w_ssim, _p(out), _p(scratch), B, C, H, W, _stream())
    return out[0]


# ------------------------------------------------------------------------------- optimiser
def adam_step(param, grad, exp_avg, exp_avg_sq, lr, step, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
              grad_scale=1.0):
    weights_changed()
    call("rcf_adam_step_f32", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, betas[0], betas[1],
         eps, weight_decay, step, grad_scale, _stream())


def ema_update(dest, src, m, invalidate=True):
    """dest = dest m + src (1 - m).  invalidate=False: the caller drops the cached operands of the module it wrote itself
    (momentum_update_param_and_buffer: only the teacher's) instead of every cached operand of the process"""
    if invalidate:
        weights_changed()
    call("rcf_ema_update_f32", _p(dest), _p(src), dest.numel(), m, _stream())


def ema_update_multi(table, count, m):
    """every entry of a state dict in one launch: `table` int64 [count, 4] on the device = rcf_ema_chunk {dst, src, n, kind}
    (model._EmaPlan).  Counters (kind 1) take float(1.0 - m) as torch's int64 * python-float arithmetic does."""
    import numpy as np
    call("rcf_ema_update_multi", _p(table), int(count), float(m), float(np.float32(1.0 - float(m))), _stream())


def dropout2d_scale(n, channels, p, seed, device):
    """nn.Dropout2d's draw as the [n, channels] fp32 scale the head's last batch-norm pass multiplies in: 0 with probability p,
    1 / (1 - p) otherwise (rcf_dropout2d_scale_f32: Philox keyed by `seed`, one launch, no torch kernels)"""
    out = torch.empty((n, channels), dtype=torch.float32, device=device)
    call("rcf_dropout2d_scale_f32", _p(out), n * channels, float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream())
    return out


def fill(t, v, weights=False):
    """t[:] = v.  weights=True when `t` may be (part of) a parameter: the cached weight operands are dropped.  The callers in
    this package fill gradient buffers only -- a bump there would throw away, at the top of every step, the operands the
    trainer prepared after the previous optimizer step (trainer.WeightPrep)."""
    if weights:
        weights_changed()
    call("rcf_fill_f32", _p(t), t.numel(), float(v), _stream())


# ---- DINO ViT / soft NCut helpers (csrc/vit.hip, rcf_gemm_nt_f32) ---------------------------------------------------
def _row_pitch(t):
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D view expected"
    return t.stride(0)


def gemm_nt(a, b, bias=None, out=None, act=0, beta=0, amax=None, b_pairs=None, amax_out=None):
    """out[M,N] (+)= a[M,K] @ b[N,K]^T + bias, act 0 none / 2 GELU.  a, b, out: row-major 2-D views (rows may be pitched).
    amax = (range of a, range of b) -> fp16-pair arithmetic; b_pairs: b split beforehand (weight_pairs_2d);
    amax_out: int32 [1] that receives the range of what is written"""
    _need_cuda(a, b)
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    if out is None:
        out = torch.empty((M, (N + 3) // 4 * 4), dtype=torch.float32, device=a.device)[:, :N]
    aa, ab = amax if amax is not None else (None, None)
    call("rcf_gemm_nt_f32", _p(a), _row_pitch(a), _p(b), _row_pitch(b), _p(bias), _p(out), _row_pitch(out), M, N, K,
         int(act), 0.0, int(beta), _p(aa), _p(ab), _p(b_pairs), _p(amax_out), _stream())
    return out


def weight_pairs_2d(w, amax_w):
    """weight_pairs for an nn.Linear weight [out, in] (contiguous): the B operand of gemm_nt, split once"""
    _need_cuda(w)
    N, K = w.shape
    assert w.is_contiguous() and K % 4 == 0
    planes = torch.empty(_lib.load().rcf_conv_weight_pairs_bytes(N, K, 1, 1), dtype=torch.uint8, device=w.device)
    call("rcf_conv_weight_pairs_f32", _p(w), N, K, 1, 1, _p(amax_w), _p(planes), 0, _stream())
    return planes


def gemm_nt_batched(a, lda, a_strides, b, ldb, b_strides, out, ldc, c_strides, batch, M, N, K, beta=0):
    """batch[0] x batch[1] products out_i = a_i[M,K] @ b_i[N,K]^T in one launch; a / b / out are base tensors (their data
    pointers may carry an offset), strides in elements"""
    call("rcf_gemm_nt_batched_f32", _p(a), int(lda), int(a_strides[0]), int(a_strides[1]), _p(b), int(ldb),
         int(b_strides[0]), int(b_strides[1]), _p(out), int(ldc), int(c_strides[0]), int(c_strides[1]), int(batch[0]),
         int(batch[1]), int(M), int(N), int(K), 0, 0.0, int(beta), _stream())
    return out


def attention(qkv, B, T, nh, scale, out=None, amax=None, amax_out=None):
    """fused softmax(scale q k^T) v for every image and head; qkv [B*T, 3*nh*64] -> [B*T, nh*64].
    amax: absmax(qkv) -> fp16-pair arithmetic (3 partial products instead of 6)"""
    _need_cuda(qkv)
    dim = qkv.shape[1] // 3
    if out is None:
        out = torch.empty((B * T, dim), dtype=torch.float32, device=qkv.device)
    call("rcf_attention_fwd_f32", _p(qkv), _row_pitch(qkv), _p(out), _row_pitch(out), B, T, nh, dim // nh, float(scale),
         _p(amax), _p(amax_out), _stream())
    return out


def layernorm(x, gamma, beta, eps, out=None, amax_out=None):
    _need_cuda(x)
    rows, C = x.shape
    if out is None:
        out = torch.empty((rows, C), dtype=torch.float32, device=x.device)
    call("rcf_layernorm_f32", _p(x), _row_pitch(x), _p(out), _row_pitch(out), rows, C, _p(gamma), _p(beta), float(eps),
         _p(amax_out), _stream())
    return out


def softmax_rows_(s, n, scale):
    """in place on a [rows, pitch] buffer: softmax(scale * s[:, :n]); columns >= n zeroed"""
    call("rcf_softmax_rows_f32", _p(s), _row_pitch(s), s.shape[0], int(n), float(scale), _stream())
    return s


def transpose2d(src, out_cols=None):
    """src [rows, cols] (pitched) -> dense [cols, out_cols >= rows], extra columns zero"""
    rows, cols = src.shape
    oc = rows if out_cols is None else int(out_cols)
    dst = torch.empty((cols, oc), dtype=torch.float32, device=src.device)
    call("rcf_transpose2d_f32", _p(src), _row_pitch(src), _p(dst), oc, rows, cols, _stream())
    return dst


def l2_normalize_rows(x):
    rows, C = x.shape
    out = torch.empty((rows, C), dtype=torch.float32, device=x.device)
    call("rcf_l2_normalize_rows_f32", _p(x), _row_pitch(x), _p(out), C, rows, C, _stream())
    return out
