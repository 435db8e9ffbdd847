"""The reference's data transform (dataset/transforms.py:884-924 `Transform`, selected by `get_transform` :926-929) over
the HIP gather kernels of csrc/datapipe.hip.

The reference transforms ONE sample at a time on a DataLoader worker (numpy / cv2) and collates afterwards; here the
loader hands over the DECODED arrays of a whole batch -- frames u8 [B,I,H,W,3], flows fp32 [B,H,W,2] per direction,
pseudo-label masks u8 [B,I,H,W] -- and one launch per output tensor produces the collated, normalised batch in HBM.
JPEG / PNG decoding stays with the loader (dataset/data.py:62-70: no decoder library on the device in this image, see the
note above `BatchUploader`); `.npy` flows are read straight into pinned staging buffers and the decoded batch is uploaded
on a copy stream (`load_flow_npy_into`, `BatchUploader`: dataset/data.py:122-128 and the DataLoader's collate / pin / copy).

`sample_params` consumes the random stream exactly as the reference's pipeline does (same generator, same calls, same
order: Resize.random_sample_ratio :130, RandomCrop.get_crop_bbox :447-448, RandomFlip :281, PhotoMetricDistortion
:599-679), so that under `np.random.seed(s)` a sample gets the decisions the reference would give it.

No CPU path: the tensors must live on the GPU, and the extension must load.
"""
import numpy as np
import torch

from . import _lib
from .ops import _p, _stream

MEAN = (0.485, 0.456, 0.406)                              # dataset/transforms.py:893
STD = (0.229, 0.224, 0.225)

# mirrors `struct rcf_aug_params` of include/rcf_hip.h (C layout, natural alignment)
PARAMS_DTYPE = np.dtype([("rw", "<i4"), ("rh", "<i4"), ("crop_x", "<i4"), ("crop_y", "<i4"), ("flip", "<i4"), ("ops", "<i4"),
                         ("beta", "<f4"), ("alpha_c", "<f4"), ("alpha_s", "<f4"), ("flow_sx", "<f4"), ("flow_sy", "<f4"),
                         ("hue_delta", "<f8")], align=True)
OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION, OP_HUE, OP_CONTRAST_LAST = 1, 2, 4, 8, 16


def rescale_size(w, h, scale):
    """mmcv.imrescale's size rule for a (long edge, short edge) bound: factor = min(long/max(h,w), short/min(h,w)),
    size = int(edge * factor + 0.5)"""
    max_long, max_short = max(scale), min(scale)
    f = min(max_long / max(h, w), max_short / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)


class Transform:
    """Same constructor as the reference's (`training`, `strong_aug`, `has_flow`, `has_attn`, `has_pl`, `scale_flow`).
    `has_attn` (AttnTransform, the AMD baseline's attention maps) is not on the RCF path and is refused."""

    img_scale = (9999, 400)                               # :897, :911
    crop_size = (384, 384)                                # :898
    brightness_delta, contrast_range, saturation_range, hue_delta = 32, (0.5, 1.5), (0.5, 1.5), 18     # :580-584

    def __init__(self, training, strong_aug=False, has_flow=True, has_attn=False, has_pl=False, scale_flow=False):
        if has_attn:
            raise NotImplementedError("has_attn belongs to the AMD baseline's loader, not to the RCF path")
        self.training, self.strong_aug, self.has_flow, self.has_pl, self.scale_flow = training, strong_aug, has_flow, has_pl, scale_flow
        self.ratio_range = (0.96, 1.0) if training else (0.98, 0.98)

    # -- host: the random decisions, in the reference's order -------------------------------------------------------------
    def sample_params(self, H, W, rng=np.random):
        """one sample's decisions as a PARAMS_DTYPE record; `rng` is numpy's global generator (what the reference
        draws from) or any RandomState"""
        p = np.zeros((), dtype=PARAMS_DTYPE)
        lo, hi = self.ratio_range
        ratio = rng.random_sample() * (hi - lo) + lo                                          # :130
        scale = int(self.img_scale[0] * ratio), int(self.img_scale[1] * ratio)                # :131
        rw, rh = rescale_size(W, H, scale)
        p["rw"], p["rh"] = rw, rh
        p["flow_sx"], p["flow_sy"] = 1.0, 1.0
        if self.training:
            ch, cw = self.crop_size
            if rh < ch or rw < cw:
                raise ValueError(f"resized frame {rh}x{rw} is smaller than the crop {ch}x{cw}: the reference rescales such "
                                 f"frames again (transforms.py:487-493); not supported on the device path")
            p["crop_y"] = rng.randint(0, max(rh - ch, 0) + 1)                                 # :447
            p["crop_x"] = rng.randint(0, max(rw - cw, 0) + 1)                                 # :448
            if self.strong_aug:
                p["flip"] = 1 if rng.rand() < 0.5 else 0                                      # :281
                ops = 0
                if rng.randint(2):                                                            # brightness :601
                    ops |= OP_BRIGHTNESS
                    p["beta"] = rng.uniform(-self.brightness_delta, self.brightness_delta)
                mode = rng.randint(2)                                                         # :664

                def contrast():
                    nonlocal ops
                    if rng.randint(2):                                                        # :610
                        ops |= OP_CONTRAST
                        p["alpha_c"] = rng.uniform(*self.contrast_range)
                if mode == 1:
                    contrast()
                if rng.randint(2):                                                            # saturation :626
                    ops |= OP_SATURATION
                    p["alpha_s"] = rng.uniform(*self.saturation_range)
                if rng.randint(2):                                                            # hue :643
                    ops |= OP_HUE
                    p["hue_delta"] = rng.uniform(-self.hue_delta, self.hue_delta)
                if mode == 0:
                    ops |= OP_CONTRAST_LAST
                    contrast()
                p["ops"] = ops
            if self.has_flow and self.scale_flow:                                             # :838-843, :208-209
                p["flow_sx"], p["flow_sy"] = np.float32(rw / W), np.float32(rh / H)
        return p

    def output_size(self, params):
        if self.training:
            return self.crop_size
        sizes = {(int(p["rh"]), int(p["rw"])) for p in params}
        assert len(sizes) == 1, "evaluation batches hold frames of one size"
        return sizes.pop()

    # -- device ---------------------------------------------------------------------------------------------------------
    def __call__(self, data, params=None, rng=np.random):
        """data: {'imgs': u8 [B,I,H,W,3], 'gt_fw_flows' / 'gt_bw_flows': fp32 [B,H,W,2] (has_flow, training),
        'pl_masks': u8 [B,I,H,W] (has_pl, training), 'ann': passed through} as CUDA tensors -> the collated batch the
        reference's loader yields: 'imgs' = I tensors [B,3,h,w], 'gt_fw_flows' / 'gt_bw_flows' = [tensor [B,2,h,w]],
        'pl_masks' = I tensors [B,h,w]; other keys are kept."""
        frames = data["imgs"]
        if not frames.is_cuda:
            raise _lib.RcfHipError("rcf_amd.data_pipeline needs CUDA (HIP) tensors: there is no CPU fallback")
        assert frames.dtype == torch.uint8 and frames.dim() == 5 and frames.shape[-1] == 3, "frames: u8 [B,I,H,W,3]"
        frames = frames.contiguous()
        B, I, H, W, _ = frames.shape
        if params is None:
            params = np.stack([self.sample_params(H, W, rng) for _ in range(B)])
        params = np.ascontiguousarray(params, dtype=PARAMS_DTYPE)
        assert params.shape == (B,)
        oh, ow = self.output_size(params)
        dev = frames.device
        prm = torch.from_numpy(params.view(np.uint8).reshape(B, -1)).to(dev, non_blocking=True)
        mean = np.asarray(MEAN, dtype=np.float32)
        std = np.asarray(STD, dtype=np.float32)
        out = dict(data)
        imgs = torch.empty((I, B, 3, oh, ow), dtype=torch.float32, device=dev)
        _lib.call("rcf_aug_frames_u8", _p(frames), B, I, H, W, _p(prm), _p(imgs), oh, ow, mean.ctypes.data, std.ctypes.data,
                  _stream())
        out["imgs"] = list(imgs.unbind(0))
        if self.training and self.has_flow:
            fl = torch.stack([data["gt_fw_flows"], data["gt_bw_flows"]], dim=1).contiguous()      # [B,2,H,W,2]
            assert fl.dtype == torch.float32 and fl.shape == (B, 2, H, W, 2), "flows: fp32 [B,H,W,2] per direction"
            fo = torch.empty((2, B, 2, oh, ow), dtype=torch.float32, device=dev)
            _lib.call("rcf_aug_flows_f32", _p(fl), B, 2, H, W, _p(prm), _p(fo), oh, ow, _stream())
            out["gt_fw_flows"], out["gt_bw_flows"] = [fo[0]], [fo[1]]
        if self.training and self.has_pl:
            pl = data["pl_masks"].contiguous()
            assert pl.dtype == torch.uint8 and pl.shape == (B, I, H, W), "pl_masks: u8 [B,I,H,W]"
            po = torch.empty((I, B, oh, ow), dtype=torch.float32, device=dev)
            _lib.call("rcf_aug_masks_u8", _p(pl), B, I, H, W, _p(prm), _p(po), oh, ow, _stream())
            out["pl_masks"] = list(po.unbind(0))
        out["aug_params"] = params
        release = out.pop("_release", None)                 # BatchUploader: every kernel reading its buffers is enqueued now
        if release is not None:
            release()
        return out

    def __repr__(self):
        return (f"Transform(training={self.training}, strong_aug={self.strong_aug}, has_flow={self.has_flow}, "
                f"has_pl={self.has_pl}, scale_flow={self.scale_flow})")


def get_transform(args, training):
    """dataset/transforms.py:926-929"""
    kw = args.train_transform_kwargs if training else args.test_transform_kwargs
    cls = getattr(args, "transform_cls", "Transform")
    if cls != "Transform":
        raise NotImplementedError(cls)
    return Transform(training=training, **kw)


# ---- decoded batch -> HBM (dataset/data.py:70-151 hands PIL images and np.load'ed flows to the transform) ---------------
# JPEG / PNG decoding stays on the host: this image has no hardware or device decoder library, a software Huffman + IDCT
# decoder on the GPU would be a second project, and the bytes are small (a 480x854 JPEG pair is ~0.2 MB against 2.5 MB
# decoded).  What IS on this path is everything after the decoder: the decoded arrays of a batch are written once into
# page-locked staging buffers (np.load reads the `.npy` flows straight into them: no intermediate array), uploaded on a
# copy stream while the previous batch trains, and handed to `Transform.__call__` in the layout it gathers from.
def load_flow_npy_into(path, out):
    """dataset/data.py:122-128: `np.load(path)` of a RAFT flow [H,W,2] float32 -- read into `out` (a numpy view of pinned
    memory).  `.npy` v1/v2 headers are parsed by numpy itself; the payload is read without a temporary array."""
    with open(path, "rb") as f:
        major, minor = np.lib.format.read_magic(f)
        shape, fortran, dtype = (np.lib.format.read_array_header_1_0 if major == 1 else np.lib.format.read_array_header_2_0)(f)
        if fortran or tuple(shape) != tuple(out.shape) or np.dtype(dtype) != out.dtype:
            out[...] = np.load(path).astype(out.dtype, copy=False).reshape(out.shape)     # odd file: the slow, general way
            return out
        n = f.readinto(memoryview(out).cast("B"))
        if n != out.nbytes:
            raise IOError(f"{path}: {n} of {out.nbytes} payload bytes")
    return out


class BatchUploader:
    """Double-buffered host -> device path for decoded batches.

        up = BatchUploader(B, I, H, W, has_pl=..., device="cuda:0")
        stage = up.stage()                 # dict of numpy views on pinned memory: fill them (decoder / np.load workers)
        ...  stage["imgs"][b, i] = np.asarray(pil_image);  load_flow_npy_into(path, stage["gt_fw_flows"][b])
        data = up.upload()                 # async copies on the copy stream; returns device tensors for Transform(data)

    `upload` makes the compute stream wait for the copies (an event, no host sync) and flips to the other staging set, so
    the loader may fill batch t+1 while batch t is copied and transformed.  Every staging set keeps the event of its own
    last host -> device copy and `stage()` blocks on it before handing the set out again: a host that runs several
    batches ahead of the GPU (normal with asynchronous launches) cannot overwrite pinned memory a queued copy has yet to
    read.  Pageable -> pinned copies, `torch.stack` collation and per-sample `.cuda()` calls of a default DataLoader path
    do not exist here."""

    def __init__(self, B, I, H, W, has_flow=True, has_pl=False, device="cuda:0", sets=2):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchUploader stages batches for the GPU: there is no CPU path")
        shapes = {"imgs": ((B, I, H, W, 3), torch.uint8)}
        if has_flow:
            shapes["gt_fw_flows"] = ((B, H, W, 2), torch.float32)
            shapes["gt_bw_flows"] = ((B, H, W, 2), torch.float32)
        if has_pl:
            shapes["pl_masks"] = ((B, I, H, W), torch.uint8)
        self.host = [{k: torch.empty(s, dtype=dt).pin_memory() for k, (s, dt) in shapes.items()} for _ in range(sets)]
        self.dev = [{k: torch.empty(s, dtype=dt, device=self.device) for k, (s, dt) in shapes.items()} for _ in range(sets)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.done = [None] * sets            # event: set i's device tensors were last read (compute stream)
        self.copied = [None] * sets          # event: set i's pinned buffers were last read by a host -> device copy (copy stream)
        self.cur = 0
        self.nbytes = sum(t.numel() * t.element_size() for t in self.host[0].values())

    def stage(self):
        """numpy views of the current pinned staging set (host memory the loader writes).  Blocks until the last upload
        that read THIS set has left its pinned buffers (normally long past: there are `sets` - 1 uploads in between)."""
        ev = self.copied[self.cur]
        if ev is not None:
            ev.synchronize()
        return {k: t.numpy() for k, t in self.host[self.cur].items()}

    def upload(self):
        i = self.cur
        comp = torch.cuda.current_stream(self.device)
        if self.done[i] is not None:
            self.copy_stream.wait_event(self.done[i])          # the batch that used these device buffers has been consumed
        with torch.cuda.stream(self.copy_stream):
            for k, t in self.host[i].items():
                self.dev[i][k].copy_(t, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        comp.wait_event(ready)
        self.cur = (i + 1) % len(self.host)
        out = dict(self.dev[i])
        out["_release"] = lambda: self._release(i)
        self.copied[i] = ready
        return out

    def _release(self, i):
        """call after the last kernel that reads the returned tensors was enqueued (Transform.__call__ is the only reader:
        it writes fresh output tensors)"""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.done[i] = ev

    def wait_host(self):
        """block until every upload issued so far has left its pinned buffers (`stage()` already waits for the set it hands
        out; this is for a caller that wants to touch all of them, e.g. before freeing the uploader)"""
        for ev in self.copied:
            if ev is not None:
                ev.synchronize()
