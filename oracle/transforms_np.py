"""ORACLE (test infrastructure): numpy restatement of the third-party image operators under the reference's data
pipeline (dataset/transforms.py:170-237 Resize, :249-306 RandomFlip, :427-511 RandomCrop, :557-687
PhotoMetricDistortion) -- mmcv-full 1.6.2's `imrescale` / `imflip` / `rgb2hsv` / `hsv2rgb` (requirements.txt:9), which
are thin wrappers over OpenCV (`cv2.resize`, `cv2.cvtColor`).  Neither mmcv nor cv2 can be imported in this environment
and the reference holds no vectors for them: these functions restate OpenCV's published algorithms for 8-bit images
(imgproc/resize.cpp: INTER_LINEAR with 11-bit fixed-point coefficients, INTER_NEAREST index rule;
imgproc/color_hsv: RGB2HSV_b integer tables with hrange 180, HSV2RGB through the float path) and are
PARITY-UNPINNED.  Everything the reference itself does around them -- the order of the random draws, the scale rule,
crop, flip, the u8 round trips of the photometric chain, /255, normalisation, the flow / pseudo-label handling -- IS
pinned: tests/golden/make_golden_data.py runs the reference's own `Transform` with these functions standing in for mmcv's.
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def rescale_size(old_wh, scale):
    """mmcv.image.geometric.rescale_size with a (max_long_edge, max_short_edge) tuple -> ((new_w, new_h), factor)"""
    w, h = old_wh
    max_long, max_short = max(scale), min(scale)
    f = min(max_long / max(h, w), max_short / min(h, w))
    return (int(w * float(f) + 0.5), int(h * float(f) + 0.5)), f


def _linear_taps(dst, src, horizontal=True):
    """cv2 resize INTER_LINEAR: per destination index the two source taps and the two 11-bit coefficients.  The
    horizontal table resets the fraction at the borders (sx < 0 -> fx = 0, sx = 0; sx >= W-1 -> fx = 0, sx = W-1); the
    vertical one only clips the row indices (resizeGeneric_Invoker's `clip(sy + k, 0, H)`) and keeps the weights."""
    scale = 1.0 / (float(dst) / float(src))                 # scale_x = 1 / inv_scale_x (double)
    d = np.arange(dst, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = fx - sx.astype(np.float32)
    if horizontal:
        lo = sx < 0
        fx[lo], sx[lo] = 0.0, 0
        hi = sx >= src - 1
        fx[hi], sx[hi] = 0.0, src - 1
    a0 = np.rint((1.0 - fx).astype(np.float32) * np.float32(COEF_SCALE)).astype(np.int64)     # saturate_cast<short>
    a1 = np.rint(fx * np.float32(COEF_SCALE)).astype(np.int64)
    return np.clip(sx, 0, src - 1), np.clip(sx + 1, 0, src - 1), a0, a1


def resize_linear_u8(img, new_wh):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for uint8 [H,W,C]"""
    H, W = img.shape[:2]
    w, h = new_wh
    x0, x1, ax0, ax1 = _linear_taps(w, W)
    y0, y1, ay0, ay1 = _linear_taps(h, H, horizontal=False)
    s = img.astype(np.int64)
    rows = s[:, x0] * ax0[None, :, None] + s[:, x1] * ax1[None, :, None]          # horizontal pass, [H, w, C] ints
    r0, r1 = rows[y0], rows[y1]
    out = ((((ay0[:, None, None] * (r0 >> 4)) >> 16) + ((ay1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2)
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_nearest(img, new_wh):
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST): source index = min(floor(dst * src/dst_size), src - 1)"""
    H, W = img.shape[:2]
    w, h = new_wh
    ifx, ify = 1.0 / (float(w) / float(W)), 1.0 / (float(h) / float(H))
    xs = np.minimum(np.floor(np.arange(w) * ifx).astype(np.int64), W - 1)
    ys = np.minimum(np.floor(np.arange(h) * ify).astype(np.int64), H - 1)
    return img[ys][:, xs]


def imrescale(img, scale, return_scale=False, interpolation="bilinear"):
    """mmcv.imrescale(img, (long, short), return_scale, interpolation)"""
    h, w = img.shape[:2]
    new_wh, f = rescale_size((w, h), scale)
    if interpolation == "nearest":
        out = resize_nearest(img, new_wh)
    elif interpolation == "bilinear":
        assert img.dtype == np.uint8, "the bilinear restatement covers 8-bit images (what the pipeline resizes)"
        out = resize_linear_u8(img, new_wh)
    else:
        raise NotImplementedError(interpolation)
    return (out, f) if return_scale else out


def imflip(img, direction="horizontal"):
    assert direction == "horizontal"
    return np.flip(img, axis=1)


_SDIV = np.zeros(256, dtype=np.int64)
_HDIV = np.zeros(256, dtype=np.int64)
_SDIV[1:] = np.rint((255 << 12) / np.arange(1, 256, dtype=np.float64)).astype(np.int64)
_HDIV[1:] = np.rint((180 << 12) / (6.0 * np.arange(1, 256, dtype=np.float64))).astype(np.int64)


def rgb2hsv(img):
    """cv2.cvtColor(img, cv2.COLOR_RGB2HSV) for uint8: integer RGB2HSV_b, H in [0, 180)"""
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = v - vmin
    vr, vg = v == r, v == g
    s = (diff * _SDIV[v] + (1 << 11)) >> 12
    h = np.where(vr, g - b, np.where(vg, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * _HDIV[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return np.stack([h, s, v], axis=-1).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])     # (b, g, r) table indices


def hsv2rgb(img):
    """cv2.cvtColor(img, cv2.COLOR_HSV2RGB) for uint8: through the float path (HSV2RGB_f, hscale = 6/180), rounded"""
    h = img[..., 0].astype(np.float32) * np.float32(6.0 / 180.0)
    s = img[..., 1].astype(np.float32) * np.float32(1.0 / 255.0)
    v = img[..., 2].astype(np.float32) * np.float32(1.0 / 255.0)
    sector = np.floor(h).astype(np.int64)
    f = h - sector.astype(np.float32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    f = np.where(bad, np.float32(0), f)
    one = np.float32(1.0)
    tab = np.stack([v, v * (one - s), v * (one - s * f), v * (one - s * (one - f))], axis=-1)
    idx = _SECTOR[sector]                                     # [..., 3] -> (b, g, r)
    bgr = np.take_along_axis(tab, idx, axis=-1)
    gray = (s == 0)[..., None]
    bgr = np.where(gray, v[..., None], bgr)
    rgb = bgr[..., ::-1]
    return np.clip(np.rint(rgb * np.float32(255.0)), 0, 255).astype(np.uint8)


# ----------------------------------------------------------------------------------------------------------------------
# The reference's own pipeline (dataset/transforms.py:884-924), restated as a function of one sample's random decisions
# -- PINNED by tests/golden/data_pipeline.json (the reference's `Transform` run with the operators above for mmcv's).
MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def convert_one_img(img, alpha=1, beta=0):
    """dataset/transforms.py:590-594 (numpy keeps float32: python scalars are weak)"""
    out = img.astype(np.float32) * np.float32(alpha) + np.float32(beta)
    return np.clip(out, 0, 255).astype(np.uint8)


def photometric(img, p):
    """dataset/transforms.py:650-682 with the decisions of `p` (ops bits as in include/rcf_hip.h)"""
    ops = int(p["ops"])
    if ops & 1:
        img = convert_one_img(img, beta=p["beta"])
    if (ops & 2) and not (ops & 16):
        img = convert_one_img(img, alpha=p["alpha_c"])
    if ops & 4:
        hsv = rgb2hsv(img)
        hsv[:, :, 1] = convert_one_img(hsv[:, :, 1], alpha=p["alpha_s"])
        img = hsv2rgb(hsv)
    if ops & 8:
        hsv = rgb2hsv(img)
        hsv[:, :, 0] = ((hsv[:, :, 0].astype(int) + float(p["hue_delta"])) % 180).astype(np.uint8)
        img = hsv2rgb(hsv)
    if (ops & 2) and (ops & 16):
        img = convert_one_img(img, alpha=p["alpha_c"])
    return img


def _geom(a, p, oh, ow, nearest):
    H, W = a.shape[:2]
    size = (int(p["rw"]), int(p["rh"]))
    r = resize_nearest(a, size) if nearest else resize_linear_u8(a, size)
    r = r[int(p["crop_y"]):int(p["crop_y"]) + oh, int(p["crop_x"]):int(p["crop_x"]) + ow]
    return np.flip(r, axis=1) if int(p["flip"]) else r


def apply_params(sample, p, oh, ow):
    """sample: dict(frames u8 [I,H,W,3], fw / bw fp32 [H,W,2] or None, pl u8 [I,H,W] or None) -> dict of float32 arrays:
    imgs [I,3,oh,ow], fw / bw [2,oh,ow], pl [I,oh,ow]"""
    out = {}
    imgs = []
    for f in sample["frames"]:
        x = photometric(np.ascontiguousarray(_geom(f, p, oh, ow, False)), p)
        x = x.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
        imgs.append((x - MEAN[:, None, None]) / STD[:, None, None])
    out["imgs"] = np.stack(imgs)
    for k in ("fw", "bw"):
        if sample.get(k) is not None:
            f = _geom(sample[k], p, oh, ow, True) * np.array([p["flow_sx"], p["flow_sy"]], dtype=np.float32)
            out[k] = np.ascontiguousarray(f.transpose(2, 0, 1)).astype(np.float32)
    if sample.get("pl") is not None:
        out["pl"] = np.stack([_geom(m, p, oh, ow, True).astype(np.float32) / np.float32(255.0) for m in sample["pl"]])
    return out
