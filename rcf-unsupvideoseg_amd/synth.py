"""Deterministic synthetic inputs and weights.

Everything comes from numpy `Generator(PCG64(seed))`, so the build container and the GPU
box produce identical bytes.  seed = 1000 * config_id + sample_index for data; weights
have their own seed.  No dataset or checkpoint exists in either environment.
"""
import numpy as np

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)   # dataset/transforms.py:893
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def _rng(seed):
    return np.random.Generator(np.random.PCG64(int(seed)))


def smooth_rgb(H, W, seed):
    """u8 [H,W,3]: low-frequency sinusoids + 3 coloured convex blobs + N(0,4) noise."""
    g = _rng(seed)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    img = np.zeros((H, W, 3), dtype=np.float32)
    for c in range(3):
        acc = np.full((H, W), 128.0, dtype=np.float32)
        for _ in range(4):
            fy, fx = g.uniform(0.5, 3.0, size=2) * 2 * np.pi / np.array([H, W])
            ph, amp = g.uniform(0, 2 * np.pi), g.uniform(10, 40)
            acc += amp * np.sin(fy * yy + fx * xx + ph).astype(np.float32)
        img[..., c] = acc
    for _ in range(3):
        cy, cx = g.uniform(0.2, 0.8) * H, g.uniform(0.2, 0.8) * W
        ry, rx = g.uniform(0.08, 0.25) * H, g.uniform(0.08, 0.25) * W
        col = g.uniform(0, 255, size=3).astype(np.float32)
        inside = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
        img[inside] = 0.25 * img[inside] + 0.75 * col
    img += g.normal(0, 2.0, size=img.shape).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def noise_rgb(H, W, seed):
    """u8 [H,W,3] uniform noise: worst-case permutohedral lattice occupancy."""
    return _rng(seed).integers(0, 256, size=(H, W, 3), dtype=np.uint8)


def voronoi_affine_flow(H, W, seed, nseg=3, max_mag=20.0):
    """f32 [2,H,W] (x,y pixel units): per-Voronoi-cell affine motion, |flow| <= max_mag, + N(0,0.25)."""
    g = _rng(seed)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    sites = np.stack([g.uniform(0, H, nseg), g.uniform(0, W, nseg)], 1)
    d = np.stack([(yy - sy) ** 2 + (xx - sx) ** 2 for sy, sx in sites], 0)
    seg = np.argmin(d, 0)
    flow = np.zeros((2, H, W), dtype=np.float32)
    for k in range(nseg):
        A = g.normal(0, 0.02, size=(2, 2)).astype(np.float32)
        t = g.uniform(-8, 8, size=2).astype(np.float32)
        u = A[0, 0] * (xx - W / 2) + A[0, 1] * (yy - H / 2) + t[0]
        v = A[1, 0] * (xx - W / 2) + A[1, 1] * (yy - H / 2) + t[1]
        m = seg == k
        flow[0][m], flow[1][m] = u[m], v[m]
    flow += g.normal(0, 0.5, size=flow.shape).astype(np.float32)
    mag = np.sqrt((flow ** 2).sum(0, keepdims=True))
    flow *= np.minimum(1.0, max_mag / np.maximum(mag, 1e-6))
    return flow.astype(np.float32), seg.astype(np.uint8)


def warp_np(img, flow):
    """Backward bilinear warp (border clamp) of f32 [C,H,W] by flow [2,H,W]; numpy only."""
    C, H, W = img.shape
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    x = np.clip(xx + flow[0], 0, W - 1)
    y = np.clip(yy + flow[1], 0, H - 1)
    x0, y0 = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    x1, y1 = np.minimum(x0 + 1, W - 1), np.minimum(y0 + 1, H - 1)
    ax, ay = (x - x0).astype(np.float32), (y - y0).astype(np.float32)
    out = (img[:, y0, x0] * (1 - ax) * (1 - ay) + img[:, y0, x1] * ax * (1 - ay)
           + img[:, y1, x0] * (1 - ax) * ay + img[:, y1, x1] * ax * ay)
    return out.astype(np.float32)


def normalize_rgb(u8):
    """u8 [H,W,3] -> f32 [3,H,W], ImageNet-normalised (dataset/transforms.py:893)."""
    x = u8.astype(np.float32) / 255.0
    return ((x - IMAGENET_MEAN) / IMAGENET_STD).transpose(2, 0, 1).copy()


def make_pair(H, W, seed):
    """One training sample: 2 normalised frames + fw + bw flow (SURVEY Appendix F shapes)."""
    rgb0 = smooth_rgb(H, W, seed)
    fw, _ = voronoi_affine_flow(H, W, seed + 500000)
    f0 = rgb0.astype(np.float32).transpose(2, 0, 1)
    f1 = np.clip(warp_np(f0, fw), 0, 255)                    # frame 1 = frame 0 backward-warped
    bw = -warp_np(fw, fw)                                    # bw flow = -fw warped
    rgb1 = f1.transpose(1, 2, 0).astype(np.uint8)
    return normalize_rgb(rgb0), normalize_rgb(rgb1), fw, bw.astype(np.float32)


def make_batch(B, H, W, config_id=1, first_index=0):
    """Batch dict in the layout `RCFModel.forward` consumes (numpy arrays; caller converts)."""
    s = [make_pair(H, W, 1000 * config_id + first_index + i) for i in range(B)]
    return {
        "imgs": [np.stack([p[0] for p in s]), np.stack([p[1] for p in s])],
        "gt_fw_flows": [np.stack([p[2] for p in s])],
        "gt_bw_flows": [np.stack([p[3] for p in s])],
        "seq_ids": np.arange(B, dtype=np.int64),
        "seq_names": [f"synth{first_index + i}" for i in range(B)],
        "paths": [[f"synth/{first_index + i:05d}.jpg" for i in range(B)] for _ in range(2)],
    }


def dropout_scale(n, channels, p=0.1, seed=0):
    """f32 [n, channels]: one nn.Dropout2d draw written down -- Bernoulli(1 - p) per (sample, channel) plane, kept planes
    scaled by 1 / (1 - p) (models/decode_head.py:84-87, models/fcn_head.py:142-147).  Parity tests hand the SAME draw to the
    reference / oracle (oracle.FixedDropout2d) and to FCNHead.keep_mask."""
    keep = (_rng(910000 + seed).uniform(size=(n, channels)) >= p).astype(np.float32)
    return keep / np.float32(1.0 - p)


def soft_blob_mask(H, W, seed):
    """f32 [H,W] in [0,1]: Gaussian-blurred blob (CRF input of SURVEY §8(d))."""
    g = _rng(seed)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    cy, cx = g.uniform(0.3, 0.7) * H, g.uniform(0.3, 0.7) * W
    ry, rx = g.uniform(0.15, 0.3) * H, g.uniform(0.15, 0.3) * W
    d = np.sqrt(((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2)
    return (1.0 / (1.0 + np.exp((d - 1.0) * 6.0))).astype(np.float32)


def make_pl_masks(B, H, W, config_id=1, first_index=0):
    """batch['pl_masks'] of stage 2.2 (dataset/data.py:137-151): one soft mask in [0,1] per frame, 2 x [B,H,W]"""
    return [np.stack([soft_blob_mask(H, W, 1000 * config_id + first_index + i + 7000 * (f + 1)) for i in range(B)])
            for f in range(2)]


def loader_sample(seed, H=480, W=854):
    """What dataset/data.py:73-151 hands to the transform for one training sample, decoded: frames u8 [2,H,W,3] (smooth
    content; the right quarter is uniform noise so that every hue sector / saturation extreme occurs), forward / backward
    flow fp32 [H,W,2] (the `.npy` layout, :122-128) and pseudo-label masks u8 [2,H,W] (:137-150)."""
    frames = np.stack([smooth_rgb(H, W, seed + 31 * i) for i in range(2)])
    for i in range(2):
        frames[i, :, W - W // 4:] = noise_rgb(H, W // 4, seed + 77 + i)
    fw, _ = voronoi_affine_flow(H, W, seed + 500000)
    bw, _ = voronoi_affine_flow(H, W, seed + 600000)
    pl = np.stack([np.round(soft_blob_mask(H, W, seed + 7000 * (i + 1)) * 255.0).astype(np.uint8) for i in range(2)])
    return dict(frames=frames, fw=np.ascontiguousarray(fw.transpose(1, 2, 0)), bw=np.ascontiguousarray(bw.transpose(1, 2, 0)), pl=pl)


def eval_inputs(seed, N=6, C=4, h=30, w=54, H=120, W=214):
    """Inputs of the evaluation-metric fixtures (tests/golden/make_golden_eval.py): smooth soft masks [N,C,h,w] (softmax
    of low-frequency logits), annotations [N,H,W] u8 in {0, 128 (ignore), 255}, sequence names (two frames each)."""
    g = _rng(seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    logits = np.zeros((N, C, h, w), dtype=np.float32)
    for n in range(N):
        for c in range(C):
            for _ in range(3):
                fy, fx = g.uniform(0.5, 2.5, size=2) * 2 * np.pi / np.array([h, w])
                logits[n, c] += g.uniform(1.0, 3.0) * np.sin(fy * yy + fx * xx + g.uniform(0, 2 * np.pi)).astype(np.float32)
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    masks = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
    YY, XX = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    ann = np.zeros((N, H, W), dtype=np.uint8)
    for n in range(N):
        cy, cx = g.uniform(0.3, 0.7) * H, g.uniform(0.3, 0.7) * W
        ry, rx = g.uniform(0.15, 0.35) * H, g.uniform(0.15, 0.35) * W
        d = ((YY - cy) / ry) ** 2 + ((XX - cx) / rx) ** 2
        ann[n][d < 1.0] = 255
        ann[n][(d >= 1.0) & (d < 1.15)] = 128                  # a ring of "ignore" pixels around the object
    ann[N - 1] = 0                                             # an empty annotation
    names = [f"seq{n // 2}" for n in range(N)]
    return masks, ann, names


def fill_state_dict(shapes, seed=7, bn3_gamma=0.5, seg_scale=10.0):
    """Seeded weights for every entry of a state-dict `shapes` mapping name -> shape.

    conv/linear weights: He-normal(fan_out); BN gamma 1 (bn3: `bn3_gamma`, so residual branches
    are alive), beta 0, running stats (0,1); conv_seg scaled by `seg_scale` so the 4 mask logits
    separate (SURVEY §7 'bit-exact argmax' note).  Order of draws = sorted(names)."""
    g = _rng(seed)
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = np.zeros((), dtype=np.int64)
        elif leaf == "running_mean":
            out[name] = np.zeros(shp, dtype=np.float32)
        elif leaf == "running_var":
            out[name] = np.ones(shp, dtype=np.float32)
        elif len(shp) == 1 and leaf == "weight":            # BN gamma
            gamma = bn3_gamma if name.split(".")[-2] == "bn3" else 1.0
            out[name] = np.full(shp, gamma, dtype=np.float32)
        elif len(shp) == 1 and leaf == "bias":
            is_bn = (name[:-len("bias")] + "running_mean") in shapes
            out[name] = np.zeros(shp, dtype=np.float32) if is_bn else \
                g.normal(0, 0.01, size=shp).astype(np.float32)
        else:                                               # conv weight [Co,Ci,(k,k)|(k)]
            fan_out = shp[0] * int(np.prod(shp[2:]))
            w = g.normal(0, np.sqrt(2.0 / fan_out), size=shp).astype(np.float32)
            if "conv_seg" in name:
                w *= seg_scale
            out[name] = w
    return out


def fill_vit_state_dict(shapes, seed=21):
    """Seeded weights for the DINO ViT parity fixtures (no checkpoint can be downloaded): N(0, 0.05) matrices,
    N(0, 0.02) biases / tokens, LayerNorm scale 1 + N(0, 0.1).  Same generator on both sides of every comparison."""
    g = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for k, shp in shapes.items():
        if "norm" in k and k.endswith("weight"):
            out[k] = (1.0 + 0.1 * g.standard_normal(shp)).astype(np.float32)
        elif k.endswith("bias") or k in ("cls_token", "pos_embed"):
            out[k] = (0.02 * g.standard_normal(shp)).astype(np.float32)
        else:
            out[k] = (0.05 * g.standard_normal(shp)).astype(np.float32)
    return out


def grad_sketch(named_grads, k=8):
    """A small linear fingerprint of a set of gradient tensors: per tensor, k sums of its elements under pseudo-random +-1
    sign patterns (a hash of the element index: the same pattern on every device and in every process).  E[(s - t)^2] of two
    sketches is the squared distance of the tensors they came from, so |sketch(g) - sketch(truth)| / |sketch(truth)| over a
    module's tensors estimates the relative VECTOR error of the module's gradient from a few kilobytes of fixture instead of
    100 MB of float64 gradients (tools/oracle_b8_selfdev.py, tests/test_model_gpu.py::test_fullsize_b8_gradients_vs_oracle).
    named_grads: {name: torch tensor}; returns {name: [k floats]}."""
    import torch
    out = {}
    for name, g in named_grads.items():
        v = g.detach().reshape(-1).double()
        idx = torch.arange(v.numel(), device=v.device, dtype=torch.int64)
        row = []
        for j in range(k):
            h = idx * (2654435761 + 81006 * j) + 2654435769 * (j + 1)
            h = (h ^ (h >> 15)) * 2246822519
            h = h ^ (h >> 13)
            sign = 1.0 - 2.0 * ((h >> 7) & 1).double()
            row.append(float((sign * v).sum()))
        out[name] = row
    return out


def sketch_error(sketch, truth, prefix=None):
    """relative distance of two grad_sketch() results over the tensors whose name starts with `prefix` (None: all)"""
    num = den = 0.0
    for name, t in truth.items():
        if prefix is not None and not name.startswith(prefix):
            continue
        s = sketch[name]
        num += sum((a - b) ** 2 for a, b in zip(s, t))
        den += sum(b ** 2 for b in t)
    return (num / den) ** 0.5 if den > 0 else 0.0
