"""Offline 480x854 CRF callers of the reference on the HIP mean-field CRF:

* `refine` / `refine_batch`: the post-processor of tools/pydenseCRF/crf.py:57-89 (Gaussian pre-blur of the u8 mask,
  normalise by its maximum, clip, unary = -log, bilateral pairwise term only, 50 iterations, MAP), same signature.
* `double_crf_merge`: the CRF / NCut-CRF merge of tools/SemanticConstraintsAndMAA/semantic_constraints.py:299-324
  (`crf_head_single`, crf_scale 0.7, on the input mask; `crf_head`, crf_scale 0.5, on the refined mask; product, or
  the single-CRF mask when the two disagree by more than `umi_th`).

`refine` reproduces pydensecrf's arithmetic: DenseCRF2D.addPairwiseBilateral's default SYMMETRIC kernel normalisation
(N^1/2 K N^1/2, Kraehenbuehl & Koltun's densecrf), selected in the HIP kernel by rcf_crf_soft_ex(normalization=1); the
merge uses tools/torchCRF's (the reference's `CRFHead.crf`).  pydensecrf is not importable in this environment and is not
vendored by the reference: the symmetric mode is checked against oracle/crf_ref.c's restatement of the published
algorithm (parity-unpinned; tests/test_crf_gpu.py, tests/test_crf_oracle_cpu.py).
"""
import numpy as np
import torch

from .crf import crf_soft_batched


def _unary_from_u8(mask, gk):
    """tools/pydenseCRF/crf.py:60-68 on the host (an offline tool; the reference does the same in numpy)"""
    U = mask.astype(np.float64)
    if int(4.0 * gk + 0.5) > 0:                       # scipy's default truncate: sigma 0.1 -> a 1-tap kernel
        from scipy.ndimage import gaussian_filter
        U = gaussian_filter(mask, sigma=gk).astype(np.float64)
    U = U / (np.amax(U) + 1e-8)
    U = np.clip(U, 1e-6, 1.0 - 1e-6)
    UU = np.stack([-np.log(1.0 - U), -np.log(U)], axis=-1)          # [H, W, 2]: label-minor, as the C ABI wants it
    return np.float32(UU).reshape(-1, 2)


def refine_batch(masks, images, gk=0.1, sxy=60.0, srgb=5.0, compat=5.0, iters=50, device="cuda:0", symmetric=True):
    """masks u8 [n,H,W] (0..255), images u8 [n,H,W,3] -> float32 [n,H,W] of 0/1 (one library call for all frames).
    symmetric=True is pydensecrf's normalisation (what tools/pydenseCRF/crf.py runs); False is tools/torchCRF's."""
    masks, images = np.asarray(masks), np.ascontiguousarray(images)
    n, H, W = masks.shape
    unary = torch.from_numpy(np.stack([_unary_from_u8(m, gk) for m in masks])).to(device)
    rgb = torch.from_numpy(images).to(device)
    out = crf_soft_batched(rgb, unary, W, H, 0.0, 0.0, float(compat), float(sxy), float(srgb), int(iters),
                           symmetric=symmetric)
    return out.float().cpu().numpy()


def refine(mask, image, gk, sxy, srgb, compat, gtmask, iters=50, device="cuda:0", symmetric=True):
    """tools/pydenseCRF/crf.py:57 `refine(mask, image, gk, sxy, srgb, compat, gtmask)`"""
    new_mask = refine_batch(mask[None], image[None], gk, sxy, srgb, compat, iters, device, symmetric)[0]
    if gtmask is not None:
        gt, bm = gtmask > 0.1, new_mask > 0.1
        return new_mask, np.float32(np.sum(gt & bm)) / np.float32(np.sum(gt | bm))
    return new_mask


def umi(a, b):
    """semantic_constraints.py:271-277: union minus intersection (nan for two empty masks)"""
    i, u = a & b, a | b
    if u.sum() == 0:
        return float("nan")
    return u.sum() - i.sum()


@torch.no_grad()
def double_crf_merge(crf_head_single, crf_head, images, masks, refined_masks, umi_th=None):
    """semantic_constraints.py:303-322.  images [n,H,W,3] in [0,1] (`unstandardize=False`), masks / refined_masks
    [n,H,W] in [0,1] -> merged masks [n,H,W]"""
    a = crf_head_single(images, masks, unstandardize=False)
    b = crf_head(images, refined_masks, unstandardize=False)
    if umi_th is None:
        return a * b
    out = a * b
    for i in range(a.shape[0]):
        if umi(a[i].cpu().numpy() > 0.5, b[i].cpu().numpy() > 0.5) > umi_th:
            out[i] = a[i]                                # likely captures different things: keep the single CRF
    return out
