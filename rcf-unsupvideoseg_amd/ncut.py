"""Soft-NCut on DINO features: `soft_ncut_value`, `ncut_refine`, `NCutHead`
(tools/SemanticConstraintsAndMAA/semantic_constraints.py:21-184) and `NCutEvalHead` (maa.py:19-138) with the
reference's signatures, on the HIP kernels.

The reference builds the [n, n] affinity with autograd and lets torch differentiate the NCut through two dense
mat-vecs per step.  Here the thresholded affinity A = (K^ K^T > tau ? 1 : eps) is built ONCE per image (one Gram
product on the split-bf16 GEMM + a threshold pass) and every Adam step costs one mat-vec u = A x plus a closed-form
gradient:  with s = A 1, a = s.x, S = sum s, cut = a - x.u,
    NCut = cut/a + cut/(S-a),     dNCut/dx = (s - 2u)(1/a + 1/(S-a)) - cut s/a^2 + cut s/(S-a)^2
(A is symmetric).  Reductions in fp64; the Adam update is the fused kernel of the trainer (coupled weight decay, like
torch.optim.Adam).
"""
import torch
import torch.nn as nn

from . import _lib, ops
from .ops import _p, _stream


def _affinity(feats, tau, eps):
    """feats [1, T, C] (row 0 = [CLS], dropped) -> thresholded affinity [n, npad] (n = T-1) on the GPU"""
    f = feats[0, 1:, :].contiguous().float()
    fn = ops.l2_normalize_rows(f)                                       # F.normalize(p=2, dim=1)
    n = fn.shape[0]
    npad = (n + 3) // 4 * 4
    A = torch.empty((n, npad), dtype=torch.float32, device=f.device)
    ops.gemm_nt(fn, fn, out=A[:, :n])
    _lib.call("rcf_affinity_threshold_f32", _p(A), npad, n, float(tau), float(eps), _stream())
    return A, n, npad


class _NCut:
    """affinity of one image + the scratch of the value / gradient evaluation"""

    def __init__(self, feats, tau, eps):
        self.A, self.n, self.npad = _affinity(feats, tau, eps)
        dev = self.A.device
        self.u = torch.empty(self.n, dtype=torch.float64, device=dev)
        self.s = torch.empty(self.n, dtype=torch.float64, device=dev)
        self.have_s = False
        self.val = torch.empty(1, dtype=torch.float32, device=dev)

    def value_grad(self, x, grad=None):
        _lib.call("rcf_ncut_value_grad_f32", _p(self.A), self.npad, self.n, _p(x), _p(self.u), _p(self.s),
                  0 if self.have_s else 1, _p(grad), _p(self.val), _stream())
        self.have_s = True
        return self.val


@torch.no_grad()
def soft_ncut_value(feats, mask, tau, eps):
    """semantic_constraints.py:21-41 / maa.py:19-36: feats [1,T,C], mask [h,w] -> 0-dim NCut value"""
    x = mask.reshape(-1).contiguous().float()
    return _NCut(feats, tau, eps).value_grad(x).clone()[0]


@torch.no_grad()
def ncut_refine(feats, masks, tau=0.2, eps=1e-5, steps=10, learning_rate=1e-1, weight_decay=1e-6,
                visualize_interval=10, visualize=False):
    """semantic_constraints.py:44-77: `steps` Adam steps on the mask against the soft NCut, clamped to [0,1] after each"""
    shape = masks.shape
    x = masks.reshape(-1).contiguous().float().clone()
    nc = _NCut(feats, tau, eps)
    assert x.numel() == nc.n, f"mask has {x.numel()} cells, the feature map {nc.n}"
    g = torch.empty_like(x)
    m, v = torch.zeros_like(x), torch.zeros_like(x)
    for i in range(steps):
        nc.value_grad(x, g)
        ops.adam_step(x, g, m, v, learning_rate, i + 1, (0.9, 0.999), 1e-8, weight_decay)
        _lib.call("rcf_clamp01_f32", _p(x), x.numel(), _stream())
    return x.view(shape)


class _NCutBase(nn.Module):
    def __init__(self, args, resize_imgs_size=(480, 856), resize_masks_size=(480, 854), arch="vit_small", patch_size=8,
                 which_features="k", tau=0.2, eps=1e-5, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), model=None):
        super().__init__()
        from . import vit
        assert "vit" in arch, arch
        self.args, self.arch, self.patch_size, self.which_features = args, arch, patch_size, which_features
        self.resize_imgs_size, self.resize_masks_size = tuple(resize_imgs_size), tuple(resize_masks_size)
        self.tau, self.eps = tau, eps
        self.register_buffer("mean", torch.tensor(mean, dtype=torch.float32)[None, :, None, None], persistent=False)
        self.register_buffer("std", torch.tensor(std, dtype=torch.float32)[None, :, None, None], persistent=False)
        # get_dino_model downloads the DINO checkpoint (models/dino_vit.py:448-521); here the caller loads it with
        # load_state_dict -- the parameter names are the reference's
        self.model = model if model is not None else getattr(vit, arch)(patch_size=patch_size)
        for p in self.model.parameters():
            p.requires_grad = False
        self.h_featuremap = self.resize_imgs_size[0] // patch_size
        self.w_featuremap = self.resize_imgs_size[1] // patch_size

    def normalize(self, imgs):
        return (imgs.permute(0, 3, 1, 2) - self.mean) / self.std          # B,H,W,3 -> B,3,H,W

    def get_feats(self, imgs):
        return self.model.get_last_qkv(imgs, self.which_features)          # [B, T, dim]

    def _inputs(self, imgs, masks, standardize):
        if standardize:
            imgs = self.normalize(imgs)
        imgs = ops.resize_nchw(imgs.contiguous().float(), self.resize_imgs_size, False)      # F.interpolate(bilinear)
        hf, wf = self.h_featuremap, self.w_featuremap
        H, W = masks.shape[-2:]
        # F.interpolate(mode='nearest'): source index floor(dst * in / out)
        iy = (torch.arange(hf, device=masks.device) * (H / hf)).floor().long().clamp_(max=H - 1)
        ix = (torch.arange(wf, device=masks.device) * (W / wf)).floor().long().clamp_(max=W - 1)
        return imgs, masks[:, iy][:, :, ix].float()


class NCutHead(_NCutBase):
    """semantic_constraints.py:80-184: refine each mask with `steps` Adam steps on its soft NCut"""

    def __init__(self, args, steps=10, learning_rate=1e-1, weight_decay=1e-6, visualize_interval=10, visualize=False,
                 **kw):
        super().__init__(args, **kw)
        self.steps, self.learning_rate, self.weight_decay = steps, learning_rate, weight_decay

    @torch.no_grad()
    def forward(self, imgs, masks, standardize=False):
        imgs, small = self._inputs(imgs, masks, standardize)
        feats = self.get_feats(imgs)
        # the reference's soft_ncut_value reads feats[0] only (its callers pass one image at a time); batches are
        # refined image by image here
        out = torch.stack([ncut_refine(feats[b:b + 1], small[b], self.tau, self.eps, self.steps, self.learning_rate,
                                       self.weight_decay) for b in range(feats.shape[0])])
        return ops.resize_nchw(out[:, None].contiguous(), self.resize_masks_size, False)[:, 0].float()


class NCutEvalHead(_NCutBase):
    """maa.py:39-138: the soft NCut value of a mask (numpy [1])"""

    @torch.no_grad()
    def forward(self, imgs, masks, standardize=False):
        imgs, small = self._inputs(imgs, masks, standardize)
        feats = self.get_feats(imgs)
        return soft_ncut_value(feats, small[0], self.tau, self.eps)[None].cpu().numpy()
