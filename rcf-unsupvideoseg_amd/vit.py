"""DINO ViT forward on the HIP kernels: `VisionTransformer` with the parameter names,
constructor keywords and forward surfaces of models/dino_vit.py:176-276 (`forward`, `get_intermediate_layers`,
`get_last_selfattention`) plus `get_last_qkv` (what the reference reads through a forward hook on the last block's
qkv layer, tools/SemanticConstraintsAndMAA/maa.py:63-68).  Inference only (the reference freezes it).

Every matrix product -- patch embedding (an 8x8/8 conv), qkv / proj / MLP linears, q k^T and attn v per head -- runs on
the split-bf16 implicit-GEMM kernel (fp32 accuracy); LayerNorm, the row softmax and the small transposes are the
HBM-bound helpers of csrc/vit.hip; GELU and the residual adds live in the GEMM epilogue.  torch is used for parameter
preprocessing only (the bicubic position-embedding interpolation, cached per input size).
"""
import math
from functools import partial

import torch
import torch.nn as nn

from . import ops
from .layers import Conv2d

FP16_PAIRS = True      # fused attention on the fp16-pair arithmetic (operand range = absmax of qkv)


class _Linear(nn.Module):
    """nn.Linear's parameters ([out, in] weight = the B[N][K] operand of rcf_gemm_nt_f32)"""

    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        nn.init.trunc_normal_(self.weight, std=.02)


class _LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.weight, self.bias, self.eps = nn.Parameter(torch.ones(dim)), nn.Parameter(torch.zeros(dim)), eps


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.fc2 = _Linear(dim, hidden), _Linear(hidden, dim)


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads, self.scale = num_heads, (dim // num_heads) ** -0.5
        self.qkv, self.proj = _Linear(dim, dim * 3, bias=qkv_bias), _Linear(dim, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, eps):
        super().__init__()
        self.norm1, self.attn = _LayerNorm(dim, eps), Attention(dim, num_heads, qkv_bias)
        self.norm2, self.mlp = _LayerNorm(dim, eps), Mlp(dim, int(dim * mlp_ratio))


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size, self.patch_size = img_size, patch_size
        self.num_patches = (img_size // patch_size) * (img_size // patch_size)
        self.proj = Conv2d(in_chans, embed_dim, patch_size, stride=patch_size, bias=True)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=(224,), patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 norm_layer=None, eps=1e-6, **kwargs):
        super().__init__()
        if num_classes or qk_scale is not None or drop_rate or attn_drop_rate or drop_path_rate:
            raise NotImplementedError("inference-only DINO backbone: no classifier head, dropout or custom qk scale")
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.patch_embed = PatchEmbed(img_size[0], patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, eps) for _ in range(depth)])
        self.norm = _LayerNorm(embed_dim, eps)
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self._pos_cache = {}
        self.attn_bytes = 6 << 30       # budget for the materialised attention scores of one launch
        self.fused_attention = True     # False: q k^T / softmax / P v as separate launches (also used when maps are asked for)

    # ---------------------------------------------------------------- parameter preprocessing (torch, cached)
    def interpolate_pos_encoding(self, npatch, w, h):
        """models/dino_vit.py:219-238 (same call sequence), cached per input size"""
        key = (npatch, w, h, self.pos_embed.device, self.pos_embed._version)
        if key in self._pos_cache:
            return self._pos_cache[key]
        N = self.pos_embed.shape[1] - 1
        if npatch == N and w == h:
            pe = self.pos_embed
        else:
            cls_pe, patch_pe = self.pos_embed[:, 0], self.pos_embed[:, 1:]
            dim = self.pos_embed.shape[-1]
            w0, h0 = w // self.patch_embed.patch_size + 0.1, h // self.patch_embed.patch_size + 0.1
            patch_pe = nn.functional.interpolate(
                patch_pe.reshape(1, int(math.sqrt(N)), int(math.sqrt(N)), dim).permute(0, 3, 1, 2),
                scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)), mode="bicubic")
            assert int(w0) == patch_pe.shape[-2] and int(h0) == patch_pe.shape[-1]
            pe = torch.cat((cls_pe.unsqueeze(0), patch_pe.permute(0, 2, 3, 1).view(1, -1, dim)), dim=1)
        pe = pe.detach().contiguous()
        self._pos_cache = {key: pe}
        return pe

    # ---------------------------------------------------------------- forward pieces
    def prepare_tokens(self, x):
        """[B,3,w,h] image -> token buffer [B*T, dim] (row b*T is image b's [CLS]); models/dino_vit.py:240-251"""
        if not x.is_cuda:
            raise RuntimeError("VisionTransformer (HIP) needs its input on the GPU: there is no CPU fallback")
        B, _, w, h = x.shape
        pe = self.patch_embed
        img = ops.nchw_to_nhwc(x.contiguous().float(), pe.proj.cin_pad)
        tok = ops.conv2d_fwd(img, pe.proj._packed_weight(), pe.proj.bias, pe.patch_size, 0, 1)      # [B, w/p, h/p, dim]
        npatch, dim = tok.shape[1] * tok.shape[2], tok.shape[3]
        pos = self.interpolate_pos_encoding(npatch, w, h)[0]                                        # [T, dim]
        T = npatch + 1
        X = torch.empty((B, T, dim), dtype=torch.float32, device=x.device)
        X[:, 0] = self.cls_token[0, 0].detach() + pos[0]
        torch.add(tok.view(B, npatch, dim), pos[1:], out=X[:, 1:])
        return X.view(B * T, dim), B, T

    def _linear(self, lin, x, x_range, **kw):
        """x @ W^T + b on the fp16-pair kernels when the range of x is known: the (frozen) weight's range and its
        fp16-pair split are made once and kept until the parameter changes"""
        if not FP16_PAIRS or x_range is None:
            return ops.gemm_nt(x, lin.weight, lin.bias, **kw)
        w = lin.weight
        key = (w.data_ptr(), w._version)
        c = getattr(lin, "_pairs_cache", None)
        if c is None or c[0] != key:
            rw = ops.absmax(w.detach())
            c = lin._pairs_cache = (key, rw, ops.weight_pairs_2d(w.detach(), rw))
        return ops.gemm_nt(x, w, lin.bias, amax=(x_range, c[1]), b_pairs=c[2], **kw)

    def _attention(self, blk, h1, B, T, want_attn=False, keep_qkv=False, h1_range=None):
        """h1 [B*T, dim] (already normalised) -> (attention output [B*T, dim] before the projection, attention maps or
        None, the output's range or None)"""
        dim, nh = self.embed_dim, self.num_heads
        hd = dim // nh
        a = blk.attn
        r_qkv = ops.new_amax(h1.device) if FP16_PAIRS else None
        qkv = self._linear(a.qkv, h1, h1_range, amax_out=r_qkv)                                     # [B*T, 3 dim]
        if keep_qkv:
            self._last_qkv = qkv
        if not want_attn and hd == 64 and self.fused_attention:
            # scores never leave the chip (csrc/attention.hip); fp16-pair arithmetic with the range of qkv, which the
            # qkv GEMM's epilogue left behind
            r_o = ops.new_amax(h1.device) if FP16_PAIRS else None
            return ops.attention(qkv, B, T, nh, a.scale, amax=r_qkv, amax_out=r_o), None, r_o
        Tp = (T + 3) // 4 * 4
        out = torch.empty((B * T, dim), dtype=torch.float32, device=h1.device)
        attn = torch.empty((B, nh, T, T), dtype=torch.float32, device=h1.device) if want_attn else None
        # all heads of a chunk of images per launch (grid.y = image x head); the chunk bounds the [T, T] score buffers
        chunk = max(1, min(B, int(self.attn_bytes // (nh * T * Tp * 4))))
        S = torch.empty((chunk, nh, T, Tp), dtype=torch.float32, device=h1.device)
        for b0 in range(0, B, chunk):
            nb = min(chunk, B - b0)
            rows = qkv[b0 * T:(b0 + nb) * T]                                                        # [nb*T, 3 dim]
            ops.gemm_nt_batched(rows, 3 * dim, (T * 3 * dim, hd), rows[:, dim:], 3 * dim, (T * 3 * dim, hd),
                                S, Tp, (nh * T * Tp, T * Tp), (nb, nh), T, T, hd)                   # q k^T
            ops.softmax_rows_(S.view(-1, Tp)[:nb * nh * T], T, a.scale)                             # softmax(scale * .)
            if want_attn:
                attn[b0:b0 + nb].copy_(S[:nb, :, :, :T])
            Vt = torch.stack([ops.transpose2d(rows[i * T:(i + 1) * T, 2 * dim:], Tp) for i in range(nb)])   # [nb, dim, Tp]
            ops.gemm_nt_batched(S, Tp, (nh * T * Tp, T * Tp), Vt, Tp, (dim * Tp, hd * Tp), out[b0 * T:], dim,
                                (T * dim, hd), (nb, nh), T, hd, Tp)                                 # attn v
        return out, attn, None

    def _block(self, blk, X, B, T, last=False, want_attn=False):
        # every GEMM input's range comes out of the kernel that produced it (LayerNorm, attention, the GELU epilogue)
        new = (lambda: ops.new_amax(X.device)) if FP16_PAIRS else (lambda: None)
        r1 = new()
        h1 = ops.layernorm(X, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, amax_out=r1)
        o, attn, r_o = self._attention(blk, h1, B, T, want_attn=want_attn, keep_qkv=last, h1_range=r1)
        if want_attn:
            return attn
        self._linear(blk.attn.proj, o, r_o, out=X, beta=1)                                          # x = x + proj(.)
        r2, rm = new(), new()
        h2 = ops.layernorm(X, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, amax_out=r2)
        m = self._linear(blk.mlp.fc1, h2, r2, act=2, amax_out=rm)                                   # GELU in the epilogue
        self._linear(blk.mlp.fc2, m, rm, out=X, beta=1)                                             # x = x + mlp(.)
        return X

    # ---------------------------------------------------------------- reference surfaces
    @torch.no_grad()
    def forward(self, x):
        X, B, T = self.prepare_tokens(x)
        for i, blk in enumerate(self.blocks):
            self._block(blk, X, B, T, last=(i == len(self.blocks) - 1))
        return ops.layernorm(X, self.norm.weight, self.norm.bias, self.norm.eps).view(B, T, -1)

    @torch.no_grad()
    def get_intermediate_layers(self, x, n=1):
        X, B, T = self.prepare_tokens(x)
        outs = []
        for i, blk in enumerate(self.blocks):
            self._block(blk, X, B, T, last=(i == len(self.blocks) - 1))
            if len(self.blocks) - i <= n:
                outs.append(ops.layernorm(X, self.norm.weight, self.norm.bias, self.norm.eps).view(B, T, -1))
        return outs

    @torch.no_grad()
    def get_last_selfattention(self, x):
        """attention maps [B, heads, T, T] of the last block (models/dino_vit.py:260-267); also keeps its qkv"""
        X, B, T = self.prepare_tokens(x)
        for blk in self.blocks[:-1]:
            self._block(blk, X, B, T)
        return self._block(self.blocks[-1], X, B, T, last=True, want_attn=True)

    @torch.no_grad()
    def get_last_qkv(self, x, which="k"):
        """q / k / v of the last block as [B, T, dim] (heads side by side), without materialising its attention maps --
        the features the reference's NCut heads take from a forward hook (maa.py:63-68,90-118)"""
        X, B, T = self.prepare_tokens(x)
        for blk in self.blocks[:-1]:
            self._block(blk, X, B, T)
        last = self.blocks[-1]
        h1 = ops.layernorm(X, last.norm1.weight, last.norm1.bias, last.norm1.eps)
        qkv = ops.gemm_nt(h1, last.attn.qkv.weight, last.attn.qkv.bias)
        i = "qkv".index(which)
        return qkv[:, i * self.embed_dim:(i + 1) * self.embed_dim].reshape(B, T, self.embed_dim)


def vit_tiny(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4, qkv_bias=True, **kw)


def vit_small(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True, **kw)


def vit_base(patch_size=16, **kw):
    return VisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True, **kw)
