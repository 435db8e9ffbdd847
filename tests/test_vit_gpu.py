"""GPU: DINO ViT-S/8 forward and soft-NCut refinement against vectors captured from the
reference (tests/golden/vit_small8.npz, written by make_golden_vit.py): tokens, the last two normalised layers, the
last block's attention maps and K features; the NCut value and the 10-step refined mask.  Tolerance 1e-4 relative."""
import os

import numpy as np
import pytest
import torch

import rcf_amd
from rcf_amd import ncut, ops, synth, vit

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _model(fx):
    m = vit.vit_small(patch_size=8)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_vit_state_dict(shapes, seed=int(fx["weight_seed"])).items()})
    return m.to(DEV).eval()


def test_gemm_layernorm_softmax_helpers(report):
    g = torch.Generator().manual_seed(1)
    a = torch.randn(97, 1152, generator=g).to(DEV)
    q, k = a[:, 64:128], a[:, 448:512]                                       # pitched 64-wide slices, like q / k of a head
    s = ops.gemm_nt(q, k)
    e_g = rel(s.cpu(), (q.double() @ k.double().T).cpu())
    w, b = torch.randn(200, 1152, generator=g).to(DEV), torch.randn(200, generator=g).to(DEV)
    y = ops.gemm_nt(a, w, b, act=2)
    e_gelu = rel(y.cpu(), torch.nn.functional.gelu(a.double() @ w.double().T + b.double()).cpu())
    x = torch.randn(50, 384, generator=g).to(DEV)
    gam, bet = torch.randn(384, generator=g).to(DEV), torch.randn(384, generator=g).to(DEV)
    e_ln = rel(ops.layernorm(x, gam, bet, 1e-6).cpu(),
               torch.nn.functional.layer_norm(x.double(), (384,), gam.double(), bet.double(), 1e-6).cpu())
    S = torch.randn(97, 100, generator=g).to(DEV)
    ref = torch.softmax(S[:, :97].double() * 0.125, dim=-1)
    ops.softmax_rows_(S, 97, 0.125)
    e_sm = rel(S[:, :97].cpu(), ref.cpu())
    pad0 = float(S[:, 97:].abs().max())
    vt = ops.transpose2d(a[:, 800:864], 100)
    ok_t = bool(torch.equal(vt[:, :97], a[:, 800:864].T.contiguous())) and float(vt[:, 97:].abs().max()) == 0.0
    report(f"vit helpers: gemm {e_g:.2e} gemm+gelu {e_gelu:.2e} layernorm {e_ln:.2e} softmax {e_sm:.2e} transpose {ok_t}")
    assert max(e_g, e_gelu, e_ln, e_sm) < 2e-5 and pad0 == 0.0 and ok_t


def test_linear_fp16_pairs_with_fused_ranges(report):
    """nn.Linear on the fp16-pair kernels: LayerNorm leaves the range of its output, the GEMM takes it together with
    the weight's range and pre-split weights, and its GELU epilogue leaves the range of what it wrote"""
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(301, 384, generator=g) * 3).to(DEV)
    gam, bet = torch.randn(384, generator=g).to(DEV), torch.randn(384, generator=g).to(DEV)
    w, b = (torch.randn(1536, 384, generator=g) * 0.05).to(DEV), torch.randn(1536, generator=g).to(DEV)
    r1, rm = ops.new_amax(DEV), ops.new_amax(DEV)
    h = ops.layernorm(x, gam, bet, 1e-6, amax_out=r1)
    assert float(r1.view(torch.float32)) == float(h.abs().max())
    rw = ops.absmax(w)
    y = ops.gemm_nt(h, w, b, act=2, amax=(r1, rw), b_pairs=ops.weight_pairs_2d(w, rw), amax_out=rm)
    assert float(rm.view(torch.float32)) == float(y.abs().max())
    assert torch.equal(y, ops.gemm_nt(h, w, b, act=2, amax=(r1, rw)))             # pre-split weights: the same bits
    ref = torch.nn.functional.gelu(h.double() @ w.double().T + b.double())
    e2, e3 = rel(y.cpu(), ref.cpu()), rel(ops.gemm_nt(h, w, b, act=2).cpu(), ref.cpu())
    report(f"linear + GELU 384->1536: fp16 pairs {e2:.2e}, bf16 triples {e3:.2e} (max error / max |ref| vs float64)")
    assert e2 < 2e-6 and e3 < 2e-6


@pytest.mark.parametrize("pairs,qscale", [(False, 1.5), (True, 1.5), (True, 3e-4), (True, 2e3)])
@pytest.mark.parametrize("T", [97, 130, 257])
def test_fused_attention_vs_float64(T, pairs, qscale, report):
    """csrc/attention.hip against float64 softmax(q k^T * scale) v on random qkv (tails: T not a multiple of the 64-key /
    128-query tiles); bf16 triples, and fp16 pairs with operand magnitudes from 3e-4 to 2e3"""
    g = torch.Generator().manual_seed(T)
    B, nh, dim = 2, 6, 384
    qkv = (torch.randn(B * T, 3 * dim, generator=g) * qscale).to(DEV)
    sc = 0.125 * (1.5 / qscale) ** 2                      # keeps the logits' spread the same
    ro = ops.new_amax(DEV)
    out = ops.attention(qkv, B, T, nh, sc, amax=ops.absmax(qkv) if pairs else None, amax_out=ro)
    assert float(ro.view(torch.float32)) == float(out.abs().max())
    out = out.cpu().double()
    q, k, v = (qkv.cpu().double().view(B, T, 3, nh, 64).permute(2, 0, 3, 1, 4)[i] for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-2, -1) * sc, dim=-1) @ v).transpose(1, 2).reshape(B * T, dim)
    e = float((out - ref).abs().max() / ref.abs().max())
    report(f"fused attention T={T} {'fp16 pairs' if pairs else 'bf16 triples'} |qkv|~{qscale:g}: {e:.2e}")
    assert e < 2e-5


def test_vit_small8_vs_reference_golden(golden_dir, report):
    fx = np.load(os.path.join(golden_dir, "vit_small8.npz"))
    m = _model(fx)
    x = torch.from_numpy(fx["img"]).to(DEV)
    tokens = m(x)
    inter = m.get_intermediate_layers(x, n=2)
    attn = m.get_last_selfattention(x)
    k = m.get_last_qkv(x, "k")
    e = {"tokens": rel(tokens.cpu(), fx["tokens"]), "inter[-2]": rel(inter[0].cpu(), fx["inter0"]),
         "attn": rel(attn.cpu(), fx["attn_last"]), "k_last": rel(k.cpu(), fx["k_last"])}
    report("ViT-S/8 vs reference: " + " ".join(f"{a} {b:.2e}" for a, b in e.items()))
    assert max(e.values()) < TOL, e


def test_soft_ncut_vs_reference_golden(golden_dir, report):
    fx = np.load(os.path.join(golden_dir, "vit_small8.npz"))
    feats = torch.from_numpy(fx["ncut_feats"]).to(DEV)
    mask = torch.from_numpy(fx["mask"]).to(DEV)
    v0 = float(ncut.soft_ncut_value(feats, mask, 0.2, 1e-5))
    refined = ncut.ncut_refine(feats, mask, tau=0.2, eps=1e-5, steps=10, learning_rate=0.45, weight_decay=1e-6)
    e_v = abs(v0 - float(fx["ncut0"])) / float(fx["ncut0"])
    d = np.abs(refined.cpu().numpy() - fx["refined"])
    report(f"soft NCut vs reference: value {e_v:.2e}, refined mask max |d| {d.max():.2e} (cells off by > 1e-3: {(d > 1e-3).sum()})")
    # Adam's first steps are sign-like (g / (|g| + 1e-8)); with lr 0.45 and the clamp the mask saturates, so the
    # comparison is on the final mask: every cell within 1e-3
    assert e_v < 1e-5 and d.max() < 1e-3


def test_vit_small8_fullsize_vs_float64(golden_dir, report):
    """480 x 856 (6 421 tokens: the size the semantic-constraint tools run, models/dino_vit.py on a 480p frame): the twelve
    blocks and the final norm of the HIP model against the same arithmetic in float64 torch on the prepared tokens
    (models/dino_vit.py:110-167: pre-norm attention + MLP blocks, softmax(q k^T / 8), GELU(erf), LayerNorm eps 1e-6).  The
    reference fixture pins the small size end to end; this pins the attention / linear kernels at the real token count."""
    fx = np.load(os.path.join(golden_dir, "vit_small8.npz"))
    m = _model(fx)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 3, 480, 856, generator=g).to(DEV)
    X0, B, T = m.prepare_tokens(x)
    assert T == 60 * 107 + 1
    out = m(x)[0].double()
    torch.set_grad_enabled(False)                          # the float64 mirror reads the module's parameters
    X = X0.clone().double()
    F = torch.nn.functional
    nh = m.blocks[0].attn.num_heads
    for blk in m.blocks:
        h = F.layer_norm(X, (X.shape[1],), blk.norm1.weight.double(), blk.norm1.bias.double(), blk.norm1.eps)
        qkv = (h @ blk.attn.qkv.weight.double().T + blk.attn.qkv.bias.double()).view(T, 3, nh, -1).permute(1, 2, 0, 3)
        a = torch.softmax(qkv[0] @ qkv[1].transpose(-2, -1) * (qkv.shape[-1] ** -0.5), dim=-1) @ qkv[2]      # [nh, T, 64]
        X = X + a.transpose(0, 1).reshape(T, -1) @ blk.attn.proj.weight.double().T + blk.attn.proj.bias.double()
        h = F.layer_norm(X, (X.shape[1],), blk.norm2.weight.double(), blk.norm2.bias.double(), blk.norm2.eps)
        X = X + F.gelu(h @ blk.mlp.fc1.weight.double().T + blk.mlp.fc1.bias.double()) @ blk.mlp.fc2.weight.double().T + blk.mlp.fc2.bias.double()
    ref = F.layer_norm(X, (X.shape[1],), m.norm.weight.double(), m.norm.bias.double(), m.norm.eps)
    torch.set_grad_enabled(True)
    e = float((out - ref).abs().max() / ref.abs().max())
    report(f"ViT-S/8 at 480x856 ({T} tokens), 12 blocks vs float64: max error / max |ref| {e:.2e}")
    assert e < TOL
