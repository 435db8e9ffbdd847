#!/usr/bin/env python3
"""Golden vectors for the DINO ViT-S/8 forward and the soft-NCut refinement, captured from the
REFERENCE: models/dino_vit.py (loaded from its file: it needs nothing but torch) and the `soft_ncut_value` /
`ncut_refine` functions of tools/SemanticConstraintsAndMAA/semantic_constraints.py (extracted with `ast`, because the
script itself imports matplotlib / the CUDA CRF).  Run in the build container only:
    python tests/golden/make_golden_vit.py
"""
import ast
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def main():
    import rcf_amd                                           # noqa
    from rcf_amd import synth
    spec = importlib.util.spec_from_file_location("ref_dino_vit", os.path.join(REF, "models", "dino_vit.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    tree = ast.parse(open(os.path.join(REF, "tools", "SemanticConstraintsAndMAA", "semantic_constraints.py")).read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("soft_ncut_value", "ncut_refine")]
    ns = {"torch": torch, "F": F, "plt": None}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "semantic_constraints.py", "exec"), ns)

    torch.manual_seed(0)
    torch.set_num_threads(8)
    W_, H_ = 64, 96                                           # the reference names the two spatial dims (w, h)
    model = ref.vit_small(patch_size=8).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_vit_state_dict(shapes, seed=21).items()})
    g = np.random.Generator(np.random.PCG64(5))
    img = g.standard_normal((2, 3, W_, H_)).astype(np.float32)
    x = torch.from_numpy(img)
    feat = {}
    model.blocks[-1].attn.qkv.register_forward_hook(lambda m, i, o: feat.__setitem__("qkv", o))
    with torch.no_grad():
        tokens = model(x)
        attn = model.get_last_selfattention(x)
        inter = model.get_intermediate_layers(x, n=2)
    B, T = tokens.shape[:2]
    k = feat["qkv"].reshape(B, T, 3, 6, -1).permute(2, 0, 3, 1, 4)[1].transpose(1, 2).reshape(B, T, -1)
    # soft NCut (tau 0.2, eps 1e-5; NCutHead: 10 Adam steps, lr 0.45, wd 1e-6).  A random-init ViT gives nearly
    # identical tokens (affinity all ones, NCut == 1, zero gradient), so the fixture uses clustered synthetic
    # features of the same shape: 3 clusters + noise, intra-cluster cosine > tau > inter-cluster cosine.
    hf, wf = W_ // 8, H_ // 8
    centers = g.standard_normal((3, 384))
    lab = (np.arange(hf * wf) * 3 // (hf * wf) + (g.random(hf * wf) > 0.85)) % 3
    fe = centers[lab] + 1.2 * g.standard_normal((hf * wf, 384))
    feats = torch.from_numpy(np.concatenate([g.standard_normal((1, 384)), fe])[None].astype(np.float32))   # + [CLS] row
    mask = torch.from_numpy((g.random((hf, wf)) > 0.5).astype(np.float32) * 0.8 + 0.1)
    with torch.no_grad():
        ncut0 = ns["soft_ncut_value"](feats, mask, 0.2, 1e-5)
        fn = F.normalize(feats[0, 1:], p=2)
        dens = float(((fn @ fn.T) > 0.2).float().mean())
    refined = ns["ncut_refine"](feats, mask, tau=0.2, eps=1e-5, steps=10, learning_rate=0.45, weight_decay=1e-6)
    np.savez_compressed(os.path.join(HERE, "vit_small8.npz"), img=img, weight_seed=21,
                        schema=np.array([f"{k}:{'x'.join(map(str, v))}" for k, v in shapes.items()]), tokens=tokens.numpy(),
                        attn_last=attn.numpy().astype(np.float32), inter0=inter[0].numpy(), k_last=k.numpy(),
                        ncut_feats=feats.numpy(), mask=mask.numpy(), ncut0=np.float64(float(ncut0)), refined=refined.numpy())
    print("tokens", tuple(tokens.shape), "attn", tuple(attn.shape), "affinity density", dens, "ncut0", float(ncut0),
          "refined mean", float(refined.mean()), "moved", float((refined - mask).abs().mean()))


if __name__ == "__main__":
    main()
