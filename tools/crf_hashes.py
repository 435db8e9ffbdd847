"""SHA-256 of the CRF's MAP / marginals / vertex counts over a fixed set of calls (natural and noise frames, list walk and tile splat,
two potentials, symmetric normalisation, T = 1 / 5 / 12) -- to compare two builds of the library bit for bit.
usage: python tools/crf_hashes.py <tag>  ->  gpurun_out/crf_hashes_<tag>.json"""
import hashlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rcf_amd  # noqa
from rcf_amd import synth
from rcf_amd.crf import crf_soft_batched

tag = sys.argv[1] if len(sys.argv) > 1 else "x"
GATHER, TILES = 0x4000 >> 8, 0x8000 >> 8


def unary(m):
    m = np.clip(m, 1e-4, 1 - 1e-4).reshape(-1)
    return np.stack([-np.log(1 - m), -np.log(m)], axis=1).astype(np.float32)


out = {}
for kind, (H, W), F in (("smooth", (480, 854), 3), ("smooth", (97, 131), 3), ("noise", (120, 214), 2), ("mixed", (200, 320), 4)):
    frames = []
    for i in range(F):
        gen = synth.noise_rgb if (kind == "noise" or (kind == "mixed" and i % 2)) else synth.smooth_rgb
        frames.append(gen(H, W, 5100 + i))
    rgb = torch.from_numpy(np.stack(frames)).cuda()
    un = torch.from_numpy(np.stack([unary(synth.soft_blob_mask(H, W, 5100 + i)) for i in range(F)])).cuda()
    for params, sym in (((0., 0., 10., 60., 20.), False), ((3., 3., 5., 60., 5.), False), ((0., 0., 5., 60., 5.), True)):
        for iters in (1, 5, 12):
            for mode, fl in (("gather", GATHER), ("default", 0), ("tiles", TILES), ("sort", 3)):
                r = crf_soft_batched(rgb, un, W, H, *params, iters, want_q=True, want_nvert=True, symmetric=sym, build=fl)
                h = hashlib.sha256()
                for t in r:
                    h.update(t.cpu().numpy().tobytes())
                out[f"{kind} {H}x{W} {params} sym={sym} T={iters} {mode}"] = h.hexdigest()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/crf_hashes_{tag}.json", "w"), indent=0)
modes = {}
for k, v in out.items():
    modes.setdefault(k.rsplit(" ", 1)[0], set()).add(v)
print(f"{len(out)} calls hashed; configurations whose modes disagree: {sum(1 for v in modes.values() if len(v) > 1)}")
