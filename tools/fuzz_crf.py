"""Random CRF calls (sizes 5 ... 260, 1 ... 5 frames, natural / noisy / mixed content, one or two potentials, both normalisations): the list
walk, the tile splat (default rule and forced), the fused and the separate slice, the small-table overflow path and the sort build must
give identical MAPs, marginals and vertex counts.  usage: python tools/fuzz_crf.py [cases=60] [seed=0]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rcf_amd  # noqa
from rcf_amd import synth
from rcf_amd.crf import crf_soft_batched

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
GATHER, TILES, SEPARATE, SMALL, SORT = 0x4000 >> 8, 0x8000 >> 8, 0x10000 >> 8, 2, 3
bad = 0
for c in range(cases):
    H, W, F = int(rng.integers(5, 261)), int(rng.integers(5, 261)), int(rng.integers(1, 6))
    T = int(rng.integers(0, 4))
    amp = int(rng.choice([0, 0, 3, 10, 40, 255]))
    frames = []
    for i in range(F):
        base = synth.smooth_rgb(H, W, 9000 + 10 * c + i).astype(np.int32)
        frames.append(np.clip(base + rng.integers(-amp, amp + 1, base.shape), 0, 255).astype(np.uint8))
    rgb = torch.from_numpy(np.stack(frames)).cuda()
    m = np.clip(np.stack([synth.soft_blob_mask(H, W, 9000 + 10 * c + i) for i in range(F)]), 1e-4, 1 - 1e-4).reshape(F, -1)
    un = torch.from_numpy(np.stack([-np.log(1 - m), -np.log(m)], axis=2).astype(np.float32)).cuda()
    two = bool(rng.integers(0, 3) == 0)
    sym = bool(not two and rng.integers(0, 3) == 0)
    params = (3.0, 3.0, 5.0, 60.0, 5.0) if two else (0.0, 0.0, float(rng.choice([5.0, 10.0])), float(rng.choice([20.0, 60.0])), float(rng.choice([5.0, 20.0])))
    ref = None
    for name, fl in (("gather", GATHER), ("default", 0), ("tiles", TILES), ("tiles separate", TILES | SEPARATE), ("tiles overflow", TILES | SMALL), ("sort", SORT)):
        r = crf_soft_batched(rgb, un, W, H, *params, T, want_q=True, want_nvert=True, symmetric=sym, build=fl)
        if ref is None:
            ref = r
        elif not all(bool(torch.equal(a, b)) for a, b in zip(ref, r)):
            bad += 1
            print(f"MISMATCH case {c}: {H}x{W} F={F} T={T} amp={amp} params={params} sym={sym}: {name}")
print(f"{cases} cases, {bad} mismatches")
