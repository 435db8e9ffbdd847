"""DINO ViT-S/8 forward at 480x856 (6421 tokens) and the soft-NCut refinement:
ms per frame and TF/s on the algorithmic 1.03 TF per frame (12 blocks x 86 GF)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rcf_amd
from rcf_amd import ncut, synth, vit
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m = vit.vit_small(patch_size=8)
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_vit_state_dict(shapes, seed=21).items()})
m = m.to("cuda:0").eval()
x = torch.randn(B, 3, 480, 856, device="cuda:0")
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n
T = 60 * 107 + 1
blk = 2 * T * (384 * 1152 + 384 * 384 + 2 * 384 * 1536) + 2 * 2 * T * T * 384
flops = 12 * blk + 2 * 60 * 107 * 384 * 192
t = timeit(lambda: m.get_last_qkv(x, "k"))
print(f"ViT-S/8 get_last_qkv (11 blocks + qkv) B={B}: {t*1e3/B:.1f} ms/frame  {flops*11.3/12/t*B/1e12:.1f} TF/s")
t = timeit(lambda: m(x))
print(f"ViT-S/8 forward (12 blocks)      B={B}: {t*1e3/B:.1f} ms/frame  {flops/t*B/1e12:.1f} TF/s  ({flops/1e9:.0f} GF/frame)")
feats = m.get_last_qkv(x[:1], "k")
mask = (torch.rand(60, 107, device="cuda:0") > 0.5).float() * 0.8 + 0.1
t = timeit(lambda: ncut.ncut_refine(feats, mask, steps=10, learning_rate=0.45))
print(f"soft NCut refine (affinity 6420^2 + 10 Adam steps): {t*1e3:.2f} ms/frame")
t = timeit(lambda: ncut.soft_ncut_value(feats, mask, 0.2, 1e-5))
print(f"soft NCut value: {t*1e3:.2f} ms/frame")
