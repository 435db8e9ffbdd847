"""debug: sampled-gradient error vs the float64 truth (tests/golden/rcf_small.npz) when every parameter is moved by one
unit in the last place (random signs): how much of the per-element gradient error is conditioning, per realization"""
import copy, os, sys
import numpy as np, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import rcf_amd
import test_model_gpu as T

fx = np.load(os.path.join(R, "tests", "golden", "rcf_small.npz"))
H, W = 96, 160
for seed in range(0, 9):
    model = T._build(H, W, False, "cuda:0", rcf_amd.RCFModel)
    if seed:
        g = torch.Generator().manual_seed(1000 + seed)
        sd = model.state_dict()
        for k, v in sd.items():
            if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")):
                s = (torch.randint(0, 2, v.shape, generator=g).float() * 2 - 1).to(v.device)
                v.mul_(1 + s * 2.0 ** -23)
    tr = rcf_amd.Trainer(model, lr=1e-4, weight_decay=1e-4, device="cuda:0")
    tr.step(T._batch(int(fx["B"]), H, W, "cuda:0"))
    named = dict(model.named_parameters())
    e = [T.rel(named[str(n)].grad.detach().cpu().contiguous().numpy().ravel()[:256], fx[f"truth_grad_{i}"]) for i, n in enumerate(fx["sampled"])]
    print("ulp seed", seed, " ".join(f"{v:.2e}" for v in e), flush=True)
print("limits (4 x ref32)", " ".join(f"{4*v:.2e}" for v in fx["ref32_err_grad"]))
