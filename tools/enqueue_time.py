"""how long the host takes to enqueue one training step vs how long the GPU takes to run it"""
import os, sys, time, types, copy
import numpy as np, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import rcf_amd
import test_model_gpu as T
H, W, B = 480, 854, 8
model = T._build(H, W, False, "cuda:0", rcf_amd.RCFModel)
tr = rcf_amd.Trainer(model, lr=1e-4, weight_decay=1e-4, device="cuda:0")
batch = T._batch(B, H, W, "cuda:0")
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    tr.step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms, step complete {1e3*(t2-t0):.1f} ms")
