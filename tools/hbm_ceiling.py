"""What a plain streaming kernel reaches on this part (calibration for the HBM-bound kernels): torch element-wise ops on
tensors far larger than the caches"""
import torch
dev = "cuda:0"
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for dt in (torch.float32, torch.bfloat16):
    for mb in (210, 840, 3360):
        n = mb * 2**20 // torch.tensor([], dtype=dt).element_size()
        x = torch.randn(n, device=dev).to(dt); y = torch.empty_like(x); z = torch.randn(n, device=dev).to(dt)
        b = x.numel() * x.element_size()
        t_copy = t(lambda: y.copy_(x)); t_add = t(lambda: torch.add(x, z, out=y)); t_sum = t(lambda: x.sum())
        print(f"{dt} {mb} MB: copy {2*b/t_copy/1e12:.2f} TB/s | add (2 reads + 1 write) {3*b/t_add/1e12:.2f} TB/s | sum (read only) {b/t_sum/1e12:.2f} TB/s")
