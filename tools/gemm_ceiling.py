import torch, time
dev="cuda:0"
def t(fn,n=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e-3
for (M,K,N) in [(102720,4608,512),(102720,512,2048),(102720,2048,512),(102720,2304,256),(102720,256,1024),(102720,1024,256),(102720,18432,256),(410880,2304,256),(8192,8192,8192)]:
    a=torch.randn(M,K,device=dev,dtype=torch.bfloat16); b=torch.randn(N,K,device=dev,dtype=torch.bfloat16)
    dt=t(lambda: torch.matmul(a,b.t()))
    print(f"M={M} K={K} N={N}: {dt*1e3:.3f} ms {2*M*K*N/dt/1e12:.0f} TF/s")
