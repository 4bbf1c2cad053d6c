import torch
def bench(M,N,K,dt=torch.bfloat16):
    a=torch.randn(M,K,device='cuda',dtype=dt); b=torch.randn(N,K,device='cuda',dtype=dt)
    for _ in range(5): c=a@b.t()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): c=a@b.t()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/50*1e3
    print(f"torch mm {M}x{N}x{K}: {us:.1f} us  {2*M*N*K/us/1e6:.0f} TF")
for s in [(5120,3072,768),(5120,768,3072),(5120,2304,768),(5120,768,768),(8192,8192,8192),(16384,256,2304),(65536,128,1152),(16384,1024,256)]:
    bench(*s)
