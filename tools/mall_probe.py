"""Does the 256 MiB Infinity Cache speed up a producer -> consumer chain whose tensors fit in it?  Ping-pong b = a + 1, a = b + 1
over tensors of S MB each (trivially coalesced torch kernels); reports the algorithmic rate 2 S / t per launch.
usage: python tools/mall_probe.py"""
import torch
MB = 1 << 20
for S in (8, 16, 32, 48, 64, 96, 128, 192, 268, 536):
    a = torch.empty(S * MB // 2, dtype=torch.bfloat16, device="cuda").normal_()
    b = torch.empty_like(a)
    for _ in range(20):
        torch.add(a, 1.0, out=b); torch.add(b, 1.0, out=a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        torch.add(a, 1.0, out=b); torch.add(b, 1.0, out=a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (2 * n) * 1e3
    print(f"S = {S:4d} MB per tensor (working set {2 * S} MB): {us:7.1f} us per launch, {2 * S * MB / us / 1e6:.2f} TB/s")
