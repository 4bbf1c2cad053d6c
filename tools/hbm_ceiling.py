"""What HBM rate do read:write mixes like the ResNet expansion convs see from trivially coalesced torch kernels?
(ceiling for the write-dominated 1x1 convs of layer1/layer2).  usage: python tools/hbm_ceiling.py"""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
MB = 1 << 20
a = torch.empty(268 * MB // 2, dtype=torch.bfloat16, device="cuda").normal_()
b = torch.empty_like(a)
small = torch.empty(67 * MB // 2, dtype=torch.bfloat16, device="cuda").normal_()
us = t(lambda: b.fill_(1.0)); print(f"pure write 268 MB: {us:.1f} us  {268 * MB / us / 1e6:.2f} TB/s")
us = t(lambda: b.copy_(a)); print(f"copy 268 MB -> 268 MB: {us:.1f} us  {2 * 268 * MB / us / 1e6:.2f} TB/s")
us = t(lambda: torch.add(a, a, out=b)); print(f"read 268 (x2 same) write 268: {us:.1f} us  {2 * 268 * MB / us / 1e6:.2f} TB/s (algorithmic 2 tensors)")
v = b.view(4, -1)
us = t(lambda: v.copy_(small.view(1, -1).expand(4, -1))); print(f"read 67 MB write 268 MB (broadcast): {us:.1f} us  {(67 + 268) * MB / us / 1e6:.2f} TB/s")
c = torch.empty_like(a)
us = t(lambda: torch.add(a, c, out=b)); print(f"read 2 x 268 write 268 (residual-like): {us:.1f} us  {3 * 268 * MB / us / 1e6:.2f} TB/s")
us = t(lambda: torch.relu_(a)); print(f"in-place 268 MB r+w: {us:.1f} us  {2 * 268 * MB / us / 1e6:.2f} TB/s")
