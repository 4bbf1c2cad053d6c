"""RGB layer3 bottleneck (256 mid channels, 16 x 16 maps): fused tail + next reduction (bneck231r_kernel<..., 256, 256>) vs the three launches.
usage: python tools/bneck256_bench.py [B H]   (B = 128: the hi|lo pair of a 64-environment batch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
CODE, tdt = _lib.HCM_F16, torch.float16
P = lambda t: t.data_ptr()
C1, C3, CN = 256, 1024, 256
x = torch.randn(B, H, H, C1, device="cuda").to(tdt)
w2 = (torch.randn(C1, 3, 3, C1, device="cuda") * 0.03).to(tdt); b2 = torch.randn(C1, device="cuda")
w3 = (torch.randn(C3, 1, 1, C1, device="cuda") * 0.05).to(tdt); b3 = torch.randn(C3, device="cuda")
w1 = (torch.randn(CN, 1, 1, C3, device="cuda") * 0.03).to(tdt); b1 = torch.randn(CN, device="cuda")
r = torch.randn(B, H, H, C3, device="cuda").to(tdt); y = torch.empty_like(r); y2 = torch.empty_like(r)
mid = torch.empty(B, H, H, C1, device="cuda", dtype=tdt)
o1 = torch.empty(B, H, H, CN, device="cuda", dtype=tdt); o2 = torch.empty_like(o1)
def c2(): return lib.hcm_op_conv2d(P(x), P(w2), P(b2), None, P(mid), CODE, B, H, H, C1, C1, 3, 3, 1, 1, 1, None)
def c3(): return lib.hcm_op_conv2d(P(mid), P(w3), P(b3), P(r), P(y2), CODE, B, H, H, C1, C3, 1, 1, 1, 0, 1, None)
def c1n(): return lib.hcm_op_conv2d(P(y2), P(w1), P(b1), None, P(o2), CODE, B, H, H, C3, CN, 1, 1, 1, 0, 1, None)
def three():
    c2(); c3(); return c1n()
def fused(): return lib.hcm_op_bottleneck_tail_next(P(x), P(w2), P(b2), P(w3), P(b3), P(r), P(y), P(w1), P(b1), P(o1), CODE, B, H, H, C1, 1, CN, None)
assert three() == 0 and fused() == 0
torch.cuda.synchronize()
print("bit-identical:", torch.equal(y.view(torch.int16), y2.view(torch.int16)), torch.equal(o1.view(torch.int16), o2.view(torch.int16)))
M = B * H * H
gf = 2.0 * M * (9 * C1 * C1 + C1 * C3 + C3 * CN) / 1e9
for name, fn in (("3x3 conv", c2), ("expansion + identity", c3), ("next reduction", c1n), ("three launches", three), ("fused", fused)):
    for _ in range(50): assert fn() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    print(f"B={B} 256ch @{H}: {name:22s} {us:7.1f} us" + (f"  ({gf / us * 1e-3 * 1e3:.0f} TFLOP/s)" if name in ("three launches", "fused") else ""))
