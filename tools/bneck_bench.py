"""Fused bottleneck tail (3x3 conv -> 1x1 expansion + identity, one launch) vs the two stand-alone conv launches, bf16.
usage: python tools/bneck_bench.py [B H C1 stride]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
cases = [[int(v) for v in sys.argv[1:5]]] if len(sys.argv) >= 5 else [[128, 64, 64, 1], [128, 32, 128, 1], [128, 64, 128, 2]]
CODE, tdt = (CODE, torch.bfloat16) if os.environ.get("BNECK_DT") == "bf16" else (_lib.HCM_F16, torch.float16)
P = lambda t: t.data_ptr()
for B, H, C1, stride in cases:
    C3 = 4 * C1
    Ho = (H + 2 - 3) // stride + 1
    x = torch.randn(B, H, H, C1, device="cuda").to(tdt)
    w2 = (torch.randn(C1, 3, 3, C1, device="cuda") * 0.05).to(tdt); b2 = torch.randn(C1, device="cuda")
    w3 = (torch.randn(C3, 1, 1, C1, device="cuda") * 0.05).to(tdt); b3 = torch.randn(C3, device="cuda")
    r = torch.randn(B, Ho, Ho, C3, device="cuda").to(tdt); y = torch.empty_like(r); mid = torch.empty(B, Ho, Ho, C1, device="cuda", dtype=tdt)
    fused = lambda: lib.hcm_op_bottleneck_tail(P(x), P(w2), P(b2), P(w3), P(b3), P(r), P(y), CODE, B, H, H, C1, stride, None)
    def two():
        lib.hcm_op_conv2d(P(x), P(w2), P(b2), None, P(mid), CODE, B, H, H, C1, C1, 3, 3, stride, 1, 1, None)
        return lib.hcm_op_conv2d(P(mid), P(w3), P(b3), P(r), P(y), CODE, B, Ho, Ho, C1, C3, 1, 1, 1, 0, 1, None)
    CN = int(os.environ.get('BNECK_CN', C1))
    w1 = (torch.randn(CN, 1, 1, C3, device="cuda") * 0.05).to(tdt); b1 = torch.randn(CN, device="cuda"); o1 = torch.empty(B, Ho, Ho, CN, device="cuda", dtype=tdt)
    fused3 = lambda: lib.hcm_op_bottleneck_tail_next(P(x), P(w2), P(b2), P(w3), P(b3), P(r), P(y), P(w1), P(b1), P(o1), CODE, B, H, H, C1, stride, CN, None)
    def three():
        fused()
        return lib.hcm_op_conv2d(P(y), P(w1), P(b1), None, P(o1), CODE, B, Ho, Ho, C3, CN, 1, 1, 1, 0, 1, None)
    for name, fn in (("two launches", two), ("fused", fused), ("fused + next c1", three), ("fused incl. c1", fused3)):
        for _ in range(100): assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} {C1}ch @{H} stride {stride}: {name:13s} {e0.elapsed_time(e1) / 200 * 1e3:7.1f} us")
