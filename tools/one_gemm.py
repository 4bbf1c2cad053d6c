"""Run one igemm shape N times (for rocprofv3 --pmc / kernel-trace probes).
usage: one_gemm.py lin M N K [reps] | conv H Cin Cout K stride pad [reps]   (B=64, bf16)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
tdt = torch.bfloat16
a = sys.argv[1:]
st = None
if a[0] == "lin":
    M, N, K = int(a[1]), int(a[2]), int(a[3]); reps = int(a[4]) if len(a) > 4 else 20
    x = torch.randn(M, K, device="cuda").to(tdt); w = (torch.randn(N, K, device="cuda") * 0.05).to(tdt); b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=tdt)
    run = lambda: lib.hcm_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), _lib.HCM_BF16, M, N, K, 0, 0, st)
else:
    H, Cin, Cout, K, stride, pad = [int(v) for v in a[1:7]]; reps = int(a[7]) if len(a) > 7 else 20
    B = 64; Ho = (H + 2 * pad - K) // stride + 1
    x = torch.randn(B, H, H, Cin, device="cuda").to(tdt); w = (torch.randn(Cout, K, K, Cin, device="cuda") * 0.05).to(tdt); b = torch.randn(Cout, device="cuda")
    y = torch.empty(B, Ho, Ho, Cout, device="cuda", dtype=tdt)
    run = lambda: lib.hcm_op_conv2d(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), _lib.HCM_BF16, B, H, H, Cin, Cout, K, K, stride, pad, 1, st)
for _ in range(reps):
    assert run() == 0
torch.cuda.synchronize()
