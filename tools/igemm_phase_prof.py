"""K-loop phase breakdown of the 8-wave bf16 implicit-GEMM kernels (HCM_IGEMM_PROF=1 instrumented builds).
usage: HCM_IGEMM_PROF=1 [HCM_IGEMM_FORCE=c] python tools/igemm_phase_prof.py lin M N K | conv H Cin Cout K stride pad"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
assert os.environ.get("HCM_IGEMM_PROF"), "set HCM_IGEMM_PROF=1"
lib = _lib.lib()
tdt = torch.bfloat16
a = sys.argv[1:]
if a[0] == "lin":
    M, N, K = int(a[1]), int(a[2]), int(a[3])
    x = torch.randn(M, K, device="cuda").to(tdt); w = (torch.randn(N, K, device="cuda") * 0.05).to(tdt); b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=tdt)
    run = lambda: lib.hcm_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), _lib.HCM_BF16, M, N, K, 0, 0, None)
else:
    H, Cin, Cout, K, stride, pad = [int(v) for v in a[1:7]]
    B = 64; Ho = (H + 2 * pad - K) // stride + 1
    x = torch.randn(B, H, H, Cin, device="cuda").to(tdt); w = (torch.randn(Cout, K, K, Cin, device="cuda") * 0.05).to(tdt); b = torch.randn(Cout, device="cuda")
    y = torch.empty(B, Ho, Ho, Cout, device="cuda", dtype=tdt)
    run = lambda: lib.hcm_op_conv2d(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), _lib.HCM_BF16, B, H, H, Cin, Cout, K, K, stride, pad, 1, None)
out = (C.c_uint64 * 8)()
for _ in range(3):
    assert run() == 0
torch.cuda.synchronize()
assert lib.hcm_debug_igemm_prof(out, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    assert run() == 0
e1.record(); torch.cuda.synchronize()
assert lib.hcm_debug_igemm_prof(out, 1) == 0
v = list(out)
waves, iters = v[6], v[7]
names = ["prologue", "dma issue", "reads+mfma", "dma wait", "barrier", "epilogue"]
print(f"{' '.join(a)} force={os.environ.get('HCM_IGEMM_FORCE')}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us/launch (instrumented), {waves // reps} waves/launch, {iters / max(waves, 1):.1f} K-iterations per wave")
tot = sum(v[:6])
for n, c in zip(names, v[:6]):
    per = c / waves
    print(f"  {n:12s} {per:9.0f} cycles/wave  {100 * c / tot:5.1f} %" + (f"   ({c / iters:7.0f} per K-iteration)" if n in names[1:5] else ""))
print(f"  total        {tot / waves:9.0f} cycles/wave (s_memtime ticks)")
