"""Timing of one skinny long-K linear layer (hcm_op_linear) per K-slice count: HCM_DEV_LIB=1 HCM_SPLITK_FORCE=<S> python tools/splitk_bench.py M N K"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hcm_pkg
hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
M, N, K = (int(a) for a in sys.argv[1:4])
x = (torch.rand(M, K, device="cuda") - 0.5).half(); w = ((torch.rand(N, K, device="cuda") - 0.5) * 0.02).half(); b = torch.rand(N, device="cuda")
y = torch.empty(M, N, device="cuda", dtype=torch.half)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
run = lambda: lib.hcm_op_linear(p(x), p(w), p(b), None, p(y), 5, M, N, K, 1, 0, st)
for _ in range(20): assert run() == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print(f"M={M} N={N} K={K} HCM_SPLITK_FORCE={os.environ.get('HCM_SPLITK_FORCE', '-')} HCM_SPLITK_POW2={os.environ.get('HCM_SPLITK_POW2', '-')}: {e0.elapsed_time(e1) * 5:.1f} us (GEMM + reduction)")
