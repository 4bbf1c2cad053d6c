"""Persistent 256 x 256 GEMM (gemm256p_kernel, HCM_DEV_LIB=1 HCM_GEMM256_PERSIST=1) against the one-tile workgroups at tile grids of more than one tile per CU.
usage: HCM_DEV_LIB=1 [HCM_GEMM256_PERSIST=1] python tools/gemm256_persist_bench.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for (M, N, K, act) in [(20480, 3072, 768, 2), (20480, 2304, 768, 0), (20480, 768, 768, 0), (10240, 3072, 768, 2), (10240, 2304, 768, 0), (5120, 3072, 768, 2), (40960, 3072, 768, 2), (20480, 1024, 256, 1)]:
    x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half(); b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    f = lambda: lib.hcm_op_linear_impl(p(x), p(w), p(b), None, p(y), 5, M, N, K, act, 0, 2, None)
    assert f() == 0
    for _ in range(10): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    print(f"M={M} N={N} K={K} act={act}: {best:.1f} us  {2.0*M*N*K/best/1e6:.0f} TFLOP/s", flush=True)
