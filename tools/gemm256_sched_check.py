"""Bit-identity of an experiment build of the 256-wide GEMM kernel (VAR = variant number, `make DEV=1` library) against the shipped schedule.
usage: VAR=9 [RES=1] python tools/gemm256_sched_check.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HCM_DEV_LIB', '1')      # the experiment variants live in the `make DEV=1` library
import torch, hcm_pkg
hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in ((5120, 3072, 768), (5120, 2304, 768), (20480, 768, 3072), (5000, 3072, 256), (5120, 3072, 320)):
    x = (torch.rand(M, K, device="cuda") * 2 - 1).half(); w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.06).half(); b = torch.rand(N, device="cuda")
    ys = []
    torch.manual_seed(1)
    res = (torch.rand(M, N, device='cuda') - 0.5).half() if os.environ.get('RES') else None      # ONE residual for both variants
    for impl in (2, 2 + 16 * int(os.environ.get('VAR', '9'))):
        y = torch.zeros(M, N, device="cuda", dtype=torch.float16)
        rc = lib.hcm_op_linear_impl(x.data_ptr(), w.data_ptr(), b.data_ptr(), res.data_ptr() if res is not None else None, y.data_ptr(), 5, M, N, K, 2, 0, impl, st)
        assert rc == 0, rc
        ys.append(y)
    torch.cuda.synchronize()
    print(M, N, K, "equal" if torch.equal(ys[0], ys[1]) else "DIFFERENT", (ys[0].float() - ys[1].float()).abs().max().item())
