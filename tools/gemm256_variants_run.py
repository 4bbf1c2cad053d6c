"""Run chosen variants of the 256 x 256 GEMM (`make DEV=1` library) a few times each -- the target of `rocprofv3 --pmc ...` probes.
usage: VARS=0,12 KS=768,3072 python tools/gemm256_variants_run.py [M N reps]   (variant 0 = the shipped schedule)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HCM_DEV_LIB', '1')
import torch, hcm_pkg
hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for K in [int(k) for k in os.environ.get('KS', '768,3072').split(',')]:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).half(); w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.06).half(); b = torch.rand(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for var in [int(v) for v in os.environ.get('VARS', '0,12').split(',')]:
        for _ in range(reps):
            rc = lib.hcm_op_linear_impl(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), 5, M, N, K, 0, 0, 2 + 16 * var, st)
            assert rc == 0, rc
        torch.cuda.synchronize()
