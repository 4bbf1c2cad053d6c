"""time(K) of the 256x256 8-phase kernel at fixed M, N: slope = per-K-tile cost, intercept = prologue + epilogue + launch."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HCM_DEV_LIB', '1')      # the experiment variants live in the `make DEV=1` library
import torch, hcm_pkg
hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
act = int(sys.argv[3]) if len(sys.argv) > 3 else 0
VARS = [int(v) for v in sys.argv[4].split(',')] if len(sys.argv) > 4 else []
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for K in [int(k) for k in os.environ.get('KS', '768,3072').split(',')]:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).half(); w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.06).half(); b = torch.rand(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    out = []
    for impl in [1, 2] + [2 + 16 * v for v in VARS]:
        def run():
            rc = lib.hcm_op_linear_impl(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), 5, M, N, K, act, 0, impl, st)
            assert rc == 0, rc
        for _ in range(20): run()
        best = 1e9
        for r in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
        out.append(f"impl{impl} {best:7.1f} us {2.0 * M * N * K / best / 1e6:6.0f} TF")
    print(f"M={M} N={N} K={K:5d} act={act}  " + "   ".join(out))
