import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hcm_pkg
hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
M = 5120
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, N, K, act in (("ffn1_gelu", 3072, 768, 2), ("ffn1_relu", 3072, 768, 1), ("ffn1_none", 3072, 768, 0), ("n2304_gelu", 2304, 768, 2), ("n2304_none", 2304, 768, 0), ("n3072_k1536_none", 3072, 1536, 0), ("n3072_k1536_gelu", 3072, 1536, 2)):
    x = (torch.rand(M, K, device="cuda") * 2 - 1).half(); w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.06).half(); b = torch.rand(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    best = 1e9
    for rnd in range(3):
        def run():
            rc = lib.hcm_op_linear_impl(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), 5, M, N, K, act, 0, 2, st)
            assert rc == 0, rc
        for _ in range(20): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    print(name, f"{best:.1f} us")
