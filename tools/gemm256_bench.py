"""Stand-alone timing of the BERT GEMM shapes through hcm_op_linear_impl: impl 1 (128-wide igemm) vs impl 2 (256x256 8-phase), interleaved rounds."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hcm_pkg
hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, N, K, act in (("ffn1", 3072, 768, 2), ("qkv", 2304, 768, 0), ("vla_ffn1", 1024, 256, 1)):
    x = (torch.rand(M, K, device="cuda") * 2 - 1).half(); w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.06).half(); b = torch.rand(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    res = {}
    for rnd in range(3):
        for impl in (1, 2):
            def run():
                rc = lib.hcm_op_linear_impl(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), 5, M, N, K, act, 0, impl, st)
                assert rc == 0, rc
            try:
                for _ in range(20): run()
            except AssertionError:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(impl, []).append(e0.elapsed_time(e1) / 50 * 1e3)
    fl = 2.0 * M * N * K
    print(name, f"M={M} N={N} K={K}", {k: f"{min(v):.1f} us = {fl / min(v) / 1e6:.0f} TF" for k, v in res.items()})
