"""Packed-frame RGB stem (pack kernel + LDS-DMA implicit GEMM, K = 224) at the pair shape (Cout = 128), bf16.
usage: [HCM_IGEMM_FORCE=c] python tools/stem_packed_bench.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, Cout = 256, 128
x = torch.rand(B, H, H, 3, device="cuda") * 255
w = (torch.randn(Cout, 224, device="cuda") * 0.05).to(torch.bfloat16)
b = torch.randn(Cout, device="cuda")
y = torch.empty(B, H // 2, H // 2, Cout, device="cuda", dtype=torch.bfloat16)
scratch = torch.empty(lib.hcm_op_stem_scratch_bytes(B, H, H), device="cuda", dtype=torch.uint8)
run = lambda: lib.hcm_op_stem_conv_packed(x.data_ptr(), _lib.HCM_F32, w.data_ptr(), b.data_ptr(), y.data_ptr(), _lib.HCM_BF16, B, H, H, Cout,
                                          1 / 255.0, 1, scratch.data_ptr(), None)
for _ in range(100): assert run() == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): run()
e1.record(); torch.cuda.synchronize()
print(f"force={os.environ.get('HCM_IGEMM_FORCE')} packed stem B={B} Cout={Cout}: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us (pack + conv)")
