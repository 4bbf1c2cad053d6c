"""RGB stem section at the pair shape (B x 256 x 256 uint8 frames, Cout = 128, fp16): pack + conv(+hpool) + vpool (round 5) against pack + the one-launch
stem of round 6 (csrc/stem.hip); the pack launch alone is timed too, so the difference is the conv + pool part.
usage: python tools/stem_fused_bench.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, Cout = 256, 128
x = (torch.rand(B, H, H, 3, device="cuda") * 255).to(torch.uint8)
w = (torch.randn(Cout, 224, device="cuda") * 0.05).half()
b = torch.randn(Cout, device="cuda")
half = torch.empty(B, H // 2, H // 4, Cout, device="cuda", dtype=torch.float16)
y0 = torch.empty(B, H // 4, H // 4, Cout, device="cuda", dtype=torch.float16)
y1 = torch.empty_like(y0)
scratch = torch.empty(lib.hcm_op_stem_scratch_bytes(B, H, H), device="cuda", dtype=torch.uint8)
old = lambda: lib.hcm_op_stem_conv_packed_pool(x.data_ptr(), _lib.HCM_U8, w.data_ptr(), b.data_ptr(), y0.data_ptr(), _lib.HCM_F16, B, H, H, Cout, 1 / 255.0,
                                               scratch.data_ptr(), half.data_ptr(), None)
new = lambda: lib.hcm_op_stem_pool_fused(x.data_ptr(), _lib.HCM_U8, w.data_ptr(), b.data_ptr(), y1.data_ptr(), _lib.HCM_F16, B, H, H, Cout, 1 / 255.0,
                                         scratch.data_ptr(), None)
w1 = (torch.randn(Cout, 64, device="cuda") * 0.1).half()
b1 = torch.randn(Cout, device="cuda")
o1 = torch.empty_like(y0)
red = lambda: lib.hcm_op_stem_pool_fused_red(x.data_ptr(), _lib.HCM_U8, w.data_ptr(), b.data_ptr(), y1.data_ptr(), _lib.HCM_F16, B, H, H, Cout, 1 / 255.0,
                                             scratch.data_ptr(), w1.data_ptr(), b1.data_ptr(), o1.data_ptr(), None)
def t(run, n=50):
    for _ in range(10): assert run() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for _ in range(2):
    print(f"B={B}: pack + conv/hpool + vpool {t(old):.1f} us   |   pack + one-launch stem {t(new):.1f} us   |   ... + layer1 block 0's reduction {t(red):.1f} us")
print("bit-equal:", torch.equal(y0.view(torch.int16), y1.view(torch.int16)))
