import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
B, H = 64, 256
x = torch.randint(0, 256, (B, H, H, 3), device="cuda").float()
xu = x.to(torch.uint8)
for Cout in (64, 128):
    for rowrun, src, name in ((1, x, "f32 rowrun"), (0, x, "f32 elementwise"), (0, xu, "u8 elementwise")):
        Kk, Kp = (168, 192) if rowrun else (147, 160)
        w = (torch.randn(Cout, Kp, device="cuda") * 0.05).to(torch.bfloat16)
        b = torch.randn(Cout, device="cuda")
        y = torch.empty(B, 128, 128, Cout, device="cuda", dtype=torch.bfloat16)
        code = _lib.HCM_U8 if src.dtype == torch.uint8 else _lib.HCM_F32
        run = lambda: lib.hcm_op_stem_conv(src.data_ptr(), code, w.data_ptr(), b.data_ptr(), y.data_ptr(), _lib.HCM_BF16, B, H, H, 3, Cout, 7, 7, 2, 3,
                                           Kk, Kp, rowrun, 1 / 255.0, 1, None)
        for _ in range(3): assert run() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        print(f"stem Cout={Cout} {name}: {e0.elapsed_time(e1)/10*1e3:.1f} us")
