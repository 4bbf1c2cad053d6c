"""MaxPool2d(3, 2, 1) NHWC kernel at the two trunk stem shapes.  usage: python tools/maxpool_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
for B, H, Cc, dt, code in ((64, 128, 128, torch.bfloat16, 1), (64, 64, 64, torch.float16, 5)):
    x = torch.randn(B, H, H, Cc, device="cuda").to(dt); y = torch.empty(B, H // 2, H // 2, Cc, device="cuda", dtype=dt)
    run = lambda: lib.hcm_op_maxpool3x3s2(x.data_ptr(), y.data_ptr(), code, B, H, H, Cc, None)
    for _ in range(50): assert run() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    print(f"maxpool B={B} {H}x{H}x{Cc}: {us:.1f} us, {(x.numel() + y.numel()) * 2 / us / 1e6:.2f} TB/s")
