"""The write-dominated bottleneck expansion conv as it runs in the step: 1x1 conv + bias + residual + ReLU, bf16,
M = B*H*H pixels.  usage: [HCM_IGEMM_FORCE=c] python tools/c3_bench.py [B H Cin Cout]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
a = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else [128, 64, 64, 256]
B, H, Cin, Cout = a
tdt = torch.bfloat16
x = torch.randn(B, H, H, Cin, device="cuda").to(tdt); w = (torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.05).to(tdt)
b = torch.randn(Cout, device="cuda"); r = torch.randn(B, H, H, Cout, device="cuda").to(tdt); y = torch.empty_like(r)
for res in (None, r):
    run = lambda: lib.hcm_op_conv2d(x.data_ptr(), w.data_ptr(), b.data_ptr(), res.data_ptr() if res is not None else None, y.data_ptr(),
                                    _lib.HCM_BF16, B, H, H, Cin, Cout, 1, 1, 1, 0, 1, None)
    for _ in range(3): assert run() == 0
    for _ in range(int(os.environ.get('C3_WARM', '0'))): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    NIT = int(os.environ.get('C3_ITERS', '20'))
    for _ in range(NIT): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / NIT * 1e3
    mb = (x.numel() + y.numel() + (r.numel() if res is not None else 0)) * 2 / 1e6
    print(f"force={os.environ.get('HCM_IGEMM_FORCE')} B={B} {Cin}->{Cout}@{H} residual={res is not None}: {us:.1f} us, {mb:.0f} MB -> {mb / us:.2f} TB/s")
