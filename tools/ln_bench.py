import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
for rows, D in ((5120, 768), (5120, 256), (1024, 256)):
    x = torch.randn(rows, D, device="cuda").half(); r = torch.randn(rows, D, device="cuda").half()
    g = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda"); y = torch.empty_like(x)
    run = lambda: lib.hcm_op_layernorm(x.data_ptr(), None, g.data_ptr(), b.data_ptr(), y.data_ptr(), _lib.HCM_F16, rows, D, 1e-12, None)
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"layernorm rows={rows} D={D}: {us:.1f} us/launch, {2*rows*D*2/us/1e6:.2f} TB/s")
