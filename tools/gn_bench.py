import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
B = 64
tot = 0
for HW, Cc, G, cnt in ((4096, 32, 16, 2), (1024, 32, 16, 12), (1024, 128, 16, 8), (256, 64, 16, 16), (256, 256, 16, 10), (64, 128, 16, 24),
                       (64, 512, 16, 14), (16, 256, 16, 12), (16, 1024, 16, 8), (16, 128, 1, 2)):
    x = torch.randn(B, HW, Cc, device="cuda").half(); r = torch.randn(B, HW, Cc, device="cuda").half()
    g = torch.randn(Cc, device="cuda"); b = torch.randn(Cc, device="cuda")
    st = torch.empty(B * G * 2, device="cuda")
    # hcm_op_groupnorm allocates scratch + syncs; time via events around a loop is still indicative of kernel time
    run = lambda: lib.hcm_op_groupnorm(x.data_ptr(), r.data_ptr(), g.data_ptr(), b.data_ptr(), _lib.HCM_F16, B, HW, Cc, G, 1e-5, 1, None)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = B * HW * Cc * 2 / 1e6
    tot += us * cnt
    print(f"GN HW={HW:5d} C={Cc:5d} G={G:2d} x{cnt:2d}: {us:6.1f} us  tensor {mb:6.1f} MB -> {3*mb/us/1e6*1e6/1e6:.2f} TB/s eff")
print("sum/step ms", tot / 1e3)
