import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
B = 64
tot = 0
# (pixels per sample, channels, groups, launches per step) of the hi|lo PAIR depth trunk at 256x256 frames
for HW, Cc, G, cnt in ((4096, 64, 32, 1), (1024, 64, 32, 6), (1024, 256, 32, 4), (1024, 128, 32, 1), (256, 128, 32, 7), (256, 512, 32, 5),
                       (256, 256, 32, 1), (64, 256, 32, 11), (64, 1024, 32, 7), (64, 512, 32, 1), (16, 512, 32, 5), (16, 2048, 32, 4),
                       (16, 256, 2, 1)):
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
