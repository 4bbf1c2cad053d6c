"""Phase breakdown of the fused bottleneck launch with register epilogues (bneck231r_kernel<.,128,64,64>, layer1 shape): needs the
development library and its instrumented build.   usage: HCM_DEV_LIB=1 HCM_IGEMM_PROF=1 python tools/bneck_prof.py [B H]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
assert os.environ.get("HCM_IGEMM_PROF") and os.environ.get("HCM_DEV_LIB"), "set HCM_DEV_LIB=1 HCM_IGEMM_PROF=1"
lib = _lib.lib()
B, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 64)
C1 = int(sys.argv[3]) if len(sys.argv) > 3 else 64          # 64 (layer1 shape, @64) or 128 (layer2 shape, @32)
CN, tdt = C1, torch.float16
P = lambda t: t.data_ptr()
x = torch.randn(B, H, H, C1, device="cuda").to(tdt)
w2 = (torch.randn(C1, 3, 3, C1, device="cuda") * 0.05).to(tdt); b2 = torch.randn(C1, device="cuda")
w3 = (torch.randn(4 * C1, 1, 1, C1, device="cuda") * 0.05).to(tdt); b3 = torch.randn(4 * C1, device="cuda")
r = torch.randn(B, H, H, 4 * C1, device="cuda").to(tdt); y = torch.empty_like(r)
w1 = (torch.randn(CN, 1, 1, 4 * C1, device="cuda") * 0.05).to(tdt); b1 = torch.randn(CN, device="cuda"); o1 = torch.empty(B, H, H, CN, device="cuda", dtype=tdt)
run = lambda: lib.hcm_op_bottleneck_tail_next(P(x), P(w2), P(b2), P(w3), P(b3), P(r), P(y), P(w1), P(b1), P(o1), _lib.HCM_F16, B, H, H, C1, 1, CN, None)
out = (C.c_uint64 * 8)()
for _ in range(3):
    assert run() == 0
torch.cuda.synchronize()
assert lib.hcm_debug_igemm_prof(out, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    assert run() == 0
e1.record(); torch.cuda.synchronize()
assert lib.hcm_debug_igemm_prof(out, 1) == 0
v = list(out)
waves = v[6]
names = ["prologue (to first barrier)", "phase A K loop (9 tiles)", "park + slice waits/barriers", "expansion MFMAs", "register epilogues", "block barrier + reduction + final"]
tot = sum(v[:6])
print(f"B={B} {C1}ch @{H}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us/launch (instrumented), {waves // reps} waves/launch")
for n, c in zip(names, v[:6]):
    print(f"  {n:36s} {c / waves:9.0f} cycles/wave  {100 * c / tot:5.1f} %")
print(f"  total {tot / waves:9.0f} cycles/wave = {tot / waves / 2.4e3:.2f} us at 2.4 GHz")
