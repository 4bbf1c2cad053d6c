"""Phase times of simplecnn3_kernel (development build: s_memtime stamps per wave at the phase boundaries, HCM_S3_PROF_PTR), B frames of H x H.
usage: python tools/simplecnn3_prof.py [B=256] [H=256]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
torch.cuda.init()
h1 = (H - 8) // 4 + 1; h2 = (h1 - 4) // 2 + 1; h3 = h2 - 2; bands = (h3 + 6) // 7
prof = torch.zeros(B * bands * 8 * 5, dtype=torch.int64, device="cuda")
os.environ["HCM_S3_PROF_PTR"] = str(prof.data_ptr()); os.environ["HCM_DEV_LIB"] = "1"
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
p = lambda t: C.c_void_p(t.data_ptr())
x = torch.rand(B, H, H, 1, device="cuda")
w0 = (torch.randn(32, 64, device="cuda") * 0.2).half(); w1 = (torch.randn(64, 512, device="cuda") * 0.08).half(); w2 = (torch.randn(32, 576, device="cuda") * 0.07).half()
b0, b1, b2 = torch.randn(32, device="cuda") * 0.1, torch.randn(64, device="cuda") * 0.1, torch.randn(32, device="cuda") * 0.1
w1f, w2f = torch.empty_like(w1), torch.empty_like(w2)
lib.hcm_op_pack_frag(p(w1), p(w1f), 5, 64, 512, None); lib.hcm_op_pack_frag(p(w2), p(w2f), 5, 32, 576, None)
y = torch.empty(B, h3, h3, 32, device="cuda", dtype=torch.float16)
run = lambda: lib.hcm_op_simplecnn3(p(x), p(w0), p(b0), p(w1f), p(b1), p(w2f), p(b2), p(y), 5, B, H, None)
for _ in range(5): assert run() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
t = prof.view(B * bands, 8, 5).cpu().double()
names = ["input load + stage", "conv1", "conv2", "conv3 + store"]
clk = 2.1e3     # cycles per us, approximately (s_memtime = shader clock)
print(f"B={B} H={H}: {B * bands} workgroups, {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch; phase times per wave (cycles / {clk:.0f} = us), mean and max over waves")
for i, n in enumerate(names):
    d = t[:, :, i + 1] - t[:, :, i]
    print(f"  {n:20s} mean {d.mean().item() / clk:6.2f} us   max {d.max().item() / clk:6.2f} us")
tot = t[:, :, 4] - t[:, :, 0]
print(f"  {'workgroup total':20s} mean {tot.mean().item() / clk:6.2f} us   max {tot.max().item() / clk:6.2f} us;  span of the launch {(t[:, :, 4].max() - t[:, :, 0].min()).item() / clk:.1f} us")
