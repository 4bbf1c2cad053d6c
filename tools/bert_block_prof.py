"""Phase times of bert_attn_block_kernel (development build: s_memtime stamps per wave at the phase boundaries, HCM_BB_PROF_PTR).
usage: HCM_DEV_LIB=1 python tools/bert_block_prof.py [B=64] [L=80]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 80
torch.cuda.init()
prof = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device="cuda")
os.environ["HCM_BB_PROF_PTR"] = str(prof.data_ptr())
os.environ["HCM_DEV_LIB"] = "1"
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
D = 768
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
qkv = torch.randn(B * L, 3 * D, device="cuda").half(); wo = (torch.randn(D, D, device="cuda") * 0.03).half(); bo = torch.randn(D, device="cuda") * 0.1
res = torch.randn(B * L, D, device="cuda").half(); g = torch.rand(D, device="cuda") + 0.5; bt = torch.randn(D, device="cuda") * 0.1
y = torch.empty_like(res)
wf = torch.empty_like(wo); lib.hcm_op_pack_frag(p(wo), p(wf), 5, D, D, None)
for _ in range(5):
    assert lib.hcm_op_bert_attn_block(p(qkv), p(wf), p(bo), p(res), None, p(g), p(bt), p(y), None, 5, B, L, None, 1e-12, None) == 0
torch.cuda.synchronize()
QT = (L + 15) // 16; wps = (QT + 2) // 3
nb = (B + 7) // 8 * 8 * wps
t = prof[: nb * 32].view(nb, 8, 4).cpu().double()
t = t[t[:, 0, 3] > 0]
a = (t[:, :, 1] - t[:, :, 0]); b = (t[:, :, 2] - t[:, :, 1]); c = (t[:, :, 3] - t[:, :, 2])
print(f"B={B} L={L}: {t.shape[0]} workgroups; s_memtime ticks (100 MHz = 10 ns each) per wave, mean / max over waves and workgroups")
for name, x in (("A attention", a), ("B projection", b), ("C epilogue+LN", c), ("total", t[:, :, 3] - t[:, :, 0])):
    print(f"  {name:14s} mean {x.mean().item() * 0.01:7.2f} us   max {x.max().item() * 0.01:7.2f} us   per-wg-max mean {x.max(1).values.mean().item() * 0.01:7.2f} us")
first = t[:, :, 0].min(); last = t[:, :, 3].max()
print(f"  kernel span (first start -> last end) {(last - first).item() * 0.01:.2f} us")
