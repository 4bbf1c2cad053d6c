"""Stand-alone timing of attention_mfma_kernel (hcm_op_attention) at the step's shapes.  usage: python tools/att_bench.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
p = lambda t: C.c_void_p(t.data_ptr())
for (B, heads, L, Lk, what) in [(64, 12, 80, 80, "BERT, B = 64"), (128, 12, 160, 160, "BERT, configs[4]"), (1, 12, 80, 80, "BERT, B = 1"), (64, 4, 80, 16, "cross-modal layer 0")]:
    D = heads * 64
    qkv = torch.randn(B * L, 3 * D, device="cuda").half()
    out = torch.empty(B * L, D, device="cuda", dtype=torch.float16)
    if Lk == L:
        f = lambda: lib.hcm_op_attention(p(qkv), C.c_void_p(qkv.data_ptr() + D * 2), C.c_void_p(qkv.data_ptr() + 4 * D), p(out), 5, B, heads, L, L, 3 * D, 3 * D, 3 * D, D, None)
    else:
        kv = torch.randn(B * Lk, 2 * D, device="cuda").half()
        f = lambda: lib.hcm_op_attention(p(qkv), p(kv), C.c_void_p(kv.data_ptr() + 2 * D), p(out), 5, B, heads, L, Lk, 3 * D, 2 * D, 2 * D, D, None)
    assert f() == 0
    for _ in range(20): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(100): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    print(f"{what}: B={B} heads={heads} Lq={L} Lk={Lk}: {best:.1f} us", flush=True)
