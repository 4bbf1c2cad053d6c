"""Stand-alone timing of the fused BERT attention block launch (hcm_op_bert_attn_block) against the three launches it replaces, at B x L token rows.
usage: python tools/bert_block_bench.py [B=64] [L=80]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 80
D = 768
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
qkv = (torch.randn(B * L, 3 * D, device="cuda")).half()
wo = (torch.randn(D, D, device="cuda") * 0.03).half()
bo = torch.randn(D, device="cuda") * 0.1
res = torch.randn(B * L, D, device="cuda").half()
g = torch.rand(D, device="cuda") + 0.5; bt = torch.randn(D, device="cuda") * 0.1
ctx = torch.empty(B * L, D, device="cuda", dtype=torch.float16); tmp = torch.empty_like(ctx); y = torch.empty_like(ctx); y2 = torch.empty_like(ctx)
wf = torch.empty_like(wo); lib.hcm_op_pack_frag(p(wo), p(wf), 5, D, D, None)
def three():
    lib.hcm_op_attention(p(qkv), C.c_void_p(qkv.data_ptr() + D * 2), C.c_void_p(qkv.data_ptr() + 4 * D), p(ctx), 5, B, 12, L, L, 3 * D, 3 * D, 3 * D, D, None)
    lib.hcm_op_linear(p(ctx), p(wo), p(bo), p(res), p(tmp), 5, B * L, D, D, 0, 0, None)
    lib.hcm_op_layernorm(p(tmp), None, p(g), p(bt), p(y), 5, B * L, D, 1e-12, None)
def one():
    assert lib.hcm_op_bert_attn_block(p(qkv), p(wf), p(bo), p(res), None, p(g), p(bt), p(y2), None, 5, B, L, None, 1e-12, None) == 0
def t(f, n=200):
    for _ in range(20): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
three(); one(); torch.cuda.synchronize()
print(f"B={B} L={L}: equal={torch.equal(y, y2)}  three launches {t(three):.1f} us   one launch {t(one):.1f} us")
