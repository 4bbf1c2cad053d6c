"""Stand-alone timing of the fused cross-modal layer (hcm_op_vla_layer) at B environments, L tokens, 16 visual tokens, both streams."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hcm_pkg
hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 80
bf = torch.bfloat16
r = lambda *s: (torch.rand(*s, device="cuda") * 2 - 1)
W = dict(wo=(r(256, 256) * .1).to(bf), w1=(r(1024, 256) * .1).to(bf), w2=(r(256, 1024) * .05).to(bf), bo=r(256), b1=r(1024), b2=r(256), g1=r(256) + 1.5, be1=r(256), g2=r(256) + 1.5, be2=r(256))
q = r(B, L, 256).to(bf); I = r(B, L, 256).to(bf)
kv = [r(B, 16, 512).to(bf) for _ in range(2)]; out = [torch.empty(B, L, 256, device="cuda", dtype=bf) for _ in range(2)]
pooled = [torch.empty(B, 256, device="cuda") for _ in range(2)]
arr = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
Wf = {}
for k, (N, K) in (("wo", (256, 256)), ("w1", (1024, 256)), ("w2", (256, 1024))):
    Wf[k] = torch.empty_like(W[k])
    assert lib.hcm_op_pack_frag(p(W[k]), p(Wf[k]), 1, N, K, None) == 0
def run(frag):
    fn, Wx = (lib.hcm_op_vla_layer_frag, Wf) if frag else (lib.hcm_op_vla_layer, W)
    rc = fn(p(q), p(I), arr(kv), (C.c_int * 2)(16, 16), None, arr(out), arr(pooled) if L <= 80 else None, 256, p(Wx["wo"]), p(W["bo"]), p(Wx["w1"]), p(W["b1"]), p(Wx["w2"]), p(W["b2"]),
            p(W["g1"]), p(W["be1"]), p(W["g2"]), p(W["be2"]), None, 1, B, L, 1024, 2, st)
    assert rc == 0, rc
fl = 2 * B * (2.0 * L * (256 * 256 + 2 * 256 * 1024) + 2 * 2 * L * 16 * 256)
for frag in (0, 1, 0, 1):
    for _ in range(20): run(frag)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): run(frag)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    print(f"B={B} L={L} {'weights -> registers (fragment order)' if frag else 'weights -> LDS ring            '}: {us:.1f} us per launch (both streams), {fl / us / 1e6:.0f} TFLOP/s")
