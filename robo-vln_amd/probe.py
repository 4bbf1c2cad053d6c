"""BASELINE.json configs[3] ("Depth-only SimpleCNN encoder + 1-layer transformer, batch=256: memory-bound path") as SURVEY 8a/8d
define it: the reference cannot build its high-level model with SimpleCNN encoders, so the configuration is the composition
of two reference classes used unchanged -- `SimpleDepthCNN(obs, 128)` (models/encoders/simple_cnns.py:104-125) -> one visual
token (B,1,128) -> `Visual_Ling_Attn(N=1, vis_in_features=128)` (models/transformer/transformer.py:251-281) over a
pre-computed instruction tensor (B,L,768).  This module runs that composition on the GPU through libhcm's operator entry
points (every arithmetic step is a HIP kernel; torch only owns the buffers) -- a micro-benchmark and parity target, not part
of the drop-in model API.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_DT = {"bf16": (_lib.HCM_BF16, torch.bfloat16), "fp16": (_lib.HCM_F16, torch.float16), "fp32": (_lib.HCM_F32, torch.float32)}


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def sinusoid_table(L, d):
    """common/utils.py:167-185."""
    pos = np.arange(L, dtype=np.float32)[:, None]
    i = np.arange(d // 2, dtype=np.float32)[None]
    ang = pos / np.power(np.float32(10000.0), 2.0 * i / np.float32(d))
    pe = np.zeros((L, d), np.float32)
    pe[:, 0::2] = np.sin(ang)
    pe[:, 1::2] = np.cos(ang)
    return pe


class DepthCnnVlaProbe:
    """cnn_sd: SimpleDepthCNN state_dict (keys cnn.{0,2,4,7}.{weight,bias}); vla_sd: Visual_Ling_Attn state_dict (N = 1)."""

    def __init__(self, cnn_sd, vla_sd, depth_hw=256, instr_len=80, heads=4, precision="fp16", device="cuda", graph=False, fused_layer=True, overlap=True,
                 fused_cnn=True, frag_weights=True):
        """graph=True: forward() is captured once per batch size into a hipGraph (torch.cuda.CUDAGraph over the library's launches on
        the capture stream) with engine-owned static input / output buffers, and replayed: the ~16 dependent launches then cost one."""
        self._graph = bool(graph)
        self._unfused = not fused_layer                 # launch-per-op cross-modal layer (A/B and test aid)
        self._fused_cnn = bool(fused_cnn)               # the three convolutions in one launch (hcm_op_simplecnn3); False: launch per conv (A/B and test aid)
        self._frag_weights = bool(frag_weights)         # fused layer: weights in fragment order, read straight into registers (hcm_op_vla_layer_frag); False: LDS ring
        self._overlap = bool(overlap)                   # instruction branch on a second stream
        self._side = torch.cuda.Stream() if overlap else None
        self._graphs = {}
        self.lib = _lib.lib()
        self.code, self.tdt = _DT[precision]
        self.dev = torch.device(device)
        self.hw, self.L, self.heads = depth_hw, instr_len, heads
        t = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))
        W = lambda a: t(a).to(self.tdt).to(self.dev)
        F32 = lambda a: t(a).to(self.dev)
        g = lambda sd, k: np.asarray(sd[k], dtype=np.float32)
        # SimpleCNN convs: OIHW -> OHWI rows (k = (kh, kw, ci)), K padded to 32 for the narrow-channel first conv
        w0 = g(cnn_sd, "cnn.0.weight").transpose(0, 2, 3, 1).reshape(32, -1)
        k0 = w0.shape[1]
        w0p = np.zeros((32, (k0 + 31) // 32 * 32), np.float32)
        w0p[:, :k0] = w0
        self.c0, self.c0_k, self.c0_kp, self.b0 = W(w0p), k0, w0p.shape[1], F32(g(cnn_sd, "cnn.0.bias"))
        self.c0_plain = W(w0)                               # [32][64]: the packed-frame path of the 16-bit builds
        self.c1, self.b1 = W(g(cnn_sd, "cnn.2.weight").transpose(0, 2, 3, 1)), F32(g(cnn_sd, "cnn.2.bias"))
        self.c2, self.b2 = W(g(cnn_sd, "cnn.4.weight").transpose(0, 2, 3, 1)), F32(g(cnn_sd, "cnn.4.bias"))
        # the 4x4 and 3x3 weights once more in MFMA-fragment order, for the one-launch form of the three convolutions
        self.c1f = self.c2f = None
        if self.tdt != torch.float32 and self._fused_cnn and depth_hw % 4 == 0 and depth_hw <= 256:
            self.c1f, self.c2f = torch.empty_like(self.c1), torch.empty_like(self.c2)
            self._ck(self.lib.hcm_op_pack_frag(_p(self.c1), _p(self.c1f), self.code, 64, 512, None))
            self._ck(self.lib.hcm_op_pack_frag(_p(self.c2), _p(self.c2f), self.code, 32, 576, None))
            torch.cuda.synchronize()
        h1 = (depth_hw - 8) // 4 + 1
        h2 = (h1 - 4) // 2 + 1
        h3 = h2 - 3 + 1
        self.h1, self.h2, self.h3 = h1, h2, h3
        fc = g(cnn_sd, "cnn.7.weight")                                  # (out, 32*h3*h3) over the NCHW flatten: column c*S + s
        S = h3 * h3
        self.fc = W(fc.reshape(-1, 32, S).transpose(0, 2, 1).reshape(fc.shape[0], S * 32))   # ours is NHWC: column s*32 + c
        self.fcb = F32(g(cnn_sd, "cnn.7.bias"))
        self.out_f = fc.shape[0]
        d = g(vla_sd, "layer_norm.weight").shape[0]
        self.d = d
        lin = lambda p: (W(g(vla_sd, p + ".weight")), F32(g(vla_sd, p + ".bias")))
        self.vis_fc, self.ins_fc = lin("vis_fc"), lin("ins_fc")
        self.ln = (F32(g(vla_sd, "layer_norm.weight")), F32(g(vla_sd, "layer_norm.bias")))
        a = "layers.0.enc_att.attention."
        self.fq, self.fo = lin(a + "fc_q"), lin(a + "fc_o")
        self.fkv = (W(np.concatenate([g(vla_sd, a + "fc_k.weight"), g(vla_sd, a + "fc_v.weight")], 0)),
                    F32(np.concatenate([g(vla_sd, a + "fc_k.bias"), g(vla_sd, a + "fc_v.bias")], 0)))
        self.ln_att = (F32(g(vla_sd, "layers.0.enc_att.layer_norm.weight")), F32(g(vla_sd, "layers.0.enc_att.layer_norm.bias")))
        self.f1, self.f2 = lin("layers.0.pwff.fc1"), lin("layers.0.pwff.fc2")
        self.ln_ff = (F32(g(vla_sd, "layers.0.pwff.layer_norm.weight")), F32(g(vla_sd, "layers.0.pwff.layer_norm.bias")))
        self.pe = F32(sinusoid_table(instr_len, d))
        self.wf = None
        d_ff = self.f1[0].shape[0]
        if self._frag_weights and self.tdt != torch.float32 and d == 256 and d_ff % 256 == 0:
            self.wf = tuple(torch.empty_like(w) for w in (self.fo[0], self.f1[0], self.f2[0]))
            for src, dst, (N, K) in zip((self.fo[0], self.f1[0], self.f2[0]), self.wf, ((d, d), (d_ff, d), (d, d_ff))):
                self._ck(self.lib.hcm_op_pack_frag(_p(src), _p(dst), self.code, N, K, None))
            torch.cuda.synchronize()

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(f"libhcm operator failed: {rc}")

    @staticmethod
    def _st():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _lin(self, x, wb, M, act=0, res=None):
        w, b = wb
        y = torch.empty(M, w.shape[0], device=self.dev, dtype=self.tdt)
        self._ck(self.lib.hcm_op_linear(_p(x), _p(w), _p(b), _p(res), _p(y), self.code, M, w.shape[0], w.shape[1], act, 0, self._st()))
        return y

    def _ln(self, x, gb, rows, post=None, post_rows=0):
        y = torch.empty(rows, self.d, device=self.dev, dtype=self.tdt)
        self._ck(self.lib.hcm_op_layernorm_post(_p(x), None, _p(gb[0]), _p(gb[1]), _p(post), post_rows, _p(y), self.code, rows, self.d, 1e-5, self._st()))
        return y

    def forward(self, depth, ins):
        """depth (B,H,W,1) f32 on the device, ins (B,L,768) in the storage type -> (B,L,d).
        graph=True: the first `_INPLACE` distinct (depth, ins) buffer pairs are captured reading the caller's tensors in place (a rollout
        stages observations into the same device buffers every step: no copy); further pairs go through one graph over engine-owned
        static inputs.  The returned tensor belongs to the graph: consume it before the same graph is replayed."""
        if not self._graph:
            return self._forward(depth, ins)
        B = depth.shape[0]
        key = (B, depth.data_ptr(), ins.data_ptr())
        g = self._graphs.get(key)
        if g is None and sum(1 for k in self._graphs if len(k) == 3) < self._INPLACE and depth.is_contiguous() and ins.is_contiguous():
            g = self._graphs[key] = self._capture(depth, ins)                # keeps references: the addresses stay valid
        if g is not None:
            g[0].replay()
            return g[1]["out"]
        g = self._graphs.get((B,))
        if g is None:
            g = self._graphs[(B,)] = self._capture(depth.clone(), ins.clone())
        graph, st = g
        st["depth"].copy_(depth, non_blocking=True)
        st["ins"].copy_(ins, non_blocking=True)
        graph.replay()
        return st["out"]

    _INPLACE = 4

    def _capture(self, depth, ins):
        st = {"depth": depth, "ins": ins}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                        # warm-up outside the capture (one-time kernel attribute setup, scratch growth)
            for _ in range(2):
                self._forward(depth, ins)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            st["out"] = self._forward(depth, ins)
        return graph, st

    def _forward(self, depth, ins):
        L_, lib, code = _lib, self.lib, self.code
        B = depth.shape[0]
        e = lambda *s: torch.empty(*s, device=self.dev, dtype=self.tdt)
        # the instruction branch (ins_fc -> ReLU -> LayerNorm -> + PE over B*L rows) does not depend on the depth CNN: second stream.
        # (its tensors are handed to the main stream at the join; the next call's fork orders any reuse of their memory after this call)
        rows = B * self.L
        main, side = torch.cuda.current_stream(), (self._side if self._overlap else None)
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                I = self._ln(self._lin(ins.reshape(rows, -1), self.ins_fc, rows, act=L_.ACT_RELU), self.ln, rows, self.pe, self.L)
        else:
            I = self._ln(self._lin(ins.reshape(rows, -1), self.ins_fc, rows, act=L_.ACT_RELU), self.ln, rows, self.pe, self.L)
        y2 = None
        if self.c1f is not None:
            y2 = e(B, self.h3, self.h3, 32)
            self._ck(lib.hcm_op_simplecnn3(_p(depth), _p(self.c0_plain), _p(self.b0), _p(self.c1f), _p(self.b1), _p(self.c2f), _p(self.b2), _p(y2), code, B,
                                           self.hw, self._st()))
        y0 = e(B, self.h1, self.h1, 32) if y2 is None else None
        if y2 is not None:
            pass
        elif self.tdt != torch.float32 and self.hw % 4 == 0:
            scratch = e(B * self.hw * self.hw + 64)
            self._ck(lib.hcm_op_depth_conv8x8s4(_p(depth), _p(self.c0_plain), _p(self.b0), _p(y0), code, B, self.hw, L_.ACT_RELU, _p(scratch), self._st()))
        else:
            self._ck(lib.hcm_op_stem_conv(_p(depth), L_.HCM_F32, _p(self.c0), _p(self.b0), _p(y0), code, B, self.hw, self.hw, 1, 32, 8, 8, 4, 0,
                                          self.c0_k, self.c0_kp, 0, 1.0, L_.ACT_RELU, self._st()))
        if y2 is None:
            y1 = e(B, self.h2, self.h2, 64)
            self._ck(lib.hcm_op_conv2d(_p(y0), _p(self.c1), _p(self.b1), None, _p(y1), code, B, self.h1, self.h1, 32, 64, 4, 4, 2, 0, L_.ACT_RELU, self._st()))
            y2 = e(B, self.h3, self.h3, 32)
            self._ck(lib.hcm_op_conv2d(_p(y1), _p(self.c2), _p(self.b2), None, _p(y2), code, B, self.h2, self.h2, 64, 32, 3, 3, 1, 0, L_.ACT_NONE, self._st()))
        tok = self._lin(y2, (self.fc, self.fcb), B, act=L_.ACT_RELU)                 # (B, 128): the one visual token
        V = self._ln(self._lin(tok, self.vis_fc, B, act=L_.ACT_RELU), self.ln, B)     # (B, 1, d)
        if side is not None:
            main.wait_stream(side)                                                    # join: the instruction stream I
        # one visual token = one key: softmax over a single score is exactly 1, so the attention output is the value row whatever the
        # query is -- fc_q(I) (transformer.py:116) cannot reach the output and is not computed; the attention kernel still runs, with I
        # standing in for the queries (any finite values give the same result bit for bit)
        kv = self._lin(V, self.fkv, B)                                               # (B, 1, 2d)
        d_ff = self.f1[0].shape[0]
        if self.tdt != torch.float32 and self.d == 256 and self.heads == 4 and d_ff % 256 == 0 and not self._unfused:
            # attention + fc_o + LayerNorm + feed-forward + LayerNorm: one launch (csrc/vla_fused.hip, the model path's kernel)
            out = e(rows, self.d)
            arr = lambda *v: (C.c_void_p * len(v))(*v)
            fn = lib.hcm_op_vla_layer_frag if self.wf is not None else lib.hcm_op_vla_layer
            wo, w1, w2 = self.wf if self.wf is not None else (self.fo[0], self.f1[0], self.f2[0])
            self._ck(fn(_p(I), _p(I), arr(kv.data_ptr()), (C.c_int * 1)(1), None, arr(out.data_ptr()), None, 0,
                                          _p(wo), _p(self.fo[1]), _p(w1), _p(self.f1[1]), _p(w2), _p(self.f2[1]),
                                          _p(self.ln_att[0]), _p(self.ln_att[1]), _p(self.ln_ff[0]), _p(self.ln_ff[1]), None,
                                          code, B, self.L, d_ff, 1, self._st()))
            return out.reshape(B, self.L, self.d)
        att = e(rows, self.d)
        esz = kv.element_size()
        self._ck(lib.hcm_op_attention(_p(I), _p(kv), C.c_void_p(kv.data_ptr() + self.d * esz), _p(att), code, B, self.heads, self.L, 1,
                                      self.d, 2 * self.d, 2 * self.d, self.d, self._st()))
        o = self._ln(self._lin(att, self.fo, rows, res=I), self.ln_att, rows)
        f = self._lin(self._lin(o, self.f1, rows, act=L_.ACT_RELU), self.f2, rows, res=o)
        return self._ln(f, self.ln_ff, rows).reshape(B, self.L, self.d)
