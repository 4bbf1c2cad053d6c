"""Host-side mirror of the reference's `CMANet` flat baseline (robo_vln_baselines/models/cma.py:19-333): the same
tuple-in / tuple-out `forward(batch)` contract and properties, all arithmetic in libhcm.so (HIP, gfx950).

    net = CMANet(CMAEngine(cfg, state_dict, max_batch=...))
    output, stop_out, rnn_hidden_states = net((observations, rnn_hidden_states, prev_actions, masks))    # robo_vln_trainer.py:1096
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .config import CMAConfig
from .policy import _TORCH_DT, _np32


def _to_struct(cfg: CMAConfig, max_batch, precision):
    s = _lib.HcmCmaConfigStruct()
    s.struct_size = C.sizeof(_lib.HcmCmaConfigStruct)
    s.precision = _lib.PRECISIONS[precision]
    s.max_batch = max_batch
    s.rgb_h, s.rgb_w = cfg.rgb_shape
    s.depth_h = s.depth_w = cfg.depth_hw
    s.instr_len = cfg.instr_len
    s.vocab_size, s.embedding_size, s.instr_hidden = cfg.vocab_size, cfg.embedding_size, cfg.instr_hidden
    s.bidirectional = int(cfg.bidirectional)
    s.rgb_out, s.depth_out, s.depth_baseplanes = cfg.rgb_out, cfg.depth_out, cfg.depth_baseplanes
    s.hidden = cfg.hidden
    s.rnn_type = _lib.HCM_LSTM if cfg.rnn_type == "LSTM" else _lib.HCM_GRU
    s.num_actions = cfg.num_actions
    s.use_prev_action, s.rcm_state_encoder = int(cfg.use_prev_action), int(cfg.rcm_state_encoder)
    s.progress_monitor = int(cfg.progress_monitor)
    s.instr_rnn = _lib.HCM_LSTM if cfg.instr_rnn == "LSTM" else _lib.HCM_GRU
    s.ablate_instruction, s.ablate_depth, s.ablate_rgb = int(cfg.ablate_instruction), int(cfg.ablate_depth), int(cfg.ablate_rgb)
    return s


class CMAEngine:
    """Owns one libhcm CMANet handle (weights + workspace) on one GPU."""

    def __init__(self, cfg: CMAConfig, state_dict, max_batch=64, precision="fp16", device=None, graph=False):
        """graph=True: forward() runs on an engine-owned stream with engine-owned static I/O buffers so that libhcm replays one
        captured hipGraph per step; the returned tensors then alias those buffers and stay valid until the second-next call."""
        self._graph = bool(graph)
        self._gstream = None
        self._static = None
        cfg.validate()
        self.cfg = cfg
        self.max_batch = max_batch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._lib = _lib.lib()
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            st = _to_struct(cfg, max_batch, precision)
            _lib.check(self._lib.hcm_cma_create(C.byref(st), C.byref(self._h)))
            try:
                for k, v in state_dict.items():          # load_state_dict(strict=True) semantics
                    a, dt = _np32(v)
                    shape = (C.c_int64 * max(1, a.ndim))(*a.shape)
                    _lib.check(self._lib.hcm_load_tensor(self._h, _lib.HCM_CMA, k.encode(), a.ctypes.data_as(C.c_void_p), dt, shape, a.ndim), self._h)
                _lib.check(self._lib.hcm_finalize(self._h), self._h)
            except Exception:
                self._lib.hcm_destroy(self._h)
                self._h = C.c_void_p()
                raise

    def query(self, what):
        out = C.c_int64()
        with torch.cuda.device(self.device):       # (HCM_STEP_NONFINITE waits for the handle's device)
            _lib.check(self._lib.hcm_query(self._h, what, C.byref(out)), self._h)
        return out.value

    def nonfinite_steps(self):
        """Overflow guard, as HCMEngine.nonfinite_steps (hcm_query(HCM_STEP_NONFINITE)); synchronises the device."""
        return self.query(_lib.HCM_STEP_NONFINITE)

    @property
    def num_recurrent_layers(self):
        return self.query(_lib.HCM_NUM_RECURRENT_LAYERS)

    def close(self):
        if self._h:
            self._lib.hcm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _dev(self, t, dtypes):
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t))
        if t.dtype not in dtypes:
            t = t.to(dtypes[0])
        return t.to(self.device, non_blocking=True).contiguous()

    def forward(self, observations, hidden, masks):
        c = self.cfg
        with torch.cuda.device(self.device):
            rgb = self._dev(observations["rgb"], (torch.float32, torch.uint8))
            depth = self._dev(observations["depth"], (torch.float32,))
            B = rgb.shape[0]
            if tuple(rgb.shape[1:]) != (*c.rgb_shape, 3):
                raise ValueError(f"rgb must be (B,{c.rgb_shape[0]},{c.rgb_shape[1]},3), got {tuple(rgb.shape)}")
            if tuple(depth.shape) != (B, c.depth_hw, c.depth_hw, 1):
                raise ValueError(f"depth must be (B,{c.depth_hw},{c.depth_hw},1), got {tuple(depth.shape)}")
            ids = self._dev(observations["instruction"], (torch.int64, torch.int32, torch.float32))
            # cfg.instr_len is the longest padded instruction the workspace is sized for; every call brings its own L
            if ids.dim() != 2 or ids.shape[0] not in (1, B) or not 1 <= ids.shape[1] <= c.instr_len:
                raise ValueError(f"instruction must be (B or 1, L <= {c.instr_len}), got {tuple(ids.shape)}")
            ids = ids.expand(B, ids.shape[1]).contiguous()                       # cma.py:226
            h_in = self._dev(hidden, (torch.float32,))
            R = self.num_recurrent_layers
            if tuple(h_in.shape) != (R, B, c.hidden):
                raise ValueError(f"rnn_hidden_states must be ({R},{B},{c.hidden}), got {tuple(h_in.shape)}")
            m = self._dev(masks, (torch.float32,)).reshape(B, -1)[:, 0].contiguous()   # masks[:,0] (cma.py:219)
            if self._graph:
                return self._forward_graph(rgb, depth, ids, h_in, m, B)
            out = torch.empty(B, c.num_actions, device=self.device, dtype=torch.float32)
            stop = torch.empty(B, 1, device=self.device, dtype=torch.float32)
            h_out = torch.empty_like(h_in)
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(self._lib.hcm_cma_forward(self._h, rgb.data_ptr(), _TORCH_DT[rgb.dtype], depth.data_ptr(), ids.data_ptr(),
                                                 _TORCH_DT[ids.dtype], B, ids.shape[1], h_in.data_ptr(), m.data_ptr(), out.data_ptr(),
                                                 stop.data_ptr(), h_out.data_ptr(), st), self._h)
        return out, stop, h_out

    def _forward_graph(self, rgb, depth, ids, h_in, m, B):
        c = self.cfg
        if self._gstream is None:
            self._gstream = torch.cuda.Stream(device=self.device)
        st = self._static
        if st is None or st["B"] != B or st["rgb"].dtype != rgb.dtype or st["ids"].dtype != ids.dtype:
            st = {"B": B, "tick": 0, "rgb": torch.empty_like(rgb), "depth": torch.empty_like(depth),
                  "ids": torch.empty(B * c.instr_len, device=self.device, dtype=ids.dtype),
                  "mask": torch.empty_like(m), "h": [torch.zeros_like(h_in) for _ in range(2)],
                  "out": [torch.empty(B, c.num_actions, device=self.device) for _ in range(2)],
                  "stop": [torch.empty(B, 1, device=self.device) for _ in range(2)]}
            self._static = st
        cur, gs = torch.cuda.current_stream(), self._gstream
        gs.wait_stream(cur)
        # observation buffers whose addresses repeat from the previous call are read in place (see HCMEngine._act_graph)
        ptrs = (rgb.data_ptr(), depth.data_ptr(), ids.data_ptr())
        seen = st.setdefault("seen_ptrs", [])
        direct = ptrs in seen and not os.environ.get("HCM_NO_DIRECT_OBS")
        if ptrs in seen:
            seen.remove(ptrs)
        seen.append(ptrs)
        del seen[:-4]
        st["hold"] = (rgb, depth, ids)
        L = ids.shape[1]
        g_rgb, g_depth, g_ids = (rgb, depth, ids) if direct else (st["rgb"], st["depth"], st["ids"][:B * L].view(B, L))
        with torch.cuda.stream(gs):
            i = st["tick"] & 1
            for dst, src in ((g_rgb, rgb), (g_depth, depth), (g_ids, ids), (st["mask"], m), (st["h"][1 - i], h_in)):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
            _lib.check(self._lib.hcm_cma_forward(self._h, g_rgb.data_ptr(), _TORCH_DT[rgb.dtype], g_depth.data_ptr(),
                                                 g_ids.data_ptr(), _TORCH_DT[ids.dtype], B, L, st["h"][1 - i].data_ptr(),
                                                 st["mask"].data_ptr(), st["out"][i].data_ptr(), st["stop"][i].data_ptr(),
                                                 st["h"][i].data_ptr(), C.c_void_p(gs.cuda_stream)), self._h)
            st["tick"] += 1
        cur.wait_stream(gs)
        return st["out"][i], st["stop"][i], st["h"][i]

    # debug taps (tests)
    def enable_taps(self, on=True):
        _lib.check(self._lib.hcm_debug_enable_taps(self._h, int(on)), self._h)

    def get_tap(self, name):
        n = C.c_int64()
        shape = (C.c_int64 * 4)()
        _lib.check(self._lib.hcm_debug_get_tap(self._h, name.encode(), None, 0, C.byref(n), shape), self._h)
        buf = np.empty(n.value, dtype=np.float32)
        _lib.check(self._lib.hcm_debug_get_tap(self._h, name.encode(), buf.ctypes.data_as(C.c_void_p), n.value, C.byref(n), shape), self._h)
        return buf.reshape([d for d in shape if d > 0])


class CMANet:
    """`CMANet.forward(batch)` (models/cma.py:211-333): batch = (observations, rnn_hidden_states, prev_actions, masks) ->
    (output (B,2), stop_out (B,1), rnn_hidden_states).  Like the reference it deletes observations['instruction'] (:228);
    prev_actions is ignored (CMA.use_prev_action = False)."""

    def __init__(self, engine: CMAEngine):
        self.engine = engine

    @property
    def output_size(self):
        return self.engine.cfg.hidden

    @property
    def is_blind(self):
        return False

    @property
    def num_recurrent_layers(self):
        return self.engine.num_recurrent_layers

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def forward(self, batch):
        observations, rnn_hidden_states, prev_actions, masks = batch
        out, stop, hidden = self.engine.forward(observations, rnn_hidden_states, masks)
        if isinstance(observations, dict) and "instruction" in observations:
            del observations["instruction"]
        return out, stop, hidden

    __call__ = forward
