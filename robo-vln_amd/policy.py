"""Host-side mirror of the reference's model-call interface for the HCM hot path.

`Seq2Seq_HighLevel_CMA` / `Seq2Seq_LowLevel` keep the reference's tuple-in / tuple-out `forward(batch)`
contracts (robo_vln_baselines/models/seq2seq_highlevel_cma.py:170-233, seq2seq_lowlevel.py:116-162) and
properties, so the eval loop of hierarchical_trainer.py:1088-1197 can call them unchanged; `Policy.act()` is
the fused step (hi -> argmax -> lo) of :1095-1101.  All arithmetic happens in libhcm.so (HIP, gfx950);
torch is used only for device buffers and streams.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .config import HCMConfig

_TORCH_DT = {torch.float32: _lib.HCM_F32, torch.uint8: _lib.HCM_U8, torch.int32: _lib.HCM_I32, torch.int64: _lib.HCM_I64}


_SUB_SLOTS = {"depth": 0, "bert": 1, "vla": 2, "rgb": 3}
_SUB_DT = {"fp32": _lib.HCM_F32, "bf16": _lib.HCM_BF16, "fp16": _lib.HCM_F16}


def _ptr(t):
    """device pointer of an optional tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()


def _to_struct(cfg: HCMConfig, max_batch, precision, build_high, build_low, sub_precision=None, max_instr_len=None, keep_host_weights=False,
               share_trunks=True):
    s = _lib.HcmConfigStruct()
    for name, p in (sub_precision or {}).items():
        s.reserved[_SUB_SLOTS[name]] = _SUB_DT[p] + 1
    s.reserved[4] = int(bool(keep_host_weights))
    s.reserved[5] = int(not share_trunks)
    s.struct_size = C.sizeof(_lib.HcmConfigStruct)
    if precision not in _lib.PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {precision!r}")
    s.precision = _lib.PRECISIONS[precision]
    s.max_batch = max_batch
    s.rgb_h, s.rgb_w = cfg.rgb_shape
    s.depth_h, s.depth_w = cfg.depth_shape
    s.instr_len = max_instr_len or cfg.instr_len           # the library's instr_len is the MAXIMUM L of a call
    s.rgb_encoder = _lib.HCM_ENC_RESNET if cfg.rgb_encoder == "TorchVisionResNet50" else _lib.HCM_ENC_SIMPLECNN
    s.depth_encoder = _lib.HCM_ENC_RESNET if cfg.depth_encoder == "VlnResnetDepthEncoder" else _lib.HCM_ENC_SIMPLECNN
    s.rgb_out, s.depth_out, s.depth_baseplanes = cfg.rgb_out, cfg.depth_out, cfg.depth_baseplanes
    s.vla_layers, s.d_model, s.vla_heads, s.d_ff = cfg.vla_layers, cfg.d_model, cfg.vla_heads, cfg.d_ff
    s.vis_in, s.ins_in = cfg.vis_in, cfg.ins_in
    s.hidden = cfg.hidden
    s.rnn_type = _lib.HCM_LSTM if cfg.rnn_type == "LSTM" else _lib.HCM_GRU
    s.num_actions, s.num_sub_tasks, s.lo_actions = cfg.num_actions, cfg.num_sub_tasks, cfg.lo_actions
    s.bert_layers, s.bert_hidden, s.bert_heads = cfg.bert_layers, cfg.bert_hidden, cfg.bert_heads
    s.bert_inter, s.bert_vocab, s.bert_max_pos = cfg.bert_inter, cfg.bert_vocab, cfg.bert_max_pos
    s.build_high, s.build_low = int(build_high), int(build_low)
    s.use_prev_action, s.ablate_instruction = int(cfg.use_prev_action), int(cfg.ablate_instruction)
    s.progress_monitor = int(cfg.progress_monitor)
    s.ablate_depth, s.ablate_rgb = int(cfg.ablate_depth), int(cfg.ablate_rgb)
    return s


def _np32(v):
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().numpy()
    v = np.asarray(v)
    if v.dtype == np.int64:
        return np.require(v, requirements="C"), _lib.HCM_I64          # keeps 0-d (num_batches_tracked) 0-d
    return np.require(v, dtype=np.float32, requirements="C"), _lib.HCM_F32


class HCMEngine:
    """Owns one libhcm handle (weights + workspace) on one GPU.  One engine per device per thread."""

    def __init__(self, cfg: HCMConfig, high_level_state_dict=None, low_level_state_dict=None, max_batch=64,
                 precision="fp16", device=None, sub_precision=None, graph=False, max_instr_len=None, keep_host_weights=False, guard_every=64,
                 share_trunks=True, chain_graphs="auto"):
        """precision (fp32 accumulation, fp32 recurrent cells / heads in every mode):
          "fp16"  fp16 storage + fp16 MFMA tiles in every sub-network, behind the range calibration of hcm_finalize (the measured 16-bit mode:
                  record error 2.6e-3 at B = 64).  A trunk that leaves the fp16 range gets an exact power-of-two range fold (`range_fold`),
                  BERT / the cross-modal block fall back to bf16 tiles (`fp16_fallback`);
          "bf16"  bf16 storage + bf16 MFMA tiles in BERT, the RGB trunks and the cross-modal block; the GroupNorm depth trunks stay on
                  range-folded fp16 tiles (on bf16 that trunk alone costs 1.9e-2 of the 1e-2 record tolerance);
          "fp32"  fp32 storage + fp32 MFMA.
        `sub_precision` overrides the storage type per sub-network, e.g. {"depth": "bf16"} or {"bert": "fp16"} (keys: depth, bert, vla, rgb).
        share_trunks: when the two models' trunk weights are bit-identical (frozen pretrained encoders in both state_dicts) the trunk runs once per
        step and feeds both heads; False runs it per model (test aid).
        guard_every: act() polls the run-time overflow guard every this many steps without synchronising (hcm_guard_poll) and raises
        FloatingPointError once a recurrent cell has seen non-finite gate pre-activations; 0 disables.
        graph=True: act() runs on an engine-owned stream with engine-owned static I/O buffers, so that libhcm replays
        one captured hipGraph per step; the returned record / hidden tensors then alias those buffers and stay valid
        until the second-next act() call (ping-pong), which is what a rollout loop that rebinds them every step needs.
        chain_graphs (with graph=True): replay the step as one LINEAR hipGraph per encoder chain instead of one graph captured across the forked
        streams (HCM_ACT_CHAIN_GRAPHS, include/hcm.h): 0.18 ms of host time per step instead of 0.5-0.8 ms and a lower synchronous latency at B = 1,
        2-4 % less pipelined throughput.  "auto" = for calls of one or two environments (the reference's own evaluation loop, one policy call per
        simulator step), and for host_frames=True calls whose frame tensors are at least 4 MB each (the copies then overlap BERT: +5-12 % PCIe-inclusive
        throughput from B = 24 up); True / False force it.  Bit-identical either way.
        max_instr_len: the longest instruction (tokens) a call may carry; sizes the workspace.  Every call takes its own
        (B or 1, L <= max_instr_len) ids, as the reference model does (its eval loop feeds the unpadded tokens of the episode's
        instruction, common/utils.py:18-20).  Default: cfg.instr_len.  BERT's position table allows up to 512."""
        self.max_instr_len = int(max_instr_len or cfg.instr_len)
        # fp16 range safety: hcm_finalize checks the fp16 sub-networks on a synthetic batch and repairs what would overflow (`range_fold`,
        # `fp16_fallback`); keep_host_weights=True keeps the f32 host copies so that `calibrate(observations)` can repeat the check -- and
        # the repair -- on real observations
        self._graph = bool(graph)
        if chain_graphs not in ("auto", True, False):
            raise ValueError('chain_graphs must be "auto", True or False')
        self._chain_graphs = chain_graphs
        self.comm_world, self.comm_rank = 0, 0          # > 0 once comm_init() has created the RCCL communicator
        self._guard_every = int(guard_every)
        self._guard_tick = 0
        self._guard_seen = 0
        self._gather_B = 0                              # per-rank batch of the library collective (act(gather=True)), 0 before the first such call
        self._gather_called = False
        self.guard_alarm = 0                            # deferred alarms of act(gather=True) steps, see guard_check()
        self._gstream = None
        self._static = None
        cfg.validate()
        self.cfg = cfg
        self.max_batch = max_batch
        self.precision = precision
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._lib = _lib.lib()
        self._h = C.c_void_p()
        self.has_high = high_level_state_dict is not None
        self.has_low = low_level_state_dict is not None
        with torch.cuda.device(self.device):
            st = _to_struct(cfg, max_batch, precision, self.has_high, self.has_low, sub_precision, self.max_instr_len, keep_host_weights, share_trunks)
            _lib.check(self._lib.hcm_create(C.byref(st), C.byref(self._h)))
            try:
                # load_state_dict(strict=True) semantics (hierarchical_trainer.py:343-345)
                for model, sd in ((_lib.HCM_HIGH, high_level_state_dict), (_lib.HCM_LOW, low_level_state_dict)):
                    if sd is None:
                        continue
                    for k, v in sd.items():
                        if k.endswith(("embeddings.position_ids", "embeddings.token_type_ids")):
                            continue        # BertEmbeddings buffers of some transformers versions, not parameters
                        a, dt = _np32(v)
                        shape = (C.c_int64 * max(1, a.ndim))(*a.shape)
                        _lib.check(self._lib.hcm_load_tensor(self._h, model, k.encode(), a.ctypes.data_as(C.c_void_p), dt,
                                                             shape, a.ndim), self._h)
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

    def _guard_poll(self, stream, defer=False):
        """Every `guard_every` act() calls: non-blocking read of the overflow guard (the value is the one an EARLIER poll enqueued).
        defer=True (act(gather=True): env-sharded ranks): the alarm is only RECORDED (self.guard_alarm) -- a rank that raised on its own in
        the middle of a rollout would leave its peers blocked in the next step's all-gather; guard_check() agrees on it across ranks."""
        if self._guard_every <= 0:
            return
        self._guard_tick += 1
        if self._guard_tick % self._guard_every:
            return
        out = C.c_int64()
        _lib.check(self._lib.hcm_guard_poll(self._h, stream, C.byref(out)), self._h)
        if out.value > self._guard_seen:
            new, self._guard_seen = out.value - self._guard_seen, out.value
            if defer:
                self.guard_alarm += new
                return
            raise FloatingPointError(f"overflow guard: {new} (environment, recurrent step) pairs had non-finite activations in front of a recurrent "
                                     "cell since the last check -- broken sensor frames, or a sub-network outside its fp16 range "
                                     "(engine.calibrate(observations) on real observations; engine.calibration_report())")

    def guard_check(self, group=None):
        """Env-sharded ranks: agree on the deferred overflow-guard alarms (MAX over the ranks of `group`, one tiny all-reduce) and raise
        FloatingPointError on EVERY rank when any rank has one -- all ranks leave the rollout at the same step, nobody is left inside a
        collective.  rollout() calls it every `guard_every` steps; a single-process engine raises from act() directly and needs no call."""
        alarm = int(self.guard_alarm)
        if self.comm_world > 1 and torch.distributed.is_available() and torch.distributed.is_initialized():
            t = torch.tensor([alarm], device=self.device, dtype=torch.int64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=group)
            alarm = int(t.item())
        if alarm:
            self.guard_alarm = 0
            raise FloatingPointError(f"overflow guard: up to {alarm} (environment, recurrent step) pairs on one rank had non-finite activations in "
                                     "front of a recurrent cell since the last check (engine.calibrate(observations); engine.calibration_report())")

    def comm_abort(self):
        """Tear the library's communicator down (hcm_comm_abort): for a rank that is going to stop stepping outside an agreed point."""
        _lib.check(self._lib.hcm_comm_abort(self._h), self._h)
        self.comm_world, self.comm_rank = 0, 0
        self._gather_B = 0

    @property
    def num_recurrent_layers(self):
        return self.query(_lib.HCM_NUM_RECURRENT_LAYERS)

    @property
    def fp16_fallback(self):
        """Sub-networks the range calibration moved from fp16 to bf16 storage: subset of {"bert", "depth", "rgb", "vla"}."""
        bits = self.query(_lib.HCM_FP16_FALLBACK)
        return {n for b, n in ((1, "bert"), (2, "depth"), (4, "rgb"), (8, "vla")) if bits & b}

    @property
    def range_fold(self):
        """Trunks the range calibration kept on fp16 by folding a power of two into their weights (exact): subset of {"depth", "rgb"}."""
        bits = self.query(_lib.HCM_RANGE_FOLD)
        return {n for b, n in ((2, "depth"), (4, "rgb")) if bits & b}

    def calibration_report(self):
        return {"bert_max_abs": self.query(_lib.HCM_CALIB_MAX_BERT), "depth_max_abs": self.query(_lib.HCM_CALIB_MAX_DEPTH),
                "rgb_max_abs": self.query(_lib.HCM_CALIB_MAX_RGB), "vla_max_abs": self.query(_lib.HCM_CALIB_MAX_VLA),
                "non_finite": self.query(_lib.HCM_CALIB_NONFINITE), "fp16_fallback": sorted(self.fp16_fallback),
                "range_fold": sorted(self.range_fold)}

    def nonfinite_steps(self):
        """Overflow guard (hcm_query(HCM_STEP_NONFINITE)): number of (sample, recurrent step) pairs since construction whose gate
        pre-activations were not all finite -- an fp16 overflow or a NaN anywhere upstream of the state encoders ends up there, and the
        squashing cell would otherwise turn it into finite garbage.  0 on a healthy engine.  Synchronises the device: call it per episode
        or per evaluation, not per step."""
        return self.query(_lib.HCM_STEP_NONFINITE)

    def comm_init(self, group=None):
        """Create the library's own RCCL communicator over the ranks of a torch.distributed process group (one process per GPU): rank 0 draws the
        ncclUniqueId (hcm_comm_unique_id), the existing group broadcasts its 128 bytes, every rank calls hcm_comm_init.  Afterwards
        act(..., gather=True) returns the all-gathered (world * B, 7) record of the whole rollout batch, the collective being enqueued by the
        library behind the step on the step's stream (hcm_act_gather) -- no torch.distributed call per step."""
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        ident = (C.c_char * 128)()
        ok = 1
        if rank == 0:
            ok = 1 if self._lib.hcm_comm_unique_id(ident) == 0 else 0
        # 128 id bytes + one status byte: a rank-0 failure (no librccl to dlopen) reaches every rank as an exception, not as a hang in the broadcast
        t = torch.tensor(list(bytes(ident)) + [ok], dtype=torch.uint8)
        on_dev = dist.get_backend(group) == "nccl"
        if on_dev:
            t = t.to(self.device)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(t.cpu().numpy().tobytes())
        if raw[128] != 1:
            raise RuntimeError("hcm_comm_unique_id failed on rank 0 (librccl not loadable?)")
        with torch.cuda.device(self.device):
            rc = self._lib.hcm_comm_init(self._h, C.c_char_p(raw[:128]), rank, world)
        # every rank learns whether ALL ranks have a communicator (a rank that failed would otherwise leave the others in their first collective)
        flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=self.device if on_dev else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) != 1:
            raise RuntimeError("hcm_comm_init failed on at least one rank" + (": " + _lib.last_error(self._h) if rc != 0 else ""))
        self.comm_world, self.comm_rank = world, rank
        return world

    def calibrate(self, observations, release_host_weights=True):
        """Range-check the fp16 sub-networks on these observations (hcm_calibrate); returns calibration_report().  Needs
        keep_host_weights=True at construction to be able to re-build a sub-network on bf16 tiles."""
        with torch.cuda.device(self.device):
            rgb, depth, ids, _, B = self._obs(observations, self.has_high)
            _lib.check(self._lib.hcm_calibrate(self._h, rgb.data_ptr(), _TORCH_DT[rgb.dtype], depth.data_ptr(), _ptr(ids),
                                               _TORCH_DT[ids.dtype] if ids is not None else _lib.HCM_I64, B, ids.shape[1] if ids is not None else 1,
                                               self._stream()), self._h)
            if release_host_weights:
                _lib.check(self._lib.hcm_release_host_weights(self._h), self._h)
        # hcm_calibrate re-synchronised the library's polled view of the overflow guard with the device word (zeroed by a re-build, possibly
        # raised by the measuring passes): alarms are counted from THAT value on -- neither a stale count (which would hide new alarms) nor a
        # cached earlier one (which would raise again for steps already reported)
        self._guard_seen = self.query(_lib.HCM_STEP_NONFINITE)
        self._guard_tick = 0
        return self.calibration_report()

    def close(self):
        if self._h:
            self._lib.hcm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    def _dev(self, t, dtypes):
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t))
        if t.dtype not in dtypes:
            t = t.to(dtypes[0])
        return t.to(self.device, non_blocking=True).contiguous()

    def _host_frame(self, t, dtypes, name):
        # HCM_ACT_HOST_FRAMES: the library copies the frames itself, chain by chain -- they must be page-locked host tensors as they are
        if not isinstance(t, torch.Tensor) or t.device.type != "cpu" or not t.is_pinned() or not t.is_contiguous() or t.dtype not in dtypes:
            raise ValueError(f"host_frames=True: {name} must be a contiguous pinned CPU tensor of dtype {' / '.join(str(d) for d in dtypes)}")
        return t

    def _obs(self, observations, need_ids, host_frames=False):
        if host_frames:
            rgb = self._host_frame(observations["rgb"], (torch.float32, torch.uint8), "rgb")
            depth = self._host_frame(observations["depth"], (torch.float32,), "depth")
        else:
            rgb = self._dev(observations["rgb"], (torch.float32, torch.uint8))
            depth = self._dev(observations["depth"], (torch.float32,))
        B = rgb.shape[0]
        c = self.cfg
        if tuple(rgb.shape[1:]) != (*c.rgb_shape, 3):
            raise ValueError(f"rgb must be (B,{c.rgb_shape[0]},{c.rgb_shape[1]},3), got {tuple(rgb.shape)}")
        if tuple(depth.shape) != (B, *c.depth_shape, 1):
            raise ValueError(f"depth must be (B,{c.depth_shape[0]},{c.depth_shape[1]},1), got {tuple(depth.shape)}")
        ids = lens = None
        if need_ids:
            ids = self._dev(observations["instruction"], (torch.int64, torch.int32, torch.float32))
            if ids.dim() != 2 or ids.shape[0] not in (1, B) or not 1 <= ids.shape[1] <= self.max_instr_len:
                raise ValueError(f"instruction must be (B or 1, L) with 1 <= L <= {self.max_instr_len} (max_instr_len), got {tuple(ids.shape)}")
            # instruction.expand(B, L) (seq2seq_highlevel_cma.py:189-190)
            ids = ids.expand(B, ids.shape[1]).contiguous()
            # extension for batched rollouts (not a reference key): per-environment token counts of a padded ragged batch
            if observations.get("instruction_lengths") is not None:
                lens = self._dev(observations["instruction_lengths"], (torch.int32,)).reshape(-1)
                if lens.numel() != B:
                    raise ValueError(f"instruction_lengths must hold {B} entries, got {lens.numel()}")
        return rgb, depth, ids, lens, B

    def _mask(self, masks, B):
        m = self._dev(masks, (torch.float32,))
        # the reference reads masks[:,0] (seq2seq_highlevel_cma.py:208); accept (B,), (B,1), (B,2), (B,2,1)
        m = m.reshape(B, -1)[:, 0].contiguous()
        return m

    def _hidden(self, h, B=None):
        h = self._dev(h, (torch.float32,))
        R = self.num_recurrent_layers
        if h.dim() != 3 or h.shape[0] != R or h.shape[2] != self.cfg.hidden or (B is not None and h.shape[1] != B):
            raise ValueError(f"hidden state must be ({R},{B if B is not None else 'N'},{self.cfg.hidden}), got {tuple(h.shape)}")
        return h

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- the three calls
    def high_forward(self, observations, hidden, masks):
        with torch.cuda.device(self.device):
            rgb, depth, ids, lens, B = self._obs(observations, True)
            h_in, m = self._hidden(hidden, B), self._mask(masks, B)
            logits = torch.empty(B, self.cfg.num_actions, device=self.device, dtype=torch.float32)
            h_out = torch.empty_like(h_in)
            _lib.check(self._lib.hcm_high_forward(self._h, rgb.data_ptr(), _TORCH_DT[rgb.dtype], depth.data_ptr(),
                                                  ids.data_ptr(), _TORCH_DT[ids.dtype], _ptr(lens), B, ids.shape[1], h_in.data_ptr(), m.data_ptr(),
                                                  logits.data_ptr(), h_out.data_ptr(), self._stream()), self._h)
        return logits, h_out

    def low_forward(self, observations, hidden, masks, subtask):
        with torch.cuda.device(self.device):
            rgb, depth, _, _, B = self._obs(observations, False)
            h_in, m = self._hidden(hidden, B), self._mask(masks, B)
            st = self._dev(subtask, (torch.int64,)).reshape(B)
            vel = torch.empty(B, self.cfg.lo_actions, device=self.device, dtype=torch.float32)
            stop = torch.empty(B, 1, device=self.device, dtype=torch.float32)
            h_out = torch.empty_like(h_in)
            _lib.check(self._lib.hcm_low_forward(self._h, rgb.data_ptr(), _TORCH_DT[rgb.dtype], depth.data_ptr(), B,
                                                 h_in.data_ptr(), m.data_ptr(), st.data_ptr(), vel.data_ptr(),
                                                 stop.data_ptr(), h_out.data_ptr(), self._stream()), self._h)
        return vel, stop, h_out

    def _act_graph(self, observations, hi_hidden, lo_hidden, masks, flags=0, gather=False):
        host_frames = bool(flags & _lib.HCM_ACT_HOST_FRAMES)
        with torch.cuda.device(self.device):
            rgb, depth, ids, lens, B = self._obs(observations, True, host_frames)
            if gather:
                self._gather_B = B
            L = ids.shape[1]
            hh, lh, m = self._hidden(hi_hidden, B), self._hidden(lo_hidden, B), self._mask(masks, B)
            if self._gstream is None:
                self._gstream = torch.cuda.Stream(device=self.device)
            st = self._static
            if st is None or st["B"] != B or st["rgb"].dtype != rgb.dtype or st["ids"].dtype != ids.dtype or st["rgb"].device != rgb.device:
                # the static ids buffer holds the longest instruction; a call uses its first B*L elements, so the graph of a new L
                # differs by its key only, not by the buffer address
                st = {"B": B, "tick": 0, "rgb": torch.empty_like(rgb), "depth": torch.empty_like(depth),
                      "ids": torch.empty(B * self.max_instr_len, device=self.device, dtype=ids.dtype),
                      "lens": torch.empty(B, device=self.device, dtype=torch.int32),
                      "mask": torch.empty_like(m), "rec": [torch.empty(B, 7, device=self.device) for _ in range(2)],
                      "hh": [torch.zeros_like(hh) for _ in range(2)], "lh": [torch.zeros_like(lh) for _ in range(2)]}
                self._static = st
            cur = torch.cuda.current_stream()
            gs = self._gstream
            gs.wait_stream(cur)
            # Observation buffers the caller re-uses from step to step (an ObsStager's device tensors, a vectorised simulator's
            # output buffers) are read in place: the captured graph is keyed by their addresses like by any other argument.
            # Tensors seen for the first time go through the engine's static copies, so that fresh allocations every step
            # do not force a new capture every step.
            ptrs = (rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), m.data_ptr(), L, lens.data_ptr() if lens is not None else 0)
            # (a caller that rotates a few buffer sets -- double-buffered staging -- is recognised as well: the last four sets)
            seen = st.setdefault("seen_ptrs", [])
            direct = ptrs in seen and not os.environ.get("HCM_NO_DIRECT_OBS")
            if ptrs in seen:
                seen.remove(ptrs)
            seen.append(ptrs)
            del seen[:-4]
            st["hold"] = (rgb, depth, ids, m, lens)        # keep the caller's tensors alive while the graph may read them
            s_ids = st["ids"][:B * L].view(B, L)
            g_rgb, g_depth, g_ids, g_m = (rgb, depth, ids, m) if direct else (st["rgb"], st["depth"], s_ids, st["mask"])
            if host_frames:                                # pinned host frames are always read in place (the library stages them)
                g_rgb, g_depth = rgb, depth
            g_lens = None if lens is None else lens if direct else st["lens"]
            with torch.cuda.stream(gs):
                i = st["tick"] & 1
                for dst, src in ((g_rgb, rgb), (g_depth, depth), (g_ids, ids), (g_m, m), (g_lens, lens),
                                 (st["hh"][1 - i], hh), (st["lh"][1 - i], lh)):
                    if dst is not None and dst.data_ptr() != src.data_ptr():
                        dst.copy_(src, non_blocking=True)
                if gather:
                    if "gat" not in st:
                        st["gat"] = [torch.empty(self.comm_world * B, 7, device=self.device) for _ in range(2)]
                    self._gather_called = True
                    _lib.check(self._lib.hcm_act_gather(self._h, g_rgb.data_ptr(), _TORCH_DT[rgb.dtype], g_depth.data_ptr(),
                                                        g_ids.data_ptr(), _TORCH_DT[ids.dtype], _ptr(g_lens), B, L, st["hh"][1 - i].data_ptr(),
                                                        st["lh"][1 - i].data_ptr(), g_m.data_ptr(), st["rec"][i].data_ptr(),
                                                        st["hh"][i].data_ptr(), st["lh"][i].data_ptr(), flags, st["gat"][i].data_ptr(),
                                                        C.c_void_p(gs.cuda_stream)), self._h)
                else:
                    _lib.check(self._lib.hcm_act_ex(self._h, g_rgb.data_ptr(), _TORCH_DT[rgb.dtype], g_depth.data_ptr(),
                                                    g_ids.data_ptr(), _TORCH_DT[ids.dtype], _ptr(g_lens), B, L, st["hh"][1 - i].data_ptr(),
                                                    st["lh"][1 - i].data_ptr(), g_m.data_ptr(), st["rec"][i].data_ptr(),
                                                    st["hh"][i].data_ptr(), st["lh"][i].data_ptr(), flags, C.c_void_p(gs.cuda_stream)), self._h)
                st["tick"] += 1
            cur.wait_stream(gs)
            # (after the join: an alarm raised from here leaves the caller's stream ordered behind the step it interrupts)
            self._guard_poll(C.c_void_p(gs.cuda_stream), defer=gather)
        return (st["gat"][i] if gather else st["rec"][i]), st["hh"][i], st["lh"][i]

    # ---- training / validation path: T*N frames per call, RNNStateEncoder.seq_forward (state_encoder.py:83-133)
    def high_forward_seq(self, observations, hidden, masks):
        with torch.cuda.device(self.device):
            rgb, depth, ids, lens, TN = self._obs(observations, True)
            h_in = self._hidden(hidden)
            N = h_in.shape[1]
            if TN % N:
                raise ValueError(f"{TN} frames is not a multiple of the hidden batch {N}")
            m = self._mask(masks, TN)
            logits = torch.empty(TN, self.cfg.num_actions, device=self.device, dtype=torch.float32)
            h_out = torch.empty_like(h_in)
            _lib.check(self._lib.hcm_high_forward_seq(self._h, rgb.data_ptr(), _TORCH_DT[rgb.dtype], depth.data_ptr(), ids.data_ptr(),
                                                      _TORCH_DT[ids.dtype], _ptr(lens), TN // N, N, ids.shape[1], h_in.data_ptr(), m.data_ptr(), logits.data_ptr(),
                                                      h_out.data_ptr(), self._stream()), self._h)
        return logits, h_out

    def low_forward_seq(self, observations, hidden, masks, subtask):
        with torch.cuda.device(self.device):
            rgb, depth, _, _, TN = self._obs(observations, False)
            h_in = self._hidden(hidden)
            N = h_in.shape[1]
            if TN % N:
                raise ValueError(f"{TN} frames is not a multiple of the hidden batch {N}")
            m = self._mask(masks, TN)
            st = self._dev(subtask, (torch.int64,)).reshape(TN)
            vel = torch.empty(TN, self.cfg.lo_actions, device=self.device, dtype=torch.float32)
            stop = torch.empty(TN, 1, device=self.device, dtype=torch.float32)
            h_out = torch.empty_like(h_in)
            _lib.check(self._lib.hcm_low_forward_seq(self._h, rgb.data_ptr(), _TORCH_DT[rgb.dtype], depth.data_ptr(), TN // N, N,
                                                     h_in.data_ptr(), m.data_ptr(), st.data_ptr(), vel.data_ptr(), stop.data_ptr(),
                                                     h_out.data_ptr(), self._stream()), self._h)
        return vel, stop, h_out

    def act(self, observations, hi_hidden, lo_hidden, masks, out=None, reuse_instruction=False, host_frames=False, gather=False):
        """reuse_instruction=True: the caller asserts that every environment's instruction is the one of the previous act() call
        (no episode ended): BERT and the instruction stream of the cross-modal block are not recomputed (hcm_act_ex).  Off in
        every parity test and in bench.py's headline number.
        host_frames=True: observations["rgb"] / ["depth"] are PINNED CPU tensors (an ObsStager's host side) and stay there: the
        library copies each to the device at the head of the encoder chain that reads it (HCM_ACT_HOST_FRAMES), overlapping the copies
        with BERT and with each other's compute; bit-identical to copying first.
        gather=True (after comm_init()): the returned record is the all-gathered (world * B, 7) record of every rank's environments, rank-major
        (hcm_act_gather: one ncclAllGather enqueued by the library behind the step)."""
        if gather and not self.comm_world:
            raise RuntimeError("act(gather=True) needs comm_init() first")
        if not gather:
            return self._act_impl(observations, hi_hidden, lo_hidden, masks, out, reuse_instruction, host_frames, False)
        # env-sharded ranks: whatever fails on THIS rank in front of the library (a malformed observation, a shape error) must not leave the peers
        # alone inside this step's all-gather -- the rank joins it with an all-NaN block (hcm_gather_poison) and raises afterwards; the peers see
        # its NaN rows and leave at the same step (rollout()).  Failures inside hcm_act_gather take the same path in the library itself.
        self._gather_called = False
        try:
            return self._act_impl(observations, hi_hidden, lo_hidden, masks, out, reuse_instruction, host_frames, True)
        except Exception:
            # did this rank's all-gather go out?  Only the library knows: hcm_act_gather returns pure argument errors (a batch beyond this engine's
            # max_batch, a communicator aborted locally) WITHOUT joining, and such a cause can be rank-local (round-5 advisor)
            joined = self._gather_called and self.query(_lib.HCM_GATHER_JOINED) == 1
            if not joined and self._gather_B and self.comm_world:
                with torch.cuda.device(self.device):
                    B = self._gather_B
                    scratch = torch.empty((1 + self.comm_world) * B, 7, device=self.device, dtype=torch.float32)
                    self._lib.hcm_gather_poison(self._h, B, scratch.data_ptr(), scratch[B:].data_ptr(), self._stream())
                    torch.cuda.current_stream().synchronize()
            raise

    def _act_impl(self, observations, hi_hidden, lo_hidden, masks, out, reuse_instruction, host_frames, gather):
        # (the per-rank batch of the collective -- every rank passes the same B -- is recorded from the VALIDATED observation of every gather call,
        #  _obs() below, so that a later call that cannot even read its observations still knows how many rows its peers expect, and a job that
        #  changes its agreed B between steps does not poison with a stale count: round-5 advisor)
        flags = (_lib.HCM_ACT_REUSE_INSTRUCTION if reuse_instruction else 0) | (_lib.HCM_ACT_HOST_FRAMES if host_frames else 0)
        if self._graph:
            cg = self._chain_graphs
            if cg == "auto":
                if host_frames:
                    # the replay enqueues the two frame copies itself, at once and outside the graphs, so they run beside BERT (B = 64: 4.87 -> 4.40 ms
                    # PCIe-inclusive); below ~4 MB a pinned host -> device copy is carried out by the calling thread behind the stream's earlier work
                    cg = min(observations["rgb"].numel() * observations["rgb"].element_size(),
                             observations["depth"].numel() * observations["depth"].element_size()) >= (4 << 20)
                else:
                    cg = int(observations["rgb"].shape[0]) <= 2
            if cg:
                flags |= _lib.HCM_ACT_CHAIN_GRAPHS
        if self._graph:
            rec, hh2, lh2 = self._act_graph(observations, hi_hidden, lo_hidden, masks, flags, gather)
            if out is not None:
                out.copy_(rec)
                rec = out
            return rec, hh2, lh2
        with torch.cuda.device(self.device):
            rgb, depth, ids, lens, B = self._obs(observations, True, host_frames)
            if gather:
                self._gather_B = B
            hh, lh, m = self._hidden(hi_hidden, B), self._hidden(lo_hidden, B), self._mask(masks, B)
            hh2, lh2 = torch.empty_like(hh), torch.empty_like(lh)
            if gather:
                local = torch.empty(B, 7, device=self.device, dtype=torch.float32)
                rec = out if out is not None else torch.empty(self.comm_world * B, 7, device=self.device, dtype=torch.float32)
                self._gather_called = True
                _lib.check(self._lib.hcm_act_gather(self._h, rgb.data_ptr(), _TORCH_DT[rgb.dtype], depth.data_ptr(), ids.data_ptr(),
                                                    _TORCH_DT[ids.dtype], _ptr(lens), B, ids.shape[1], hh.data_ptr(), lh.data_ptr(), m.data_ptr(),
                                                    local.data_ptr(), hh2.data_ptr(), lh2.data_ptr(), flags, rec.data_ptr(), self._stream()), self._h)
            else:
                rec = out if out is not None else torch.empty(B, 7, device=self.device, dtype=torch.float32)
                _lib.check(self._lib.hcm_act_ex(self._h, rgb.data_ptr(), _TORCH_DT[rgb.dtype], depth.data_ptr(), ids.data_ptr(),
                                                _TORCH_DT[ids.dtype], _ptr(lens), B, ids.shape[1], hh.data_ptr(), lh.data_ptr(), m.data_ptr(),
                                                rec.data_ptr(), hh2.data_ptr(), lh2.data_ptr(), flags, self._stream()), self._h)
            self._guard_poll(self._stream(), defer=gather)
        return rec, hh2, lh2

    def refresh_instruction(self, instruction, env_indices, instruction_lengths=None):
        """Recompute the cached instruction stream of the listed environments (those that started a new episode) from
        `instruction` (B, L) -- same L as the cached step; afterwards act(..., reuse_instruction=True) is valid again
        (hcm_refresh_instruction)."""
        idx = np.ascontiguousarray(np.asarray(env_indices, dtype=np.int32).reshape(-1))
        with torch.cuda.device(self.device):
            ids = self._dev(instruction, (torch.int64, torch.int32, torch.float32))
            B, L = ids.shape
            lens = None if instruction_lengths is None else self._dev(instruction_lengths, (torch.int32,)).reshape(-1)
            if self._graph and self._static is not None and self._static["B"] == B and self._static["ids"].dtype == ids.dtype:
                gs = self._gstream
                gs.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(gs):
                    s_ids = self._static["ids"][:B * L].view(B, L)
                    s_ids.copy_(ids, non_blocking=True)
                    if lens is not None:
                        self._static["lens"].copy_(lens, non_blocking=True)
                        lens = self._static["lens"]
                    _lib.check(self._lib.hcm_refresh_instruction(self._h, s_ids.data_ptr(), _TORCH_DT[ids.dtype], _ptr(lens), B, L,
                                                                 idx.ctypes.data_as(C.POINTER(C.c_int32)), idx.size, C.c_void_p(gs.cuda_stream)), self._h)
                torch.cuda.current_stream().wait_stream(gs)
            else:
                _lib.check(self._lib.hcm_refresh_instruction(self._h, ids.data_ptr(), _TORCH_DT[ids.dtype], _ptr(lens), B, L,
                                                             idx.ctypes.data_as(C.POINTER(C.c_int32)), idx.size, self._stream()), self._h)

    # ---- debug taps
    def enable_taps(self, on=True):
        _lib.check(self._lib.hcm_debug_enable_taps(self._h, int(on)), self._h)

    def get_tap(self, name):
        n = C.c_int64()
        shape = (C.c_int64 * 4)()
        _lib.check(self._lib.hcm_debug_get_tap(self._h, name.encode(), None, 0, C.byref(n), shape), self._h)
        buf = np.empty(n.value, dtype=np.float32)
        _lib.check(self._lib.hcm_debug_get_tap(self._h, name.encode(), buf.ctypes.data_as(C.c_void_p), n.value,
                                               C.byref(n), shape), self._h)
        return buf.reshape([d for d in shape if d > 0])


class _ModelBase:
    def __init__(self, engine: HCMEngine):
        self.engine = engine
        self.model_config = engine.cfg

    def __call__(self, batch):
        return self.forward(batch)

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    @property
    def output_size(self):
        return self.engine.cfg.hidden

    @property
    def is_blind(self):
        return False

    @property
    def num_recurrent_layers(self):
        return self.engine.num_recurrent_layers


class Seq2Seq_HighLevel_CMA(_ModelBase):
    """Drop-in for robo_vln_baselines.models.seq2seq_highlevel_cma.Seq2Seq_HighLevel_CMA (inference)."""

    def forward(self, batch):
        observations, rnn_hidden_states, prev_actions, masks = batch   # prev_actions unused (use_prev_action=False)
        # RNNStateEncoder.forward dispatch (state_encoder.py:135-137): frames == hidden batch -> single step, else sequence
        if observations["rgb"].shape[0] == rnn_hidden_states.shape[1]:
            logits, hidden = self.engine.high_forward(observations, rnn_hidden_states, masks)
        else:
            logits, hidden = self.engine.high_forward_seq(observations, rnn_hidden_states, masks)
        # the reference mutates the caller's dict (seq2seq_highlevel_cma.py:196)
        if isinstance(observations, dict) and "instruction" in observations:
            del observations["instruction"]
        return logits, hidden


class Seq2Seq_LowLevel(_ModelBase):
    """Drop-in for robo_vln_baselines.models.seq2seq_lowlevel.Seq2Seq_LowLevel (inference)."""

    def forward(self, batch):
        observations, rnn_hidden_states, prev_actions, masks, discrete_actions = batch
        if observations["rgb"].shape[0] == rnn_hidden_states.shape[1]:
            return self.engine.low_forward(observations, rnn_hidden_states, masks, discrete_actions)
        return self.engine.low_forward_seq(observations, rnn_hidden_states, masks, discrete_actions)


class Policy:
    """`act()`-shaped wrapper over the fused step (the reference has no Policy class)."""

    def __init__(self, engine: HCMEngine):
        self.engine = engine
        self.high_level = Seq2Seq_HighLevel_CMA(engine)
        self.low_level = Seq2Seq_LowLevel(engine)

    def act(self, observations, hi_hidden, lo_hidden, prev_actions, masks, deterministic=True, reuse_instruction=False, host_frames=False, gather=False):
        """-> (record (B,7) = [4 sub-task logits, lin_vel, ang_vel, stop logit], hi_hidden', lo_hidden'); gather=True (env-sharded ranks, after
        engine.comm_init()): the record of ALL ranks' environments, (world * B, 7)."""
        return self.engine.act(observations, hi_hidden, lo_hidden, masks, reuse_instruction=reuse_instruction, host_frames=host_frames, gather=gather)

    def get_value(self, *a, **k):
        """Imitation-learned agent: the reference has no critic / value head anywhere."""
        return None
