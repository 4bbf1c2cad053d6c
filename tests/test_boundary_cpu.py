"""CPU-side checks of the drop-in boundary: libhcm.so loads, exports every symbol include/hcm.h declares,
validates configs / state_dict keys / shapes with the reference's error behaviour.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from robo_vln_amd import _lib, synth
from robo_vln_amd.config import HCMConfig, baseline_config
from robo_vln_amd.policy import _to_struct

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "hcm.h")).read()
    declared = set(re.findall(r"\b(hcm_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 17
    l = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(l, name), f"libhcm.so does not export {name}"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)


def _create(cfg, **over):
    l = _lib.lib()
    st = _to_struct(cfg, 4, "fp16", True, True)
    for k, v in over.items():
        setattr(st, k, v)
    h = C.c_void_p()
    rc = l.hcm_create(C.byref(st), C.byref(h))
    return l, rc, h


def test_create_rejects_broken_reference_flags():
    l, rc, h = _create(HCMConfig(), use_prev_action=1)
    assert rc == -6 and b"use_prev_action" in l.hcm_last_error(None)
    l, rc, h = _create(HCMConfig(), ablate_instruction=1)
    assert rc == -6
    l, rc, h = _create(HCMConfig(), progress_monitor=1)
    assert rc == -6
    # SimpleCNN encoders cannot build the high-level model (reference: AttributeError output_shape)
    l, rc, h = _create(HCMConfig(), depth_encoder=_lib.HCM_ENC_SIMPLECNN)
    assert rc == -6 and b"output_shape" in l.hcm_last_error(None)
    with pytest.raises(ValueError):
        HCMConfig(use_prev_action=True).validate()
    # depth frames: any multiple of 64 (habitat's ResNetEncoder sizes its compression conv from (H/2)/32; 192 -> 3x3 x 228 channels);
    # other sizes give a final map the reference's own visual_fc / spatial embedding shapes do not match, and are rejected up front
    l, rc, h = _create(HCMConfig(), depth_h=224, depth_w=224)
    assert rc == -6 and b"depth frame size" in l.hcm_last_error(None)
    with pytest.raises(ValueError):
        HCMConfig(depth_hw=224).validate()
    c192 = HCMConfig(depth_hw=192).validate()
    assert c192.depth_final_spatial() == 3 and c192.depth_compress_channels() == 228
    # RGB frames: any H x W >= 32 with the torchvision trunk (adaptive pools); depth stays square (habitat sizes the encoder from the height),
    # SimpleRGBCNN is built for square frames only
    l, rc, h = _create(HCMConfig(), depth_h=256, depth_w=192)
    assert rc == -6 and b"square" in l.hcm_last_error(None)
    with pytest.raises(ValueError):
        HCMConfig(depth_hw=256, depth_w=192).validate()
    assert HCMConfig(depth_encoder="SimpleDepthCNN", rgb_encoder="SimpleRGBCNN", depth_hw=152, depth_w=218).validate().depth_shape == (152, 218)
    l, rc, h = _create(HCMConfig(rgb_encoder="SimpleRGBCNN", depth_encoder="SimpleDepthCNN"), rgb_h=160, rgb_w=32, build_high=0)
    assert rc == -6 and b"SimpleRGBCNN" in l.hcm_last_error(None)
    assert HCMConfig(rgb_encoder="SimpleRGBCNN", rgb_hw=160, rgb_w=224).validate().rgb_shape == (160, 224)
    assert HCMConfig(rgb_hw=160, rgb_w=224).validate().rgb_shape == (160, 224) and HCMConfig().rgb_shape == (256, 256)


def test_strict_state_dict_keys_and_shapes():
    cfg = HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=1)
    l, rc, h = _create(cfg)
    assert rc == 0
    try:
        a = np.zeros((4, 512), np.float32)
        shp = (C.c_int64 * 2)(4, 512)
        assert l.hcm_load_tensor(h, _lib.HCM_HIGH, b"linear.weight", a.ctypes.data_as(C.c_void_p), _lib.HCM_F32, shp, 2) == 0
        assert l.hcm_load_tensor(h, _lib.HCM_HIGH, b"linear.wieght", a.ctypes.data_as(C.c_void_p), _lib.HCM_F32, shp, 2) == -3
        assert b"Unexpected key" in l.hcm_last_error(h)
        bad = (C.c_int64 * 2)(5, 512)
        assert l.hcm_load_tensor(h, _lib.HCM_HIGH, b"linear.weight", a.ctypes.data_as(C.c_void_p), _lib.HCM_F32, bad, 2) == -4
        assert b"size mismatch" in l.hcm_last_error(h)
        # finalize with missing keys fails before touching the GPU
        assert l.hcm_finalize(h) == -3
        assert b"Missing key" in l.hcm_last_error(h)
        out = C.c_int64()
        assert l.hcm_query(h, _lib.HCM_NUM_RECURRENT_LAYERS, C.byref(out)) == 0 and out.value == 2
        assert l.hcm_query(h, _lib.HCM_RECORD_WIDTH, C.byref(out)) == 0 and out.value == 7
        # forward before finalize
        assert l.hcm_act(h, None, 0, None, None, 0, None, 1, 20, None, None, None, None, None, None, None) == -2
    finally:
        l.hcm_destroy(h)


def test_every_synth_key_is_accepted_by_the_library():
    """The C++ spec (weights.cpp) and the Python spec (synth.py, validated against the imported reference by
    oracle/gen_golden.py with strict=True) must agree key-for-key and shape-for-shape."""
    for cfg, which in ((HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=2, vla_layers=2), "both"),
                       (HCMConfig(depth_encoder="SimpleDepthCNN", rgb_encoder="SimpleRGBCNN"), "lo"),
                       (HCMConfig(rnn_type="GRU", bert_layers=1), "both")):
        l = _lib.lib()
        st = _to_struct(cfg, 2, "fp32", which == "both", True)
        h = C.c_void_p()
        assert l.hcm_create(C.byref(st), C.byref(h)) == 0, l.hcm_last_error(None)
        try:
            specs = [(_lib.HCM_LOW, synth.low_level_spec(cfg))]
            if which == "both":
                specs.append((_lib.HCM_HIGH, synth.high_level_spec(cfg)))
            for model, spec in specs:
                for key, shape, kind, aux in spec:
                    n = int(np.prod(shape)) if len(shape) else 1
                    if n > 4_000_000:      # skip the copy of the huge tables; check key+shape via a wrong-dtype probe
                        a = np.zeros(1, np.float32)
                        shp = (C.c_int64 * max(1, len(shape)))(*shape)
                        rc = l.hcm_load_tensor(h, model, key.encode(), a.ctypes.data_as(C.c_void_p), _lib.HCM_U8, shp, len(shape))
                        assert rc == -1, (key, rc, l.hcm_last_error(h))   # key+shape accepted, dtype rejected
                        continue
                    a = np.zeros(shape, np.int64 if kind == "nbt" else np.float32)
                    shp = (C.c_int64 * max(1, len(shape)))(*shape)
                    rc = l.hcm_load_tensor(h, model, key.encode(), a.ctypes.data_as(C.c_void_p),
                                           _lib.HCM_I64 if kind == "nbt" else _lib.HCM_F32, shp, len(shape))
                    assert rc == 0, (key, l.hcm_last_error(h))
        finally:
            l.hcm_destroy(h)


def test_cma_option_branches_are_accepted_and_the_unbuilt_ones_cite_the_reference():
    """Round 6: CMANet's ablation flags (cma.py:236-241) and INSTRUCTION_ENCODER.rnn_type = "GRU" (instruction_encoder.py:42) are built; the GRU
    encoder's state_dict has three gates; final_state_only is accepted and ignored as in the reference (cma.py:32 overwrites it); use_prev_action /
    rcm_state_encoder stay rejected with the reference lines in the message."""
    from robo_vln_amd.config import CMAConfig
    from robo_vln_amd.cma import _to_struct as cma_struct
    for kw in (dict(ablate_instruction=True), dict(ablate_depth=True), dict(ablate_rgb=True), dict(instr_rnn="GRU"), dict(final_state_only=True)):
        CMAConfig(rgb_hw=128, depth_hw=128, instr_len=12, **kw).validate()
    for kw in (dict(use_prev_action=True), dict(rcm_state_encoder=True)):
        with pytest.raises(ValueError, match="default.py:211-212"):
            CMAConfig(**kw).validate()
    with pytest.raises(ValueError):
        CMAConfig(instr_rnn="RNN").validate()
    cfg = CMAConfig(rgb_hw=128, depth_hw=128, instr_len=12, instr_rnn="GRU", ablate_rgb=True).validate()
    spec = {k: shape for k, shape, _, _ in synth.cma_spec(cfg)}
    assert spec["instruction_encoder.encoder_rnn.weight_ih_l0"] == (3 * cfg.instr_hidden, cfg.embedding_size)
    assert spec["instruction_encoder.encoder_rnn.weight_hh_l0_reverse"] == (3 * cfg.instr_hidden, cfg.instr_hidden)
    l = _lib.lib()
    st = cma_struct(cfg, 2, "fp32")
    assert st.instr_rnn == _lib.HCM_GRU and st.ablate_rgb == 1 and st.ablate_depth == 0
    h = C.c_void_p()
    assert l.hcm_cma_create(C.byref(st), C.byref(h)) == 0, l.hcm_last_error(None)
    try:
        for key, shape, kind, aux in synth.cma_spec(cfg):          # the C++ spec agrees key for key (three-gate instruction encoder)
            n = int(np.prod(shape)) if len(shape) else 1
            if n > 4_000_000:
                continue
            a = np.zeros(shape, np.int64 if kind == "nbt" else np.float32)
            shp = (C.c_int64 * max(1, len(shape)))(*shape)
            assert l.hcm_load_tensor(h, _lib.HCM_CMA, key.encode(), a.ctypes.data_as(C.c_void_p), _lib.HCM_I64 if kind == "nbt" else _lib.HCM_F32, shp, len(shape)) == 0, \
                (key, l.hcm_last_error(h))
    finally:
        l.hcm_destroy(h)
    st.instr_rnn = 7
    assert l.hcm_cma_create(C.byref(st), C.byref(h)) == -1 and b"INSTRUCTION_ENCODER.rnn_type" in l.hcm_last_error(None)


def test_baseline_configs():
    assert baseline_config(1).instr_len == 80 and baseline_config(1).rgb_hw == 256
    assert baseline_config(0).vla_layers == 2 and baseline_config(4).instr_len == 160
    assert HCMConfig(depth_hw=128).depth_compress_channels() == 512 and HCMConfig().depth_compress_channels() == 128
