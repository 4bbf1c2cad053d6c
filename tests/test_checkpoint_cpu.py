"""Checkpoint reader (hierarchical_trainer.py:349-363 format): round trip, tolerant un-pickling of a yacs config,
ignored BERT buffer keys; and strict loading of the result into libhcm (no GPU needed up to hcm_finalize)."""
import ctypes as C
import io
import sys
import types

import numpy as np
import torch

from robo_vln_amd import _lib, checkpoint, synth
from robo_vln_amd.config import HCMConfig
from robo_vln_amd.policy import _to_struct, _np32


def test_checkpoint_roundtrip_and_strict_load():
    cfg = HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=1, bert_vocab=512)
    hi_sd, lo_sd = synth.make_weights(cfg, seed=5)
    # a config object of a class whose module will be gone at load time (as yacs is here)
    mod = types.ModuleType("yacs_fake.config")

    CfgNode = type("CfgNode", (dict,), {"__module__": "yacs_fake.config", "__qualname__": "CfgNode"})
    mod.CfgNode = CfgNode
    sys.modules["yacs_fake"] = types.ModuleType("yacs_fake")
    sys.modules["yacs_fake.config"] = mod
    conf = CfgNode(MODEL=dict(STATE_ENCODER=dict(rnn_type="LSTM")))
    hi_t = {k: torch.from_numpy(np.asarray(v)) for k, v in hi_sd.items()}
    hi_t["embedding_layer.embeddings.position_ids"] = torch.arange(512).unsqueeze(0)     # old-transformers buffer
    buf = io.BytesIO()
    checkpoint.save_checkpoint(buf, hi_t, lo_sd, conf)
    del sys.modules["yacs_fake.config"], sys.modules["yacs_fake"]
    buf.seek(0)
    hi2, lo2, conf2 = checkpoint.load_checkpoint(buf)
    assert set(hi2) == set(hi_sd) and set(lo2) == set(lo_sd)
    assert "embedding_layer.embeddings.position_ids" not in hi2
    assert torch.equal(hi2["linear.weight"], torch.from_numpy(hi_sd["linear.weight"]))
    assert conf2["MODEL"]["STATE_ENCODER"]["rnn_type"] == "LSTM"
    # every tensor of the loaded dicts is accepted by the library's strict loader
    l = _lib.lib()
    st = _to_struct(cfg, 2, "bf16", True, True)
    h = C.c_void_p()
    assert l.hcm_create(C.byref(st), C.byref(h)) == 0
    try:
        for model, sd in ((_lib.HCM_HIGH, hi2), (_lib.HCM_LOW, lo2)):
            for k, v in sd.items():
                a, dt = _np32(v)
                shp = (C.c_int64 * max(1, a.ndim))(*a.shape)
                assert l.hcm_load_tensor(h, model, k.encode(), a.ctypes.data_as(C.c_void_p), dt, shp, a.ndim) == 0, (k, l.hcm_last_error(h))
    finally:
        l.hcm_destroy(h)


def test_missing_state_dict_key_is_reported():
    buf = io.BytesIO()
    torch.save({"high_level_state_dict": {}}, buf)
    buf.seek(0)
    try:
        checkpoint.load_checkpoint(buf)
    except KeyError as e:
        assert "low_level_state_dict" in str(e)
    else:
        raise AssertionError("expected KeyError")
