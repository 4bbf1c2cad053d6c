"""Checkpoint reader (hierarchical_trainer.py:349-363 format): round trip, tolerant un-pickling of a yacs config,
ignored BERT buffer keys; and strict loading of the result into libhcm (no GPU needed up to hcm_finalize)."""
import ctypes as C
import io
import sys
import types

import numpy as np
import pytest
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
    st = _to_struct(cfg, 2, "fp16", True, True)
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


GOLD = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden")


def _strict_load(cfg, hi_sd, lo_sd):
    l = _lib.lib()
    st = _to_struct(cfg, 2, "fp16", True, True)
    h = C.c_void_p()
    assert l.hcm_create(C.byref(st), C.byref(h)) == 0
    try:
        for model, sd in ((_lib.HCM_HIGH, hi_sd), (_lib.HCM_LOW, lo_sd)):
            for k, v in sd.items():
                a, dt = _np32(v)
                shp = (C.c_int64 * max(1, a.ndim))(*a.shape)
                assert l.hcm_load_tensor(h, model, k.encode(), a.ctypes.data_as(C.c_void_p), dt, shp, a.ndim) == 0, (k, l.hcm_last_error(h))
        # nothing is missing either: the only failure finalize can still report without a GPU is a HIP one
        rc = l.hcm_finalize(h)
        assert rc != -3, l.hcm_last_error(h)
    finally:
        l.hcm_destroy(h)


def test_reference_written_checkpoint_structure():
    """A checkpoint dict written by the REAL reference modules' state_dict() (oracle/gen_checkpoint_fixture.py; values hollowed
    out, structure intact: 721 + 497 keys in the reference's order, BertEmbeddings buffers, num_batches_tracked scalars, OrderedDict
    metadata, a config object of a class that cannot be imported here) goes through load_checkpoint and libhcm's strict loader
    at the default full-size configuration."""
    import os
    hi, lo, conf = checkpoint.load_checkpoint(os.path.join(GOLD, "ref_checkpoint_structure.pth"))
    cfg = HCMConfig().validate()
    assert len(lo) == 497 and set(lo) == {k for k, *_ in synth.low_level_spec(cfg)}
    assert set(hi) == {k for k, *_ in synth.high_level_spec(cfg)}
    assert hi["embedding_layer.embeddings.word_embeddings.weight"].shape == (30522, 768)
    assert hi["rgb_encoder.cnn.bn1.num_batches_tracked"].dim() == 0
    # the config came back as an inert stand-in that still carries the reference's fields
    assert type(conf).__module__ != "builtins" and conf["STATE_ENCODER"]["rnn_type"] == "LSTM" and conf["VISUAL_LING_ATTN"]["N"] == 1
    _strict_load(cfg, hi, lo)


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/robo_vln_baselines"), reason="needs the reference checkout (build container only)")
def test_checkpoint_saved_from_reference_modules_roundtrips_exactly():
    """Container-only: the imported reference models (synthetic weights loaded with strict=True) are saved exactly as
    RoboDaggerTrainer.save_checkpoint does (hierarchical_trainer.py:349-363) and read back: every value identical, strict load ok."""
    from oracle import ref_shims
    cfg = HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=1).validate()
    hi_sd, lo_sd = synth.make_weights(cfg, seed=2)
    hi, lo = ref_shims.build_models(cfg, hi_sd, lo_sd)
    buf = io.BytesIO()
    torch.save({"high_level_state_dict": hi.state_dict(), "low_level_state_dict": lo.state_dict(), "config": ref_shims.model_config(cfg)}, buf)
    buf.seek(0)
    hi2, lo2, conf = checkpoint.load_checkpoint(buf)
    for got, ref in ((hi2, hi.state_dict()), (lo2, lo.state_dict())):
        keys = [k for k in ref if not k.endswith(checkpoint.IGNORED_SUFFIXES)]
        assert list(got) == keys
        for k in keys:
            assert torch.equal(got[k], ref[k]), k
    assert torch.equal(hi2["linear.weight"], torch.from_numpy(hi_sd["linear.weight"]))
    _strict_load(cfg, hi2, lo2)


def test_untrusted_pickle_cannot_run_code():
    """find_class resolves only the allow-listed tensor / container globals: a __reduce__ payload naming any other callable is
    turned into an inert stand-in instead of being called."""
    import os
    marker = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"hcm_pwned_{os.getpid()}")

    class Evil:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))
    buf = io.BytesIO()
    torch.save({"high_level_state_dict": {}, "low_level_state_dict": {}, "config": Evil()}, buf)
    buf.seek(0)
    hi, lo, conf = checkpoint.load_checkpoint(buf)
    assert not os.path.exists(marker)
    assert isinstance(conf, dict) and type(conf).__name__ == "system"


def test_ddppo_depth_checkpoint_remap_on_the_reference_written_fixture():
    """VlnResnetDepthEncoder.__init__'s DDPPO remap (models/encoders/resnet_encoders.py:38-52) on tests/golden/ref_ddppo_structure.pth -- a hollow
    DDPPO checkpoint whose visual-encoder names come from the reference class and which the REAL reference constructor loaded strictly when
    oracle/gen_ddppo_fixture.py wrote it: the remapped names are exactly the 162 `depth_encoder.visual_encoder.*` names of the trainer-checkpoint
    fixture (also written by the reference modules), shapes equal, the non-encoder keys are gone."""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    dd = checkpoint.load_ddppo_depth_weights(os.path.join(here, "golden", "ref_ddppo_structure.pth"))
    hi, lo, _ = checkpoint.load_checkpoint(os.path.join(here, "golden", "ref_checkpoint_structure.pth"))
    pre = "depth_encoder.visual_encoder."
    for sd in (hi, lo):
        own = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        assert len(own) == 162 and set(own) == set(dd)
        assert all(tuple(own[k].shape) == tuple(dd[k].shape) and own[k].dtype == dd[k].dtype for k in own)
        out = checkpoint.apply_ddppo_depth_weights(sd, dd)
        assert list(out) == list(sd)                                     # same keys, same order; only the trunk's tensors replaced
    assert not any("state_encoder" in k or "critic" in k or "prev_action" in k for k in dd)


def test_ddppo_depth_weights_values_and_strictness():
    """Values travel unchanged through the remap, and the load is strict like `load_state_dict(strict=True)` (:49): a missing, an unexpected or a
    mis-shaped tensor raises RuntimeError naming it."""
    cfg = HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=1, bert_vocab=512)
    hi_sd, lo_sd = synth.make_weights(cfg, seed=5)
    other_hi, _ = synth.make_weights(cfg, seed=6)
    pre = "depth_encoder.visual_encoder."
    sd = {"actor_critic.net.visual_encoder." + k[len(pre):]: torch.from_numpy(np.asarray(v)) for k, v in other_hi.items() if k.startswith(pre)}
    sd["actor_critic.net.state_encoder.rnn.bias_ih_l0"] = torch.zeros(8)
    sd["actor_critic.critic.fc.weight"] = torch.zeros(1, 512)
    buf = io.BytesIO()
    torch.save({"state_dict": sd, "config": {"x": 1}}, buf)
    buf.seek(0)
    dd = checkpoint.load_ddppo_depth_weights(buf)
    for tgt in (hi_sd, lo_sd):
        out = checkpoint.apply_ddppo_depth_weights(tgt, dd)
        for k in tgt:
            want = other_hi[k] if k.startswith(pre) else tgt[k]
            assert np.array_equal(np.asarray(out[k]), np.asarray(want)), k
    k0 = next(iter(dd))
    short = {k: v for k, v in dd.items() if k != k0}
    with pytest.raises(RuntimeError, match="missing"):
        checkpoint.apply_ddppo_depth_weights(hi_sd, short)
    with pytest.raises(RuntimeError, match="unexpected"):
        checkpoint.apply_ddppo_depth_weights(hi_sd, {**dd, "backbone.not_a_layer.weight": torch.zeros(1)})
    with pytest.raises(RuntimeError, match="size mismatch"):
        checkpoint.apply_ddppo_depth_weights(hi_sd, {**dd, k0: torch.zeros(3, 3)})
    buf = io.BytesIO()
    torch.save({"model": {}}, buf)
    buf.seek(0)
    with pytest.raises(KeyError):
        checkpoint.load_ddppo_depth_weights(buf)
