"""Deterministic synthetic weights and inputs for the HCM hot path.

There is no network for checkpoints, so parity and bench runs use random-init
weights of the reference architecture.  The generator is a counter-based hash
(splitmix64 over `fnv1a(key) ^ seed`, element index as counter) implemented with
numpy integer arithmetic only, so the container (where the reference is imported
to make goldens) and the GPU box regenerate bit-identical tensors without
depending on any library RNG stream.

Key names and shapes restate the reference's two state_dicts
(/root/reference/robo_vln_baselines/hierarchical_trainer.py:358-362; SURVEY.md
Appendix B).  `oracle/gen_golden.py` checks them with
`load_state_dict(strict=True)` on the imported reference modules.
"""
import math
import numpy as np

from .config import HCMConfig

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(s: str) -> int:
    h = 0xCBF29CE484222325
    for ch in s.encode():
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(key: str, n: int, seed: int = 0) -> np.ndarray:
    """n floats in [0,1) with 24 random bits each, a pure function of (key, seed, index)."""
    base = np.uint64(_fnv1a(key) ^ ((seed * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + base
    bits = _splitmix64(ctr) >> np.uint64(40)
    return (bits.astype(np.float64) * (1.0 / 16777216.0)).astype(np.float32)


def randint(key: str, n: int, lo: int, hi: int, seed: int = 0) -> np.ndarray:
    """n ints uniform in [lo, hi)."""
    u = uniform01(key, n, seed).astype(np.float64)
    return (lo + np.floor(u * (hi - lo))).astype(np.int64)


# --------------------------------------------------------------------------------------
# tensor specs: (key, shape, kind, aux)
#   kind 'w'    : U(-a, a), a = gain * sqrt(3 / fan_in)         aux = (fan_in, gain)
#   kind 'b'    : U(-0.1, 0.1)
#   kind 'gamma': U(0.5, 1.5)        (LN / BN / GN scale)
#   kind 'beta' : U(-0.1, 0.1)
#   kind 'rmean': U(-0.1, 0.1)       (BN running_mean)
#   kind 'rvar' : U(0.5, 1.5)        (BN running_var)
#   kind 'emb'  : U(-s, s)           aux = s
#   kind 'nbt'  : int64 scalar 0     (BN num_batches_tracked)
# --------------------------------------------------------------------------------------

RELU_GAIN = math.sqrt(2.0)   # keeps activation scale through conv+ReLU stacks


def _conv(key, cout, cin, kh, kw, gain=RELU_GAIN):
    return [(key, (cout, cin, kh, kw), "w", (cin * kh * kw, gain))]


def _linear(prefix, out_f, in_f, gain=1.0, bias=True):
    s = [(prefix + ".weight", (out_f, in_f), "w", (in_f, gain))]
    if bias:
        s.append((prefix + ".bias", (out_f,), "b", None))
    return s


def _norm(prefix, c):
    return [(prefix + ".weight", (c,), "gamma", None), (prefix + ".bias", (c,), "beta", None)]


def _bn(prefix, c):
    return _norm(prefix, c) + [
        (prefix + ".running_mean", (c,), "rmean", None),
        (prefix + ".running_var", (c,), "rvar", None),
        (prefix + ".num_batches_tracked", (), "nbt", None),
    ]


RESNET50_BLOCKS = (3, 4, 6, 3)


def torchvision_resnet50_spec(prefix, with_fc):
    """torchvision `resnet50` (v1.5: stride on the 3x3) key layout [SURVEY Appendix C]."""
    s = _conv(prefix + "conv1.weight", 64, 3, 7, 7) + _bn(prefix + "bn1", 64)
    inpl = 64
    for li, nb in enumerate(RESNET50_BLOCKS):
        planes = 64 << li
        for bi in range(nb):
            p = f"{prefix}layer{li + 1}.{bi}."
            s += _conv(p + "conv1.weight", planes, inpl, 1, 1) + _bn(p + "bn1", planes)
            s += _conv(p + "conv2.weight", planes, planes, 3, 3) + _bn(p + "bn2", planes)
            # last conv of a residual branch: small gain so the trunk does not blow up
            s += _conv(p + "conv3.weight", planes * 4, planes, 1, 1, gain=0.5) + _bn(p + "bn3", planes * 4)
            if bi == 0:
                s += _conv(p + "downsample.0.weight", planes * 4, inpl, 1, 1, gain=1.0)
                s += _bn(p + "downsample.1", planes * 4)
            inpl = planes * 4
    if with_fc:
        s += _linear(prefix + "fc", 1000, 2048)
    return s


def habitat_gn_resnet50_spec(prefix, in_ch, base, compress_ch):
    """habitat DDPPO `ResNetEncoder` with GroupNorm-ResNet50 backbone [SURVEY Appendix C]."""
    bb = prefix + "backbone."
    s = _conv(bb + "conv1.0.weight", base, in_ch, 7, 7) + _norm(bb + "conv1.1", base)
    inpl = base
    for li, nb in enumerate(RESNET50_BLOCKS):
        planes = base << li
        for bi in range(nb):
            p = f"{bb}layer{li + 1}.{bi}."
            s += _conv(p + "convs.0.weight", planes, inpl, 1, 1) + _norm(p + "convs.1", planes)
            s += _conv(p + "convs.3.weight", planes, planes, 3, 3) + _norm(p + "convs.4", planes)
            s += _conv(p + "convs.6.weight", planes * 4, planes, 1, 1) + _norm(p + "convs.7", planes * 4)
            if bi == 0:
                s += _conv(p + "downsample.0.weight", planes * 4, inpl, 1, 1) + _norm(p + "downsample.1", planes * 4)
            inpl = planes * 4
    s += _conv(prefix + "compression.0.weight", compress_ch, inpl, 3, 3) + _norm(prefix + "compression.1", compress_ch)
    return s


def simple_cnn_spec(prefix, in_ch, hw, out_f):
    """habitat SimpleCNN as used by simple_cnns.py:51-101 (cnn.{0,2,4,7})."""
    dh, dw = (hw, hw) if isinstance(hw, int) else hw          # simple_cnns.py:63-73: the two dimensions on their own
    for k, st in ((8, 4), (4, 2), (3, 1)):
        dh, dw = (dh - k) // st + 1, (dw - k) // st + 1
    s = [(prefix + "cnn.0.weight", (32, in_ch, 8, 8), "w", (in_ch * 64, RELU_GAIN)), (prefix + "cnn.0.bias", (32,), "b", None)]
    s += [(prefix + "cnn.2.weight", (64, 32, 4, 4), "w", (32 * 16, RELU_GAIN)), (prefix + "cnn.2.bias", (64,), "b", None)]
    s += [(prefix + "cnn.4.weight", (32, 64, 3, 3), "w", (64 * 9, 1.0)), (prefix + "cnn.4.bias", (32,), "b", None)]
    s += _linear(prefix + "cnn.7", out_f, 32 * dh * dw, gain=RELU_GAIN)
    return s


def bert_spec(prefix, cfg: HCMConfig):
    h, it = cfg.bert_hidden, cfg.bert_inter
    e = prefix + "embeddings."
    s = [
        (e + "word_embeddings.weight", (cfg.bert_vocab, h), "emb", 0.05),
        (e + "position_embeddings.weight", (cfg.bert_max_pos, h), "emb", 0.05),
        (e + "token_type_embeddings.weight", (2, h), "emb", 0.05),
    ] + _norm(e + "LayerNorm", h)
    for i in range(cfg.bert_layers):
        p = f"{prefix}encoder.layer.{i}."
        s += _linear(p + "attention.self.query", h, h, gain=2.0)   # gain>1: non-degenerate softmax
        s += _linear(p + "attention.self.key", h, h, gain=2.0)
        s += _linear(p + "attention.self.value", h, h)
        s += _linear(p + "attention.output.dense", h, h)
        s += _norm(p + "attention.output.LayerNorm", h)
        s += _linear(p + "intermediate.dense", it, h, gain=RELU_GAIN)
        s += _linear(p + "output.dense", h, it)
        s += _norm(p + "output.LayerNorm", h)
    s += _linear(prefix + "pooler.dense", h, h)
    return s


def vla_spec(prefix, cfg: HCMConfig, vis_in=None):
    d, ff = cfg.d_model, cfg.d_ff
    vis_in = cfg.vis_in if vis_in is None else vis_in
    s = []
    for i in range(cfg.vla_layers):
        p = f"{prefix}layers.{i}."
        a = p + "enc_att.attention."
        s += _linear(a + "fc_q", d, d, gain=2.0) + _linear(a + "fc_k", d, d, gain=2.0)
        s += _linear(a + "fc_v", d, d) + _linear(a + "fc_o", d, d)
        s += _norm(p + "enc_att.layer_norm", d)
        s += _linear(p + "pwff.fc1", ff, d, gain=RELU_GAIN) + _linear(p + "pwff.fc2", d, ff)
        s += _norm(p + "pwff.layer_norm", d)
    s += _linear(prefix + "vis_fc", d, vis_in, gain=RELU_GAIN)
    s += _linear(prefix + "ins_fc", d, cfg.ins_in, gain=RELU_GAIN)
    s += _norm(prefix + "layer_norm", d)
    return s


def rnn_spec(prefix, cfg: HCMConfig, in_f):
    g = 4 if cfg.rnn_type == "LSTM" else 3
    hs = cfg.hidden
    return [
        (prefix + "weight_ih_l0", (g * hs, in_f), "w", (in_f, 1.0)),
        (prefix + "weight_hh_l0", (g * hs, hs), "w", (hs, 1.0)),
        (prefix + "bias_ih_l0", (g * hs,), "b", None),
        (prefix + "bias_hh_l0", (g * hs,), "b", None),
    ]


def hi_rnn_input_size(cfg: HCMConfig):
    # seq2seq_highlevel_cma.py:120-124
    return cfg.cm_d_model * 2 + cfg.depth_out + cfg.rgb_out


def lo_rnn_input_size(cfg: HCMConfig):
    # seq2seq_lowlevel.py:79-83 (sub_task_embedding dim 32)
    return cfg.depth_out + cfg.rgb_out + 32


def high_level_spec(cfg: HCMConfig):
    """Seq2Seq_HighLevel_CMA state_dict (seq2seq_highlevel_cma.py:33-141)."""
    cfg.validate()
    if cfg.rgb_encoder != "TorchVisionResNet50" or cfg.depth_encoder != "VlnResnetDepthEncoder":
        # the reference ctor raises AttributeError (no `output_shape`) for SimpleCNN encoders
        raise ValueError("Seq2Seq_HighLevel_CMA needs TorchVisionResNet50 + VlnResnetDepthEncoder "
                         "(SimpleCNN encoders have no output_shape; seq2seq_highlevel_cma.py:87,:96)")
    fs = cfg.depth_final_spatial()
    cc = cfg.depth_compress_channels()
    dC = cc + 64
    rC = 2048 + 64
    s = bert_spec("embedding_layer.", cfg)
    s += _linear("ins_fc", 256, 768)     # TRANSFORMER_INSTRUCTION_ENCODER d_in/d_model; unused in forward
    s += habitat_gn_resnet50_spec("depth_encoder.visual_encoder.", 1, cfg.depth_baseplanes, cc)
    s += [("depth_encoder.spatial_embeddings.weight", (fs * fs, 64), "emb", 0.5)]
    s += torchvision_resnet50_spec("rgb_encoder.cnn.", with_fc=False)
    s += [("rgb_encoder.spatial_embeddings.weight", (16, 64), "emb", 0.5)]
    s += _linear("rgb_linear.2", cfg.rgb_out, rC, gain=RELU_GAIN)
    s += _linear("depth_linear.1", cfg.depth_out, dC * fs * fs, gain=RELU_GAIN)
    s += [("rgb_kv.weight", (cfg.vis_in, rC, 1), "w", (rC, 1.0)), ("rgb_kv.bias", (cfg.vis_in,), "b", None)]
    s += [("depth_kv.weight", (cfg.vis_in, dC, 1), "w", (dC, 1.0)), ("depth_kv.bias", (cfg.vis_in,), "b", None)]
    s += vla_spec("image_cm_encoder.", cfg)
    s += rnn_spec("state_encoder.rnn.", cfg, hi_rnn_input_size(cfg))
    s += _linear("progress_monitor", 1, cfg.hidden)
    s += _linear("linear", cfg.num_actions, cfg.hidden)
    return s


def low_level_spec(cfg: HCMConfig):
    """Seq2Seq_LowLevel state_dict (seq2seq_lowlevel.py:32-98)."""
    cfg.validate()
    s = []
    if cfg.depth_encoder == "VlnResnetDepthEncoder":
        fs = cfg.depth_final_spatial()
        cc = cfg.depth_compress_channels()
        s += habitat_gn_resnet50_spec("depth_encoder.visual_encoder.", 1, cfg.depth_baseplanes, cc)
        s += _linear("depth_encoder.visual_fc.1", cfg.depth_out, cc * fs * fs, gain=RELU_GAIN)
    else:
        s += simple_cnn_spec("depth_encoder.", 1, cfg.depth_shape, cfg.depth_out)
    if cfg.rgb_encoder == "TorchVisionResNet50":
        s += torchvision_resnet50_spec("rgb_encoder.cnn.", with_fc=True)
        s += _linear("rgb_encoder.fc", cfg.rgb_out, 2048, gain=RELU_GAIN)
    else:
        s += simple_cnn_spec("rgb_encoder.", 3, cfg.rgb_shape, cfg.rgb_out)
    s += [("sub_task_embedding.weight", (cfg.num_sub_tasks + 1, 32), "emb", 1.0)]
    s += rnn_spec("state_encoder.rnn.", cfg, lo_rnn_input_size(cfg))
    s += _linear("progress_monitor", 1, cfg.hidden)
    s += _linear("linear", cfg.lo_actions, cfg.hidden)
    s += _linear("stop_linear", 1, cfg.hidden)
    return s


def _rnn_generic(prefix, gates, hs, in_f, suffix=""):
    return [
        (prefix + "weight_ih_l0" + suffix, (gates * hs, in_f), "w", (in_f, 1.0)),
        (prefix + "weight_hh_l0" + suffix, (gates * hs, hs), "w", (hs, 1.0)),
        (prefix + "bias_ih_l0" + suffix, (gates * hs,), "b", None),
        (prefix + "bias_hh_l0" + suffix, (gates * hs,), "b", None),
    ]


def cma_spec(cfg):
    """CMANet state_dict (models/cma.py:28-186; InstructionEncoder instruction_encoder.py:9-47)."""
    cfg.validate()
    fs = cfg.depth_final_spatial()
    cc = cfg.depth_compress_channels()
    dC, rC = cc + 64, 2048 + 64
    hh = cfg.hidden // 2
    g = 4 if cfg.rnn_type == "LSTM" else 3
    s = [("instruction_encoder.embedding_layer.weight", (cfg.vocab_size, cfg.embedding_size), "emb", 1.0)]
    gi = 4 if cfg.instr_rnn == "LSTM" else 3              # INSTRUCTION_ENCODER.rnn_type (instruction_encoder.py:42)
    s += _rnn_generic("instruction_encoder.encoder_rnn.", gi, cfg.instr_hidden, cfg.embedding_size)
    if cfg.bidirectional:
        s += _rnn_generic("instruction_encoder.encoder_rnn.", gi, cfg.instr_hidden, cfg.embedding_size, "_reverse")
    s += habitat_gn_resnet50_spec("depth_encoder.visual_encoder.", 1, cfg.depth_baseplanes, cc)
    s += [("depth_encoder.spatial_embeddings.weight", (fs * fs, 64), "emb", 0.5)]
    s += torchvision_resnet50_spec("rgb_encoder.cnn.", with_fc=False)
    s += [("rgb_encoder.spatial_embeddings.weight", (16, 64), "emb", 0.5)]
    s += _linear("rgb_linear.2", cfg.rgb_out, rC, gain=RELU_GAIN)
    s += _linear("depth_linear.1", cfg.depth_out, dC * fs * fs, gain=RELU_GAIN)
    s += _rnn_generic("state_encoder.rnn.", g, cfg.hidden, cfg.rgb_out + cfg.depth_out)
    s += [("rgb_kv.weight", (hh + cfg.rgb_out, rC, 1), "w", (rC, 1.0)), ("rgb_kv.bias", (hh + cfg.rgb_out,), "b", None)]
    s += [("depth_kv.weight", (hh + cfg.depth_out, dC, 1), "w", (dC, 1.0)), ("depth_kv.bias", (hh + cfg.depth_out,), "b", None)]
    s += _linear("state_q", hh, cfg.hidden, gain=2.0)
    s += [("text_k.weight", (hh, cfg.instr_out, 1), "w", (cfg.instr_out, 2.0)), ("text_k.bias", (hh,), "b", None)]
    s += _linear("text_q", hh, cfg.instr_out, gain=2.0)
    s += [("_scale", (), "const", 1.0 / math.sqrt(hh))]
    s += _linear("second_state_compress.0", cfg.hidden, cfg.hidden + cfg.rgb_out + cfg.depth_out + cfg.instr_out, gain=RELU_GAIN)
    s += _rnn_generic("second_state_encoder.rnn.", g, cfg.hidden, cfg.hidden)
    s += _linear("progress_monitor", 1, cfg.hidden)
    s += _linear("linear", cfg.num_actions, cfg.hidden)
    s += _linear("stop_linear", 1, cfg.hidden)
    return s


def materialize(spec, model_tag: str, seed: int = 0):
    """spec -> {key: np.ndarray}.  `model_tag` separates the hi and lo models' streams."""
    out = {}
    for key, shape, kind, aux in spec:
        n = int(np.prod(shape)) if len(shape) else 1
        if kind == "nbt":
            out[key] = np.zeros((), dtype=np.int64)
            continue
        if kind == "const":
            out[key] = np.full(shape, aux, dtype=np.float32)
            continue
        u = uniform01(model_tag + "/" + key, n, seed)
        if kind == "w":
            fan_in, gain = aux
            a = gain * math.sqrt(3.0 / fan_in)
            v = (u * 2.0 - 1.0) * np.float32(a)
        elif kind in ("b", "beta", "rmean"):
            v = (u * 2.0 - 1.0) * np.float32(0.1)
        elif kind in ("gamma", "rvar"):
            v = u + np.float32(0.5)
        elif kind == "emb":
            v = (u * 2.0 - 1.0) * np.float32(aux)
        else:
            raise KeyError(kind)
        out[key] = v.astype(np.float32).reshape(shape)
    return out


def make_weights(cfg: HCMConfig, seed: int = 0):
    """(high_level_state_dict, low_level_state_dict) as numpy fp32 dicts."""
    hi = materialize(high_level_spec(cfg), "hi", seed)
    lo = materialize(low_level_spec(cfg), "lo", seed)
    return hi, lo


def make_observations(cfg: HCMConfig, batch: int, step: int = 0, seed: int = 0, rgb_uint8: bool = False):
    """Synthetic observations with the `batch_obs` contract (common/utils.py:59-85):
    rgb (B,H,W,3) f32 holding integers 0..255, depth (B,H,W,1) f32 in [0,1),
    instruction (B,L) token ids ([CLS]=101 first, [SEP]=102 at len-1, 0-padded)."""
    B, L = batch, cfg.instr_len
    tag = f"obs/{step}"
    rh, rw = cfg.rgb_shape
    rgb = np.floor(uniform01(tag + "/rgb", B * rh * rw * 3, seed) * 256.0)
    rgb = rgb.reshape(B, rh, rw, 3).astype(np.uint8 if rgb_uint8 else np.float32)
    dh, dw = cfg.depth_shape if hasattr(cfg, "depth_shape") else (cfg.depth_hw, cfg.depth_hw)
    depth = uniform01(tag + "/depth", B * dh * dw, seed).reshape(B, dh, dw, 1)
    # the instruction is per-episode, not per-step: keyed without `step`
    ids = randint("obs/instr", B * L, 1000, cfg.bert_vocab, seed).reshape(B, L)
    lens = randint("obs/instr_len", B, min(L, max(2, L // 2)), L + 1, seed)
    for b in range(B):
        ids[b, 0] = 101
        ids[b, lens[b] - 1] = 102
        ids[b, lens[b]:] = 0
    return {"rgb": rgb, "depth": depth.astype(np.float32), "instruction": ids}


def make_cma_weights(cfg, seed: int = 0):
    """CMANet state_dict as a numpy fp32 dict (row 0 of the word embedding is the padding row: zeros)."""
    sd = materialize(cma_spec(cfg), "cma", seed)
    sd["instruction_encoder.embedding_layer.weight"][0] = 0.0
    return sd


def make_cma_observations(cfg, batch: int, step: int = 0, seed: int = 0, rgb_uint8: bool = False):
    """As make_observations, with instruction ids from the CMANet vocabulary: 1..vocab-1, 0-padded, per-row length
    in [L/2, L] (InstructionEncoder derives the lengths from `!= 0`, instruction_encoder.py:79)."""
    B, L = batch, cfg.instr_len
    tag = f"obs/{step}"
    rh, rw = cfg.rgb_shape
    rgb = np.floor(uniform01(tag + "/rgb", B * rh * rw * 3, seed) * 256.0)
    rgb = rgb.reshape(B, rh, rw, 3).astype(np.uint8 if rgb_uint8 else np.float32)
    dh, dw = cfg.depth_shape if hasattr(cfg, "depth_shape") else (cfg.depth_hw, cfg.depth_hw)
    depth = uniform01(tag + "/depth", B * dh * dw, seed).reshape(B, dh, dw, 1)
    ids = randint("obs/cma_instr", B * L, 1, cfg.vocab_size, seed).reshape(B, L)
    lens = randint("obs/cma_instr_len", B, max(2, L // 2), L + 1, seed)
    for b in range(B):
        ids[b, lens[b]:] = 0
    return {"rgb": rgb, "depth": depth.astype(np.float32), "instruction": ids}
