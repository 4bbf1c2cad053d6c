"""CPU ORACLE for the HCM per-step policy forward.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain torch-CPU fp32 functional ops, the algorithm of the
reference's hot path.  It is the *checker* for the HIP
path: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import it.  The product (`robo-vln_amd/`) never does.

Pinning: `oracle/gen_golden.py` imports the real reference modules from
/root/reference (third-party packages shimmed, see `oracle/ref_shims.py`), runs
them on the same synthetic weights/inputs, and writes `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks this restatement against those vectors.
In-tree reference semantics (seq2seq_highlevel_cma.py, seq2seq_lowlevel.py,
resnet_encoders.py, simple_cnns.py, transformer.py, state_encoder.py,
common/utils.py) and HuggingFace `BertModel` are pinned that way.  The
torchvision-0.2.2 ResNet-50 and the habitat-lab DDPPO GroupNorm ResNet /
SimpleCNN are NOT under /root/reference and not installed: their architecture is
restated from the public definitions both here and in the
shims, so for those two trunks parity is "unpinned" against the original
packages.

All citations are relative to /root/reference/robo_vln_baselines/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

RESNET50_BLOCKS = (3, 4, 6, 3)


class Weights:
    """name -> torch fp32 tensor view over a numpy state_dict."""

    def __init__(self, sd, prefix=""):
        self.sd = sd
        self.prefix = prefix
        self._cache = {}

    def __call__(self, key):
        k = self.prefix + key
        t = self._cache.get(k)
        if t is None:
            t = torch.from_numpy(np.ascontiguousarray(self.sd[k]))
            self._cache[k] = t
        return t

    def sub(self, prefix):
        w = Weights(self.sd, self.prefix + prefix)
        w._cache = self._cache
        return w


# ------------------------------------------------------------------ RGB trunk (torchvision resnet50)
def _bn(x, w, p):
    # eval-mode BatchNorm2d, eps 1e-5 (models frozen + .eval(): encoders/resnet_encoders.py:146-149)
    return F.batch_norm(x, w(p + ".running_mean"), w(p + ".running_var"), w(p + ".weight"), w(p + ".bias"),
                        training=False, eps=1e-5)


def tv_resnet50_trunk(x, w):
    """torchvision resnet50 conv1..layer4 (v1.5, stride on the 3x3) [SURVEY Appendix C]."""
    x = F.relu(_bn(F.conv2d(x, w("conv1.weight"), stride=2, padding=3), w, "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, nb in enumerate(RESNET50_BLOCKS):
        for bi in range(nb):
            p = f"layer{li + 1}.{bi}."
            stride = 2 if (li > 0 and bi == 0) else 1
            idt = x
            o = F.relu(_bn(F.conv2d(x, w(p + "conv1.weight")), w, p + "bn1"))
            o = F.relu(_bn(F.conv2d(o, w(p + "conv2.weight"), stride=stride, padding=1), w, p + "bn2"))
            o = _bn(F.conv2d(o, w(p + "conv3.weight")), w, p + "bn3")
            if bi == 0:
                idt = _bn(F.conv2d(x, w(p + "downsample.0.weight"), stride=stride), w, p + "downsample.1")
            x = F.relu(o + idt)
    return x


def _spatial_cat(x, emb):
    """encoders/resnet_encoders.py:91-104 / :218-231 -- Embedding(arange) `.view(1,-1,h,w)`:
    a raw reinterpretation of the (h*w, 64) table as (1, 64, h, w), NOT a transpose."""
    b, c, h, w_ = x.shape
    sp = emb.view(1, -1, h, w_).expand(b, emb.shape[1], h, w_)
    return torch.cat([x, sp], dim=1)


def rgb_resnet_spatial(rgb, w):
    """TorchVisionResNet50.forward, spatial_output=True (resnet_encoders.py:189-231):
    permute, /255 (no mean/std), trunk, avgpool patched to adaptive_avg_pool2d(4,4) (:160-166), cat pos-emb."""
    x = rgb.permute(0, 3, 1, 2) / 255.0
    x = tv_resnet50_trunk(x.contiguous(), w.sub("cnn."))
    x = F.adaptive_avg_pool2d(x, (4, 4))
    return _spatial_cat(x, w("spatial_embeddings.weight"))


def rgb_resnet_flat(rgb, w):
    """TorchVisionResNet50.forward, flat mode (:189-215,:234-237): hook on avgpool (global average, see
    SURVEY 8a-a3), fc 2048->out + ReLU.  The unused cnn.fc (2048->1000) output is discarded by the
    reference and is not computed here."""
    x = rgb.permute(0, 3, 1, 2) / 255.0
    x = tv_resnet50_trunk(x.contiguous(), w.sub("cnn."))
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return F.relu(F.linear(x, w("fc.weight"), w("fc.bias")))


# ------------------------------------------------------------------ depth trunk (habitat GN-ResNet50)
def _gn(x, w, p, groups):
    return F.group_norm(x, groups, w(p + ".weight"), w(p + ".bias"), eps=1e-5)


def habitat_resnet_encoder(depth, w, ngroups):
    """habitat ResNetEncoder.forward [SURVEY Appendix C]: NHWC->NCHW, avg_pool2d(2), GN-ResNet50
    (GroupNorm(ngroups) after every conv), 3x3 compression conv + GroupNorm(1) + ReLU.
    ngroups = baseplanes // 2 (resnet_encoders.py:27-33)."""
    x = depth.permute(0, 3, 1, 2)
    x = F.avg_pool2d(x, 2)
    b = w.sub("backbone.")
    x = F.relu(_gn(F.conv2d(x, b("conv1.0.weight"), stride=2, padding=3), b, "conv1.1", ngroups))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, nb in enumerate(RESNET50_BLOCKS):
        for bi in range(nb):
            p = f"layer{li + 1}.{bi}."
            stride = 2 if (li > 0 and bi == 0) else 1
            idt = x
            o = F.relu(_gn(F.conv2d(x, b(p + "convs.0.weight")), b, p + "convs.1", ngroups))
            o = F.relu(_gn(F.conv2d(o, b(p + "convs.3.weight"), stride=stride, padding=1), b, p + "convs.4", ngroups))
            o = _gn(F.conv2d(o, b(p + "convs.6.weight")), b, p + "convs.7", ngroups)
            if bi == 0:
                idt = _gn(F.conv2d(x, b(p + "downsample.0.weight"), stride=stride), b, p + "downsample.1", ngroups)
            x = F.relu(o + idt)
    x = F.conv2d(x, w("compression.0.weight"), padding=1)
    return F.relu(_gn(x, w, "compression.1", 1))


def depth_resnet_spatial(depth, w, ngroups):
    """VlnResnetDepthEncoder.forward spatial (resnet_encoders.py:76-104)."""
    x = habitat_resnet_encoder(depth, w.sub("visual_encoder."), ngroups)
    return _spatial_cat(x, w("spatial_embeddings.weight"))


def depth_resnet_flat(depth, w, ngroups):
    """VlnResnetDepthEncoder.forward flat (:56-62,:108): Flatten (NCHW order) -> Linear -> ReLU."""
    x = habitat_resnet_encoder(depth, w.sub("visual_encoder."), ngroups)
    return F.relu(F.linear(x.flatten(1), w("visual_fc.1.weight"), w("visual_fc.1.bias")))


def simple_cnn(x_nchw, w):
    """SimpleAllCNN.cnn (encoders/simple_cnns.py:76-100): conv8/4 ReLU conv4/2 ReLU conv3/1 Flatten Linear ReLU."""
    x = F.relu(F.conv2d(x_nchw, w("cnn.0.weight"), w("cnn.0.bias"), stride=4))
    x = F.relu(F.conv2d(x, w("cnn.2.weight"), w("cnn.2.bias"), stride=2))
    x = F.conv2d(x, w("cnn.4.weight"), w("cnn.4.bias"), stride=1)
    return F.relu(F.linear(x.contiguous().flatten(1), w("cnn.7.weight"), w("cnn.7.bias")))


def simple_depth_cnn(depth, w):
    """SimpleDepthCNN.forward (simple_cnns.py:121-125)."""
    return simple_cnn(depth.permute(0, 3, 1, 2), w)


def simple_rgb_cnn(rgb, w):
    """SimpleRGBCNN.forward (simple_cnns.py:142-147): permute, /255."""
    return simple_cnn(rgb.permute(0, 3, 1, 2) / 255.0, w)


# ------------------------------------------------------------------ BERT (transformers.BertModel, eval, no mask)
def bert_encoder(ids, w, n_layers, n_heads):
    """BertModel(input_ids)[0] as called at models/seq2seq_highlevel_cma.py:192-195: no attention
    mask (padding attended), token_type 0, LN eps 1e-12, erf-GELU, post-LN; pooler output unused."""
    B, L = ids.shape
    e = w.sub("embeddings.")
    x = e("word_embeddings.weight")[ids] + e("position_embeddings.weight")[:L][None] \
        + e("token_type_embeddings.weight")[0][None, None]
    x = F.layer_norm(x, (x.shape[-1],), e("LayerNorm.weight"), e("LayerNorm.bias"), 1e-12)
    H = x.shape[-1]
    dh = H // n_heads
    for i in range(n_layers):
        p = w.sub(f"encoder.layer.{i}.")
        q = F.linear(x, p("attention.self.query.weight"), p("attention.self.query.bias"))
        k = F.linear(x, p("attention.self.key.weight"), p("attention.self.key.bias"))
        v = F.linear(x, p("attention.self.value.weight"), p("attention.self.value.bias"))
        q = q.view(B, L, n_heads, dh).transpose(1, 2)
        k = k.view(B, L, n_heads, dh).transpose(1, 2)
        v = v.view(B, L, n_heads, dh).transpose(1, 2)
        att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), dim=-1)
        ctx = (att @ v).transpose(1, 2).reshape(B, L, H)
        a = F.linear(ctx, p("attention.output.dense.weight"), p("attention.output.dense.bias"))
        x = F.layer_norm(a + x, (H,), p("attention.output.LayerNorm.weight"), p("attention.output.LayerNorm.bias"), 1e-12)
        h = F.gelu(F.linear(x, p("intermediate.dense.weight"), p("intermediate.dense.bias")))
        h = F.linear(h, p("output.dense.weight"), p("output.dense.bias"))
        x = F.layer_norm(h + x, (H,), p("output.LayerNorm.weight"), p("output.LayerNorm.bias"), 1e-12)
    return x


# ------------------------------------------------------------------ Visual_Ling_Attn
def sinusoid_table(L, d):
    """common/utils.py:167-185: pe[p,2i]=sin(p/10000^(2i/d)), pe[p,2i+1]=cos(same)."""
    pos = torch.arange(L, dtype=torch.float32).view(-1, 1)
    dim = torch.arange(d // 2, dtype=torch.float32).view(1, -1)
    ang = pos / 10000 ** (2 * dim / d)
    out = torch.zeros(L, d)
    out[:, ::2] = torch.sin(ang)
    out[:, 1::2] = torch.cos(ang)
    return out


def _mha(q_in, kv_in, w, h):
    """MultiHeadAttention.forward (models/transformer/transformer.py:111-126) over
    ScaledDotProductAttention.forward (:81-109), masks None, dropout identity (eval)."""
    B, nq, d = q_in.shape
    nk = kv_in.shape[1]
    dk = d // h
    a = w.sub("attention.")
    q = F.linear(q_in, a("fc_q.weight"), a("fc_q.bias")).view(B, nq, h, dk).permute(0, 2, 1, 3)
    k = F.linear(kv_in, a("fc_k.weight"), a("fc_k.bias")).view(B, nk, h, dk).permute(0, 2, 3, 1)
    v = F.linear(kv_in, a("fc_v.weight"), a("fc_v.bias")).view(B, nk, h, dk).permute(0, 2, 1, 3)
    att = torch.softmax(torch.matmul(q, k) / np.sqrt(dk), -1)
    out = torch.matmul(att, v).permute(0, 2, 1, 3).contiguous().view(B, nq, d)
    out = F.linear(out, a("fc_o.weight"), a("fc_o.bias"))
    return F.layer_norm(q_in + out, (d,), w("layer_norm.weight"), w("layer_norm.bias"))


def _pwff(x, w):
    """PositionWiseFeedForward.forward (transformer.py:25-43): LN(x + fc2(relu(fc1 x)))."""
    d = x.shape[-1]
    y = F.linear(F.relu(F.linear(x, w("fc1.weight"), w("fc1.bias"))), w("fc2.weight"), w("fc2.bias"))
    return F.layer_norm(x + y, (d,), w("layer_norm.weight"), w("layer_norm.bias"))


def visual_ling_attn(ins, vis, w, n_layers, h):
    """Visual_Ling_Attn.forward (transformer.py:251-281).  One shared LayerNorm for both streams
    (:260,:265,:269); PE added after the LN; every layer's query (and residual) stream is the
    instruction `I`, the kv stream is the previous layer's output (layer 0: vision tokens) (:279-280)."""
    d = w("layer_norm.weight").shape[0]
    out = F.layer_norm(F.relu(F.linear(vis, w("vis_fc.weight"), w("vis_fc.bias"))), (d,),
                       w("layer_norm.weight"), w("layer_norm.bias"))
    I = F.layer_norm(F.relu(F.linear(ins, w("ins_fc.weight"), w("ins_fc.bias"))), (d,),
                     w("layer_norm.weight"), w("layer_norm.bias"))
    I = I + sinusoid_table(I.shape[1], I.shape[2])[None]
    for l in range(n_layers):
        lw = w.sub(f"layers.{l}.")
        out = _pwff(_mha(I, out, lw.sub("enc_att."), h), lw.sub("pwff."))   # InterModuleAttnLayer :218-221
    return out


# ------------------------------------------------------------------ RNN state encoder, single step
def rnn_single_forward(x, hidden, mask, w, rnn_type):
    """RNNStateEncoder.single_forward (models/decoder/state_encoder.py:72-81 with :47-70):
    hidden (R,B,H): LSTM R=2 = cat[h,c]; h and c are multiplied by mask (B,) before the step;
    gate order i,f,g,o (LSTM) / r,z,n (GRU); biases b_ih + b_hh."""
    Wih, Whh, bih, bhh = w("rnn.weight_ih_l0"), w("rnn.weight_hh_l0"), w("rnn.bias_ih_l0"), w("rnn.bias_hh_l0")
    m = mask.view(-1, 1)
    if rnn_type == "LSTM":
        h, c = hidden[0] * m, hidden[1] * m
        g = F.linear(x, Wih, bih) + F.linear(h, Whh, bhh)
        i, f, gg, o = g.chunk(4, dim=1)
        c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h2 = torch.sigmoid(o) * torch.tanh(c2)
        return h2, torch.stack([h2, c2], 0)
    h = hidden[0] * m
    gi = F.linear(x, Wih, bih)
    gh = F.linear(h, Whh, bhh)
    ir, iz, inn = gi.chunk(3, dim=1)
    hr, hz, hn = gh.chunk(3, dim=1)
    r = torch.sigmoid(ir + hr)
    z = torch.sigmoid(iz + hz)
    n = torch.tanh(inn + r * hn)
    h2 = (1 - z) * n + z * h
    return h2, h2[None]


def rnn_seq_forward(x, hidden, masks, w, rnn_type):
    """RNNStateEncoder.seq_forward (models/decoder/state_encoder.py:83-133): x is (T*N, F) time-major, hidden (R,N,H),
    masks (T*N,).  Steps are grouped into segments that start at t=0 and at every t where ANY environment has mask 0;
    the hidden state is multiplied by masks[start] at each segment start and the RNN runs over the segment."""
    n = hidden.shape[1]
    t = x.shape[0] // n
    x = x.view(t, n, -1)
    masks = masks.view(t, n)
    has_zeros = (masks[1:] == 0.0).any(dim=-1).nonzero().squeeze(-1)
    starts = [0] + (has_zeros + 1).tolist() + [t]
    outs = []
    for i in range(len(starts) - 1):
        a, b = starts[i], starts[i + 1]
        if a == b:
            continue
        m = masks[a]
        for step in range(a, b):
            # inside a segment every later mask is all-ones by construction, so only the first step is masked
            h, hidden = rnn_single_forward(x[step], hidden, m if step == a else torch.ones(n), w, rnn_type)
            outs.append(h)
    return torch.cat(outs, 0), hidden


def rnn_forward(x, hidden, masks, w, rnn_type):
    """RNNStateEncoder.forward (state_encoder.py:135-137)."""
    if x.shape[0] == hidden.shape[1]:
        return rnn_single_forward(x, hidden, masks, w, rnn_type)
    return rnn_seq_forward(x, hidden, masks, w, rnn_type)


# ------------------------------------------------------------------ the two models
class HighLevelOracle:
    """Seq2Seq_HighLevel_CMA.forward (models/seq2seq_highlevel_cma.py:170-233)."""

    def __init__(self, cfg, sd):
        self.cfg = cfg
        self.w = Weights(sd)

    @torch.no_grad()
    def forward(self, obs, hidden, mask, taps=None):
        cfg, w = self.cfg, self.w
        rgb = torch.as_tensor(obs["rgb"]).float()
        depth = torch.as_tensor(obs["depth"]).float()
        ids = torch.as_tensor(obs["instruction"]).long()
        hidden = torch.as_tensor(hidden).float()
        B = rgb.shape[0]
        mask = torch.as_tensor(mask).float().reshape(B, -1)[:, 0]   # masks[:,0] (:208)
        dep = depth_resnet_spatial(depth, w.sub("depth_encoder."), cfg.depth_baseplanes // 2).flatten(2)  # :178-179
        rg = rgb_resnet_spatial(rgb, w.sub("rgb_encoder.")).flatten(2)                                     # :180-181
        if cfg.ablate_depth:
            dep = dep * 0                                                                                  # :185-186
        if cfg.ablate_rgb:
            rg = rg * 0                                                                                    # :187-188
        ids = ids.expand(B, ids.shape[1])                                                                  # :189-190
        emb = bert_encoder(ids, w.sub("embedding_layer."), cfg.bert_layers, cfg.bert_heads)                # :192-195
        rgb_sp = F.conv1d(rg, w("rgb_kv.weight"), w("rgb_kv.bias"))                                        # :198
        dep_sp = F.conv1d(dep, w("depth_kv.weight"), w("depth_kv.bias"))                                   # :199
        vw = w.sub("image_cm_encoder.")
        a_rgb = visual_ling_attn(emb, rgb_sp.permute(0, 2, 1), vw, cfg.vla_layers, cfg.vla_heads)          # :200
        a_dep = visual_ling_attn(emb, dep_sp.permute(0, 2, 1), vw, cfg.vla_layers, cfg.vla_heads)          # :201
        p_rgb = a_rgb.mean(1)   # cross_pooler: AdaptiveAvgPool1d(1) over all L tokens incl. padding (:209-210)
        p_dep = a_dep.mean(1)
        rgb_in = F.relu(F.linear(rg.mean(2), w("rgb_linear.2.weight"), w("rgb_linear.2.bias")))            # :213
        dep_in = F.relu(F.linear(dep.flatten(1), w("depth_linear.1.weight"), w("depth_linear.1.bias")))    # :214
        x = torch.cat((rgb_in, dep_in, p_rgb, p_dep), dim=1)                                               # :215
        h, hid = rnn_forward(x, hidden, mask, w.sub("state_encoder."), cfg.rnn_type)                       # :219
        logits = F.linear(h, w("linear.weight"), w("linear.bias"))                                         # :232
        if taps is not None:
            taps.update(depth_spatial=dep, rgb_spatial=rg, bert=emb, rgb_kv=rgb_sp, depth_kv=dep_sp,
                        vla_rgb=a_rgb, vla_depth=a_dep, rnn_in=x, rnn_out=h)
        return logits, hid


class LowLevelOracle:
    """Seq2Seq_LowLevel.forward (models/seq2seq_lowlevel.py:116-162)."""

    def __init__(self, cfg, sd):
        self.cfg = cfg
        self.w = Weights(sd)

    @torch.no_grad()
    def forward(self, obs, hidden, mask, subtask, taps=None):
        cfg, w = self.cfg, self.w
        rgb = torch.as_tensor(obs["rgb"]).float()
        depth = torch.as_tensor(obs["depth"]).float()
        hidden = torch.as_tensor(hidden).float()
        mask = torch.as_tensor(mask).float().reshape(rgb.shape[0], -1)[:, 0]       # :145
        subtask = torch.as_tensor(subtask).long()
        if cfg.depth_encoder == "VlnResnetDepthEncoder":
            d = depth_resnet_flat(depth, w.sub("depth_encoder."), cfg.depth_baseplanes // 2)               # :128
        else:
            d = simple_depth_cnn(depth, w.sub("depth_encoder."))
        if cfg.rgb_encoder == "TorchVisionResNet50":
            r = rgb_resnet_flat(rgb, w.sub("rgb_encoder."))                                                # :129
        else:
            r = simple_rgb_cnn(rgb, w.sub("rgb_encoder."))
        if cfg.ablate_depth:
            d = d * 0                                                                                      # :132-133
        if cfg.ablate_rgb:
            r = r * 0                                                                                      # :134-135
        st = w("sub_task_embedding.weight")[subtask]                                                       # :141
        x = torch.cat([d, r, st], dim=1)                                                                   # :143
        h, hid = rnn_forward(x, hidden, mask, w.sub("state_encoder."), cfg.rnn_type)                       # :147
        out = F.linear(h, w("linear.weight"), w("linear.bias"))                                            # :160
        stop = F.linear(h, w("stop_linear.weight"), w("stop_linear.bias"))                                 # :161
        if taps is not None:
            taps.update(depth_flat=d, rgb_flat=r, rnn_in=x, rnn_out=h)
        return out, stop, hid


class PolicyOracle:
    """The caller-side step of the eval loop (hierarchical_trainer.py:1095-1101):
    hi -> argmax -> lo; returns the (B,7) action record [4 logits, v, w, stop] and new hidden states."""

    def __init__(self, cfg, hi_sd, lo_sd):
        self.hi = HighLevelOracle(cfg, hi_sd)
        self.lo = LowLevelOracle(cfg, lo_sd)

    def hi_forward_ragged(self, obs, hi_h, mask, lengths):
        """A padded batch whose environment b owns the first lengths[b] tokens of its row == the reference called once per
        environment with that environment's unpadded (1, lengths[b]) instruction (its eval loop runs one environment,
        hierarchical_trainer.py:1088-1197)."""
        B = len(lengths)
        mask = torch.as_tensor(mask).float().reshape(B, -1)[:, 0]
        outs, hs = [], []
        for b in range(B):
            ob = {"rgb": obs["rgb"][b:b + 1], "depth": obs["depth"][b:b + 1], "instruction": obs["instruction"][b:b + 1, :int(lengths[b])]}
            lg, hb = self.hi.forward(ob, torch.as_tensor(hi_h)[:, b:b + 1], mask[b:b + 1])
            outs.append(lg)
            hs.append(hb)
        return torch.cat(outs, 0), torch.cat(hs, 1)

    def act(self, obs, hi_h, lo_h, mask, lengths=None):
        if lengths is not None:
            logits, hi_h2 = self.hi_forward_ragged(obs, hi_h, mask, lengths)
        else:
            logits, hi_h2 = self.hi.forward(obs, hi_h, mask)
        pred = torch.argmax(logits, dim=1)                                                                 # :1098
        vel, stop, lo_h2 = self.lo.forward(obs, lo_h, mask, pred)
        return torch.cat([logits, vel, stop], dim=1), hi_h2, lo_h2


# ------------------------------------------------------------------ CMANet flat baseline
def instruction_encoder(ids, w, hidden, bidirectional, rnn_type="LSTM"):
    """InstructionEncoder.forward with final_state_only=False (models/encoders/instruction_encoder.py:70-92; CMANet sets
    the flag at cma.py:33-35): lengths = count of non-zero ids; embedding; a (bi)LSTM over the PACKED sequence -- sample
    b runs exactly len_b steps (the reverse direction starts at its token len_b-1), outputs past len_b are zero; the
    padded output is cut to the longest sequence of the batch (pad_packed_sequence) and returned as (B, C, Lmax)."""
    ids = ids.long()
    B, L = ids.shape
    lengths = (ids != 0).long().sum(dim=1)
    lmax = int(lengths.max().item())
    emb = w("embedding_layer.weight")[ids]                       # (B, L, E)

    def direction(suffix, reverse):
        Wih, Whh = w("encoder_rnn.weight_ih_l0" + suffix), w("encoder_rnn.weight_hh_l0" + suffix)
        bih, bhh = w("encoder_rnn.bias_ih_l0" + suffix), w("encoder_rnn.bias_hh_l0" + suffix)
        h = torch.zeros(B, hidden)
        c = torch.zeros(B, hidden)
        out = torch.zeros(B, lmax, hidden)
        steps = range(lmax - 1, -1, -1) if reverse else range(lmax)
        for t in steps:
            act = (t < lengths).float().view(B, 1)
            if rnn_type == "GRU":
                # nn.GRU (instruction_encoder.py:42): r, z, n gates; b_hn sits inside the reset gate's product
                gi_, gh_ = F.linear(emb[:, t], Wih, bih), F.linear(h, Whh, bhh)
                ir, iz, in_ = gi_.chunk(3, dim=1)
                hr, hz, hn = gh_.chunk(3, dim=1)
                r, z = torch.sigmoid(ir + hr), torch.sigmoid(iz + hz)
                n = torch.tanh(in_ + r * hn)
                h2 = (1 - z) * n + z * h
                h = act * h2 + (1 - act) * h
                out[:, t] = act * h2
                continue
            g = F.linear(emb[:, t], Wih, bih) + F.linear(h, Whh, bhh)
            i, f, gg, o = g.chunk(4, dim=1)
            c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h2 = torch.sigmoid(o) * torch.tanh(c2)
            h = act * h2 + (1 - act) * h
            c = act * c2 + (1 - act) * c
            out[:, t] = act * h2
        return out

    outs = [direction("", False)]
    if bidirectional:
        outs.append(direction("_reverse", True))
    return torch.cat(outs, dim=2).permute(0, 2, 1), lengths     # (B, C, Lmax)


def cma_attn(q, k, v, scale, mask=None):
    """CMANet._attn (cma.py:201-209): one query per sample; logits - mask*1e8, THEN * scale, softmax over positions."""
    logits = torch.einsum("nc,nci->ni", q, k)
    if mask is not None:
        logits = logits - mask.float() * 1e8
    attn = F.softmax(logits * scale, dim=1)
    return torch.einsum("ni,nci->nc", attn, v)


class CMAOracle:
    """CMANet.forward (models/cma.py:211-333), non-RCM state encoder, no prev-action embedding."""

    def __init__(self, cfg, sd):
        self.cfg = cfg
        self.w = Weights(sd)

    @torch.no_grad()
    def forward(self, obs, hidden, mask, taps=None):
        cfg, w = self.cfg, self.w
        rgb = torch.as_tensor(obs["rgb"]).float()
        depth = torch.as_tensor(obs["depth"]).float()
        ids = torch.as_tensor(obs["instruction"]).long()
        hidden = torch.as_tensor(hidden).float()
        B = rgb.shape[0]
        R = cfg.num_recurrent_layers // 2
        mask = torch.as_tensor(mask).float().reshape(B, -1)[:, 0]                                     # :219
        dep = depth_resnet_spatial(depth, w.sub("depth_encoder."), cfg.depth_baseplanes // 2).flatten(2)   # :220-221
        rg = rgb_resnet_spatial(rgb, w.sub("rgb_encoder.")).flatten(2)                                # :223-224
        ids = ids.expand(B, ids.shape[1])                                                             # :226
        ins, lengths = instruction_encoder(ids, w.sub("instruction_encoder."), cfg.instr_hidden, cfg.bidirectional, cfg.instr_rnn)  # :227
        ins_enc = ins                                      # what a forward hook on the encoder module sees (the golden's `instruction` tap)
        if cfg.ablate_instruction:
            ins = ins * 0                                                                             # :236-237
        if cfg.ablate_depth:
            dep = dep * 0                                                                             # :238-239
        if cfg.ablate_rgb:
            rg = rg * 0                                                                               # :240-241
        rgb_in = F.relu(F.linear(rg.mean(2), w("rgb_linear.2.weight"), w("rgb_linear.2.bias")))       # :256
        dep_in = F.relu(F.linear(dep.flatten(1), w("depth_linear.1.weight"), w("depth_linear.1.bias")))   # :257
        state_in = torch.cat([rgb_in, dep_in], dim=1)                                                 # :262
        state, hid1 = rnn_forward(state_in, hidden[:R], mask, w.sub("state_encoder."), cfg.rnn_type)  # :263-270
        scale = w("_scale")
        q_state = F.linear(state, w("state_q.weight"), w("state_q.bias"))                             # :272
        k_text = F.conv1d(ins, w("text_k.weight"), w("text_k.bias"))                                  # :273
        text_mask = (ins == 0.0).all(dim=1)                                                           # :274
        text = cma_attn(q_state, k_text, ins, scale, text_mask)                                       # :275-277
        hh = cfg.hidden // 2
        rgb_k, rgb_v = torch.split(F.conv1d(rg, w("rgb_kv.weight"), w("rgb_kv.bias")), hh, dim=1)    # :281-283
        dep_k, dep_v = torch.split(F.conv1d(dep, w("depth_kv.weight"), w("depth_kv.bias")), hh, dim=1)   # :284-286
        q_text = F.linear(text, w("text_q.weight"), w("text_q.bias"))                                 # :288
        rgb_att = cma_attn(q_text, rgb_k, rgb_v, scale)                                               # :289
        dep_att = cma_attn(q_text, dep_k, dep_v, scale)                                               # :290
        x = torch.cat([state, text, rgb_att, dep_att], dim=1)                                         # :309-311
        x = F.relu(F.linear(x, w("second_state_compress.0.weight"), w("second_state_compress.0.bias")))   # :312
        x2, hid2 = rnn_forward(x, hidden[R:], mask, w.sub("second_state_encoder."), cfg.rnn_type)     # :313-318
        out = F.linear(x2, w("linear.weight"), w("linear.bias"))                                      # :331
        stop = F.linear(x2, w("stop_linear.weight"), w("stop_linear.bias"))                           # :332
        if taps is not None:
            taps.update(depth_spatial=dep, rgb_spatial=rg, instruction=ins_enc, state=state, text=text, rgb_att=rgb_att,
                        depth_att=dep_att, compress=x, rnn2_out=x2)
        return out, stop, torch.cat([hid1, hid2], dim=0)
