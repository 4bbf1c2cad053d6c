// State-dict spec (strict keys/shapes), and weight preparation at hcm_finalize():
// BatchNorm folding, OIHW -> OHWI re-layout for the NHWC implicit GEMM, flatten-order permutations,
// fused QKV / KV / recurrent weight concatenation, conversion to the compute dtype, upload.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "model.h"

namespace hcm {

static const int RESNET50_BLOCKS[4] = {3, 4, 6, 3};

// ------------------------------------------------------------------------------------------------ spec
struct SpecB {
    std::map<std::string, HostTensor>& m;
    void add(const std::string& k, Shape s) { m[k].shape = std::move(s); }
    void conv(const std::string& k, int co, int ci, int kh, int kw) { add(k, {co, ci, kh, kw}); }
    void linear(const std::string& p, int o, int i) { add(p + ".weight", {o, i}); add(p + ".bias", {o}); }
    void norm(const std::string& p, int c) { add(p + ".weight", {c}); add(p + ".bias", {c}); }
    void bn(const std::string& p, int c) {
        norm(p, c); add(p + ".running_mean", {c}); add(p + ".running_var", {c}); add(p + ".num_batches_tracked", {});
    }
};

static void spec_tv_resnet50(SpecB& s, const std::string& pre, bool with_fc) {
    s.conv(pre + "conv1.weight", 64, 3, 7, 7); s.bn(pre + "bn1", 64);
    int inpl = 64;
    for (int li = 0; li < 4; ++li) {
        const int planes = 64 << li;
        for (int bi = 0; bi < RESNET50_BLOCKS[li]; ++bi) {
            const std::string p = pre + "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
            s.conv(p + "conv1.weight", planes, inpl, 1, 1); s.bn(p + "bn1", planes);
            s.conv(p + "conv2.weight", planes, planes, 3, 3); s.bn(p + "bn2", planes);
            s.conv(p + "conv3.weight", planes * 4, planes, 1, 1); s.bn(p + "bn3", planes * 4);
            if (bi == 0) { s.conv(p + "downsample.0.weight", planes * 4, inpl, 1, 1); s.bn(p + "downsample.1", planes * 4); }
            inpl = planes * 4;
        }
    }
    if (with_fc) s.linear(pre + "fc", 1000, 2048);
}

static void spec_gn_resnet50(SpecB& s, const std::string& pre, int in_ch, int base, int cc) {
    const std::string bb = pre + "backbone.";
    s.conv(bb + "conv1.0.weight", base, in_ch, 7, 7); s.norm(bb + "conv1.1", base);
    int inpl = base;
    for (int li = 0; li < 4; ++li) {
        const int planes = base << li;
        for (int bi = 0; bi < RESNET50_BLOCKS[li]; ++bi) {
            const std::string p = bb + "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
            s.conv(p + "convs.0.weight", planes, inpl, 1, 1); s.norm(p + "convs.1", planes);
            s.conv(p + "convs.3.weight", planes, planes, 3, 3); s.norm(p + "convs.4", planes);
            s.conv(p + "convs.6.weight", planes * 4, planes, 1, 1); s.norm(p + "convs.7", planes * 4);
            if (bi == 0) { s.conv(p + "downsample.0.weight", planes * 4, inpl, 1, 1); s.norm(p + "downsample.1", planes * 4); }
            inpl = planes * 4;
        }
    }
    s.conv(pre + "compression.0.weight", cc, inpl, 3, 3); s.norm(pre + "compression.1", cc);
}

static int simple_cnn_out_hw(int hw) {
    int d = hw;
    const int ks[3] = {8, 4, 3}, st[3] = {4, 2, 1};
    for (int i = 0; i < 3; ++i) d = (d - ks[i]) / st[i] + 1;
    return d;
}
static void spec_simple_cnn(SpecB& s, const std::string& pre, int in_ch, int h, int w, int out_f) {
    const int dh = simple_cnn_out_hw(h), dw = simple_cnn_out_hw(w);          // simple_cnns.py:63-73: the two dimensions on their own
    s.add(pre + "cnn.0.weight", {32, in_ch, 8, 8}); s.add(pre + "cnn.0.bias", {32});
    s.add(pre + "cnn.2.weight", {64, 32, 4, 4}); s.add(pre + "cnn.2.bias", {64});
    s.add(pre + "cnn.4.weight", {32, 64, 3, 3}); s.add(pre + "cnn.4.bias", {32});
    s.linear(pre + "cnn.7", out_f, 32 * dh * dw);
}

static void spec_rnn(SpecB& s, const std::string& pre, const hcm_config& c, int in_f) {
    const int g = c.rnn_type == HCM_LSTM ? 4 : 3;
    s.add(pre + "weight_ih_l0", {g * c.hidden, in_f}); s.add(pre + "weight_hh_l0", {g * c.hidden, c.hidden});
    s.add(pre + "bias_ih_l0", {g * c.hidden}); s.add(pre + "bias_hh_l0", {g * c.hidden});
}

int depth_final_spatial(const hcm_config& c) { return (c.depth_h / 2) / 32; }
int depth_compress_channels(const hcm_config& c) {
    const int fs = depth_final_spatial(c);
    return (int)std::lround(2048.0 / (fs * fs));
}
// Frames of 64*k pixels with k not a power of two give channel counts like 228 (192 px), 82 (320), 57 (384): the compression conv, its
// GroupNorm and the token rows run on the next power of two (zero weight rows, gamma = beta = 0: the extra channels are exact zeros), the
// statistics count the real channels only, and the consumers' weight columns are laid out to match (zero columns under the padding).
int depth_compress_padded(const hcm_config& c) {
    const int cc = depth_compress_channels(c);
    int p = 8;
    while (p < cc) p *= 2;
    return p;
}
// column n of a depth token row [compression channels | padding | 64 position-embedding channels] -> the reference's channel, -1 = padding
static int depth_tok_src(int n, int cc, int ccp) { return n < cc ? n : n < ccp ? -1 : n - (ccp - cc); }

void build_spec_high(hcm_ctx* ctx) {
    const hcm_config& c = ctx->cfg;
    SpecB s{ctx->sd[HCM_HIGH]};
    const int h = c.bert_hidden;
    const std::string e = "embedding_layer.embeddings.";
    s.add(e + "word_embeddings.weight", {c.bert_vocab, h});
    s.add(e + "position_embeddings.weight", {c.bert_max_pos, h});
    s.add(e + "token_type_embeddings.weight", {2, h});
    s.norm(e + "LayerNorm", h);
    for (int i = 0; i < c.bert_layers; ++i) {
        const std::string p = "embedding_layer.encoder.layer." + std::to_string(i) + ".";
        s.linear(p + "attention.self.query", h, h); s.linear(p + "attention.self.key", h, h);
        s.linear(p + "attention.self.value", h, h); s.linear(p + "attention.output.dense", h, h);
        s.norm(p + "attention.output.LayerNorm", h);
        s.linear(p + "intermediate.dense", c.bert_inter, h); s.linear(p + "output.dense", h, c.bert_inter);
        s.norm(p + "output.LayerNorm", h);
    }
    s.linear("embedding_layer.pooler.dense", h, h);
    s.linear("ins_fc", 256, 768);
    const int fs = depth_final_spatial(c), cc = depth_compress_channels(c);
    const int dC = cc + 64, rC = 2048 + 64;
    spec_gn_resnet50(s, "depth_encoder.visual_encoder.", 1, c.depth_baseplanes, cc);
    s.add("depth_encoder.spatial_embeddings.weight", {fs * fs, 64});
    spec_tv_resnet50(s, "rgb_encoder.cnn.", false);
    s.add("rgb_encoder.spatial_embeddings.weight", {16, 64});
    s.linear("rgb_linear.2", c.rgb_out, rC);
    s.linear("depth_linear.1", c.depth_out, dC * fs * fs);
    s.add("rgb_kv.weight", {c.vis_in, rC, 1}); s.add("rgb_kv.bias", {c.vis_in});
    s.add("depth_kv.weight", {c.vis_in, dC, 1}); s.add("depth_kv.bias", {c.vis_in});
    const int d = c.d_model;
    for (int i = 0; i < c.vla_layers; ++i) {
        const std::string p = "image_cm_encoder.layers." + std::to_string(i) + ".";
        const std::string a = p + "enc_att.attention.";
        s.linear(a + "fc_q", d, d); s.linear(a + "fc_k", d, d); s.linear(a + "fc_v", d, d); s.linear(a + "fc_o", d, d);
        s.norm(p + "enc_att.layer_norm", d);
        s.linear(p + "pwff.fc1", c.d_ff, d); s.linear(p + "pwff.fc2", d, c.d_ff);
        s.norm(p + "pwff.layer_norm", d);
    }
    s.linear("image_cm_encoder.vis_fc", d, c.vis_in);
    s.linear("image_cm_encoder.ins_fc", d, c.ins_in);
    s.norm("image_cm_encoder.layer_norm", d);
    spec_rnn(s, "state_encoder.rnn.", c, 2 * 256 + c.depth_out + c.rgb_out);   // IMAGE_CROSS_MODAL_ENCODER.d_model = 256
    s.linear("progress_monitor", 1, c.hidden);
    s.linear("linear", c.num_actions, c.hidden);
}

void build_spec_low(hcm_ctx* ctx) {
    const hcm_config& c = ctx->cfg;
    SpecB s{ctx->sd[HCM_LOW]};
    if (c.depth_encoder == HCM_ENC_RESNET) {
        const int fs = depth_final_spatial(c), cc = depth_compress_channels(c);
        spec_gn_resnet50(s, "depth_encoder.visual_encoder.", 1, c.depth_baseplanes, cc);
        s.linear("depth_encoder.visual_fc.1", c.depth_out, cc * fs * fs);
    } else {
        spec_simple_cnn(s, "depth_encoder.", 1, c.depth_h, c.depth_w, c.depth_out);
    }
    if (c.rgb_encoder == HCM_ENC_RESNET) {
        spec_tv_resnet50(s, "rgb_encoder.cnn.", true);
        s.linear("rgb_encoder.fc", c.rgb_out, 2048);
    } else {
        spec_simple_cnn(s, "rgb_encoder.", 3, c.rgb_h, c.rgb_w, c.rgb_out);
    }
    s.add("sub_task_embedding.weight", {c.num_sub_tasks + 1, 32});
    spec_rnn(s, "state_encoder.rnn.", c, c.depth_out + c.rgb_out + 32);
    s.linear("progress_monitor", 1, c.hidden);
    s.linear("linear", c.lo_actions, c.hidden);
    s.linear("stop_linear", 1, c.hidden);
}

// CMANet state_dict (models/cma.py:28-186; InstructionEncoder models/encoders/instruction_encoder.py:9-47)
void build_spec_cma(hcm_ctx* ctx) {
    const hcm_config& c = ctx->cfg;
    const hcm_cma_config& m = ctx->cma_cfg;
    SpecB s{ctx->sd[HCM_CMA]};
    s.add("instruction_encoder.embedding_layer.weight", {m.vocab_size, m.embedding_size});
    for (int d = 0; d < (m.bidirectional ? 2 : 1); ++d) {
        const std::string sfx = d ? "_reverse" : "";
        const std::string p = "instruction_encoder.encoder_rnn.";
        const int G = m.instr_rnn == HCM_GRU ? 3 : 4;        // nn.GRU: gates r, z, n; nn.LSTM: i, f, g, o (instruction_encoder.py:42-47)
        s.add(p + "weight_ih_l0" + sfx, {G * m.instr_hidden, m.embedding_size});
        s.add(p + "weight_hh_l0" + sfx, {G * m.instr_hidden, m.instr_hidden});
        s.add(p + "bias_ih_l0" + sfx, {G * m.instr_hidden});
        s.add(p + "bias_hh_l0" + sfx, {G * m.instr_hidden});
    }
    const int fs = depth_final_spatial(c), cc = depth_compress_channels(c);
    const int dC = cc + 64, rC = 2048 + 64, hh = c.hidden / 2;
    const int instr_out = m.instr_hidden * (m.bidirectional ? 2 : 1);
    spec_gn_resnet50(s, "depth_encoder.visual_encoder.", 1, c.depth_baseplanes, cc);
    s.add("depth_encoder.spatial_embeddings.weight", {fs * fs, 64});
    spec_tv_resnet50(s, "rgb_encoder.cnn.", false);
    s.add("rgb_encoder.spatial_embeddings.weight", {16, 64});
    s.linear("rgb_linear.2", c.rgb_out, rC);
    s.linear("depth_linear.1", c.depth_out, dC * fs * fs);
    spec_rnn(s, "state_encoder.rnn.", c, c.rgb_out + c.depth_out);
    s.add("rgb_kv.weight", {hh + c.rgb_out, rC, 1}); s.add("rgb_kv.bias", {hh + c.rgb_out});
    s.add("depth_kv.weight", {hh + c.depth_out, dC, 1}); s.add("depth_kv.bias", {hh + c.depth_out});
    s.linear("state_q", hh, c.hidden);
    s.add("text_k.weight", {hh, instr_out, 1}); s.add("text_k.bias", {hh});
    s.linear("text_q", hh, instr_out);
    s.add("_scale", {});
    s.linear("second_state_compress.0", c.hidden, c.hidden + c.rgb_out + c.depth_out + instr_out);
    spec_rnn(s, "second_state_encoder.rnn.", c, c.hidden);
    s.linear("progress_monitor", 1, c.hidden);
    s.linear("linear", c.num_actions, c.hidden);
    s.linear("stop_linear", 1, c.hidden);
}

// ------------------------------------------------------------------------------------------------ upload helpers
static uint16_t f2bf_host(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

struct Uploader {
    hcm_ctx* ctx;
    void* raw(const void* src, size_t bytes) {
        void* d = nullptr;
        if (hipMalloc(&d, bytes ? bytes : 16) != hipSuccess) throw std::runtime_error("hipMalloc failed for weights");
        if (bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("hipMemcpy H2D failed");
        ctx->dev_allocs.push_back(d);
        ctx->weight_bytes += bytes;
        return d;
    }
    float* f32(const std::vector<float>& v) { return (float*)raw(v.data(), v.size() * 4); }
    void* typed(const std::vector<float>& v, int dt) {
        if (dt == DT_F32) return raw(v.data(), v.size() * 4);
        std::vector<uint16_t> h(v.size());
        if (dt == DT_BF16) for (size_t i = 0; i < v.size(); ++i) h[i] = f2bf_host(v[i]);
        else for (size_t i = 0; i < v.size(); ++i) { _Float16 x = (_Float16)v[i]; std::memcpy(&h[i], &x, 2); }
        return raw(h.data(), h.size() * 2);
    }
};

static const HostTensor& T_(hcm_ctx* ctx, int model, const std::string& key) {
    auto it = ctx->sd[model].find(key);
    if (it == ctx->sd[model].end() || !it->second.loaded) throw std::runtime_error("missing state_dict key: " + key);
    return it->second;
}

static int round_up(int x, int m) { return (x + m - 1) / m * m; }

// conv weight OIHW (+ optional per-channel scale) -> [O][Kp] with k = (kh*KW+kw)*I + ci
static ConvW make_conv(int dt, Uploader& up, const HostTensor& w, const std::vector<float>* scale,
                       const std::vector<float>* bias, int cout_pad = 0, float fold = 1.f, int pos = -1) {
    ConvW c;
    c.dt = dt;
    c.fold = fold; c.calib_pos = pos;
    c.Cout = (int)w.shape[0]; c.Cin = (int)w.shape[1]; c.KH = (int)w.shape[2]; c.KW = (int)w.shape[3];
    c.K = c.KH * c.KW * c.Cin;
    c.Kp = round_up(c.K, 32);
    const int co = c.Cout;
    if (cout_pad > c.Cout) c.Cout = cout_pad;                 // zero rows: output channels that are exactly 0 (bias-free convs only)
    std::vector<float> r((size_t)c.Cout * c.Kp, 0.f);
    for (int o = 0; o < co; ++o) {
        const float sc = (scale ? (*scale)[o] : 1.f) * fold;
        for (int i = 0; i < c.Cin; ++i)
            for (int kh = 0; kh < c.KH; ++kh)
                for (int kw = 0; kw < c.KW; ++kw)
                    r[(size_t)o * c.Kp + (size_t)(kh * c.KW + kw) * c.Cin + i] =
                        w.f[(((size_t)o * c.Cin + i) * c.KH + kh) * c.KW + kw] * sc;
    }
    c.w = up.typed(r, dt);
    if (bias) c.bias = up.f32(*bias);
    return c;
}

// conv weight OIHW -> [O][K] rows with k = (kh*KW+kw)*I + ci as make_conv lays them out, then in MFMA-FRAGMENT order (device twin: pack_frag_kernel,
// bert_block.hip): the 16-byte chunk W[ct*16 + fr][ks*32 + fg*8 ..] at chunk index (ks*(O/16) + ct)*64 + fg*16 + fr.  O % 16 == 0, K % 32 == 0.
static void* make_conv_frag(int dt, Uploader& up, const HostTensor& w) {
    const int O = (int)w.shape[0], I = (int)w.shape[1], KH = (int)w.shape[2], KW = (int)w.shape[3], K = KH * KW * I;
    if (O % 16 || K % 32) return nullptr;
    std::vector<float> fo((size_t)O * K);
    for (int ks = 0; ks < K / 32; ++ks)
        for (int ct = 0; ct < O / 16; ++ct)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int o = ct * 16 + (l & 15), k = ks * 32 + (l >> 4) * 8 + e;
                    const int tap = k / I, ci = k - tap * I, kh = tap / KW, kw = tap - kh * KW;
                    fo[(((size_t)ks * (O / 16) + ct) * 64 + l) * 8 + e] = w.f[(((size_t)o * I + ci) * KH + kh) * KW + kw];
                }
    return up.typed(fo, dt);
}

// a linear layer's [N][K] weight in MFMA-FRAGMENT order (device twin: pack_frag_kernel, bert_block.hip): the 16-byte chunk W[ct*16 + fr][ks*32 + fg*8 ..]
// at chunk index (ks*(N/16) + ct)*64 + fg*16 + fr -- the 1 KB a wave loads for one operand fragment is contiguous.  N % 16 == 0, K % 32 == 0.
static void* make_linear_frag(int dt, Uploader& up, const HostTensor& w) {
    const int N = (int)w.shape[0], K = (int)w.shape[1];
    if (N % 16 || K % 32) return nullptr;
    std::vector<float> fo((size_t)N * K);
    for (int ks = 0; ks < K / 32; ++ks)
        for (int ct = 0; ct < N / 16; ++ct)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e)
                    fo[(((size_t)ks * (N / 16) + ct) * 64 + l) * 8 + e] = w.f[(size_t)(ct * 16 + (l & 15)) * K + ks * 32 + (l >> 4) * 8 + e];
    return up.typed(fo, dt);
}

// eval-mode BatchNorm2d folded into the preceding bias-free conv: y = conv(x)*g/sqrt(v+eps) + (b - m*g/sqrt(v+eps))
static void bn_fold(hcm_ctx* ctx, int model, const std::string& bn, int C, std::vector<float>& scale, std::vector<float>& bias, bool stem = false);
static ConvW make_conv_bn(hcm_ctx* ctx, int dt, Uploader& up, int model, const std::string& wkey, const std::string& bn, bool stem = false) {
    const HostTensor& w = T_(ctx, model, wkey);
    const int C = (int)w.shape[0];
    std::vector<float> scale, bias;
    bn_fold(ctx, model, bn, C, scale, bias, stem);
    return make_conv(dt, up, w, &scale, &bias);
}

static NormW make_norm(hcm_ctx* ctx, Uploader& up, int model, const std::string& p, int pad_to = 0) {
    NormW n;
    std::vector<float> g = T_(ctx, model, p + ".weight").f, b = T_(ctx, model, p + ".bias").f;
    if ((int)g.size() < pad_to) { g.resize(pad_to, 0.f); b.resize(pad_to, 0.f); }
    n.gamma = up.f32(g);
    n.beta = up.f32(b);
    n.C = (int)g.size();
    return n;
}

// Linear weight [N][K] (rows optionally concatenated from several tensors), K zero-padded to a multiple of 32.
// `perm` maps new column j -> source column perm[j] (-1: a zero column); its size is the new K.
// `col_scale` multiplies the first `scale_cols` (new) columns: the RGB trunk-feature columns of the projections behind a range-folded trunk.
static LinW make_linear(Uploader& up, const std::vector<const HostTensor*>& ws, const std::vector<const HostTensor*>& bs,
                        int dt, const std::vector<int>* perm = nullptr, int scale_cols = 0, float col_scale = 1.f) {
    LinW l;
    const int Ksrc = (int)ws[0]->shape[1];
    l.K = perm ? (int)perm->size() : Ksrc;                   // a perm may also widen the row: entries of -1 are zero columns
    l.Kp = round_up(l.K, 32);
    l.dt = dt;
    int N = 0;
    for (auto* w : ws) N += (int)w->shape[0];
    l.N = N;
    std::vector<float> r((size_t)N * l.Kp, 0.f);
    int row = 0;
    for (auto* w : ws) {
        const int n = (int)w->shape[0];
        for (int i = 0; i < n; ++i, ++row)
            for (int j = 0; j < l.K; ++j) {
                const int src = perm ? (*perm)[j] : j;
                if (src >= 0) r[(size_t)row * l.Kp + j] = w->f[(size_t)i * Ksrc + src] * (j < scale_cols ? col_scale : 1.f);
            }
    }
    l.w = up.typed(r, dt);
    if (!bs.empty()) {
        std::vector<float> b;
        for (auto* t : bs) b.insert(b.end(), t->f.begin(), t->f.end());
        l.bias = up.f32(b);
    }
    return l;
}

// 7x7x3 stem weights in the "row-run" K layout of the f32 fast gather (igemm.hip): k = kh*24 + kw*3 + ci, the three
// slots 21..23 of every run are zero; K = 7*24 = 168, rows padded to 192.  BN folded as usual.
static ConvW make_stem_rowrun(hcm_ctx* ctx, int dt, Uploader& up, int model, const std::string& wkey, const std::string& bn) {
    const HostTensor& w = T_(ctx, model, wkey);
    ConvW c;
    c.dt = dt;
    c.Cout = (int)w.shape[0]; c.Cin = 3; c.KH = 7; c.KW = 7;
    c.K = 7 * 24; c.Kp = 192;
    std::vector<float> r((size_t)c.Cout * c.Kp, 0.f), scale, bias;
    bn_fold(ctx, model, bn, c.Cout, scale, bias, true);
    for (int o = 0; o < c.Cout; ++o) {
        const float sc = scale[o];
        for (int ci = 0; ci < 3; ++ci)
            for (int kh = 0; kh < 7; ++kh)
                for (int kw = 0; kw < 7; ++kw)
                    r[(size_t)o * c.Kp + kh * 24 + kw * 3 + ci] = w.f[(((size_t)o * 3 + ci) * 7 + kh) * 7 + kw] * sc;
    }
    c.w = up.typed(r, dt);
    c.bias = up.f32(bias);
    return c;
}

// 7x7x3 stem weights for the packed-frame path (kernels.h: launch_pack_frame): k = kh*32 + kw*4 + ci; slots with
// ci = 3 or kw = 7 are zero; K = Kp = 224.  BN folded as usual.
static ConvW make_stem_packed(hcm_ctx* ctx, int dt, Uploader& up, int model, const std::string& wkey, const std::string& bn) {
    const HostTensor& w = T_(ctx, model, wkey);
    ConvW c;
    c.dt = dt;
    c.Cout = (int)w.shape[0]; c.Cin = 3; c.KH = 7; c.KW = 7;
    c.K = 224; c.Kp = 224;
    std::vector<float> scale, bias;
    bn_fold(ctx, model, bn, c.Cout, scale, bias, true);
    std::vector<float> r((size_t)c.Cout * c.Kp, 0.f);
    for (int o = 0; o < c.Cout; ++o)
        for (int ci = 0; ci < 3; ++ci)
            for (int kh = 0; kh < 7; ++kh)
                for (int kw = 0; kw < 7; ++kw)
                    r[(size_t)o * c.Kp + kh * 32 + kw * 4 + ci] = w.f[(((size_t)o * 3 + ci) * 7 + kh) * 7 + kw] * scale[o];
    c.w = up.typed(r, dt);
    c.bias = up.f32(bias);
    return c;
}

static ConvW make_c3ds(hcm_ctx* ctx, int dt, Uploader& up, const std::vector<int>& models, const std::string& p);
static TrunkW make_tv_trunk(hcm_ctx* ctx, Uploader& up, int model, const std::string& pre) {
    TrunkW t;
    const int dt = ctx->dt_rgb;
    t.gn = false;
    t.cin1 = 3;
    t.conv1 = make_conv_bn(ctx, dt, up, model, pre + "conv1.weight", pre + "bn1", true);
    t.conv1_rowrun = make_stem_rowrun(ctx, dt, up, model, pre + "conv1.weight", pre + "bn1");
    if (dt != DT_F32) t.conv1_packed = make_stem_packed(ctx, dt, up, model, pre + "conv1.weight", pre + "bn1");
    for (int li = 0; li < 4; ++li)
        for (int bi = 0; bi < RESNET50_BLOCKS[li]; ++bi) {
            const std::string p = pre + "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
            BottleneckW b;
            b.stride = (li > 0 && bi == 0) ? 2 : 1;
            b.c1 = make_conv_bn(ctx, dt, up, model, p + "conv1.weight", p + "bn1");
            b.c2 = make_conv_bn(ctx, dt, up, model, p + "conv2.weight", p + "bn2");
            b.c3 = make_conv_bn(ctx, dt, up, model, p + "conv3.weight", p + "bn3");
            if (bi == 0) { b.has_ds = true; b.ds = make_conv_bn(ctx, dt, up, model, p + "downsample.0.weight", p + "downsample.1"); }
            if (li == 0 && bi == 0 && dt != DT_F32) b.c3ds = make_c3ds(ctx, dt, up, {model}, p);
            t.blocks.push_back(b);
        }
    t.out_c = 2048;
    return t;
}

// 7x7 stem of the 1-channel depth trunk for the packed-frame path: k = kh*8 + kw (slot kw = 7 is zero), K = 56, Kp = 64;
// the rows of several models are concatenated along Cout (hi|lo pair: shared frame)
static ConvW make_depth_stem_packed(int dt, Uploader& up, const std::vector<const HostTensor*>& ws, float fold = 1.f) {
    ConvW c;
    c.dt = dt;
    c.fold = fold; c.calib_pos = 0;
    const int Co = (int)ws[0]->shape[0];
    c.Cout = Co * (int)ws.size(); c.Cin = 1; c.KH = 7; c.KW = 7; c.K = 56; c.Kp = 64;
    std::vector<float> r((size_t)c.Cout * c.Kp, 0.f);
    for (size_t g = 0; g < ws.size(); ++g)
        for (int o = 0; o < Co; ++o)
            for (int kh = 0; kh < 7; ++kh)
                for (int kw = 0; kw < 7; ++kw) r[((size_t)g * Co + o) * c.Kp + kh * 8 + kw] = ws[g]->f[((size_t)o * 7 + kh) * 7 + kw] * fold;
    c.w = up.typed(r, dt);
    return c;
}

// fp16 range folding of the GroupNorm trunks: position of a conv in the trunk topology -> hcm_ctx::depth_fold (model.h)
static int gn_pos(int block, int which) { return 1 + 4 * block + which; }      // which: 0 c1, 1 c2, 2 c3, 3 down-sample
static float gn_fold(hcm_ctx* ctx, int pos) { return ctx->dt_depth == DT_F16 ? ctx->depth_fold[pos] : 1.f; }
static TrunkW make_gn_trunk(hcm_ctx* ctx, Uploader& up, int model, const std::string& pre) {
    TrunkW t;
    const int dt = ctx->dt_depth;
    t.gn = true;
    t.groups = ctx->cfg.depth_baseplanes / 2;          // resnet_encoders.py:30
    t.cin1 = 1;
    const std::string bb = pre + "backbone.";
    auto W1 = [&](const std::string& k, int pos, int pad = 0) { return make_conv(dt, up, T_(ctx, model, k), nullptr, nullptr, pad, gn_fold(ctx, pos), pos); };
    t.conv1 = W1(bb + "conv1.0.weight", 0);
    if (dt != DT_F32) t.conv1_packed = make_depth_stem_packed(dt, up, {&T_(ctx, model, bb + "conv1.0.weight")}, gn_fold(ctx, 0));
    t.n_conv1 = make_norm(ctx, up, model, bb + "conv1.1");
    int blk = 0;
    for (int li = 0; li < 4; ++li)
        for (int bi = 0; bi < RESNET50_BLOCKS[li]; ++bi, ++blk) {
            const std::string p = bb + "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
            BottleneckW b;
            b.stride = (li > 0 && bi == 0) ? 2 : 1;
            b.c1 = W1(p + "convs.0.weight", gn_pos(blk, 0)); b.n1 = make_norm(ctx, up, model, p + "convs.1");
            b.c2 = W1(p + "convs.3.weight", gn_pos(blk, 1)); b.n2 = make_norm(ctx, up, model, p + "convs.4");
            b.c3 = W1(p + "convs.6.weight", gn_pos(blk, 2)); b.n3 = make_norm(ctx, up, model, p + "convs.7");
            if (bi == 0) {
                b.has_ds = true;
                b.ds = W1(p + "downsample.0.weight", gn_pos(blk, 3));
                b.nds = make_norm(ctx, up, model, p + "downsample.1");
            }
            t.blocks.push_back(b);
        }
    const int cc = depth_compress_channels(ctx->cfg), ccp = depth_compress_padded(ctx->cfg);
    t.compress = W1(pre + "compression.0.weight", hcm_ctx::kDepthPos - 1, ccp);
    t.n_compress = make_norm(ctx, up, model, pre + "compression.1", ccp);
    t.compress_true = ccp != cc ? cc : 0;
    t.out_c = t.compress.Cout;
    return t;
}

// ---- hi|lo pair of the GroupNorm depth trunk (see HighW::depth_pair)
static ConvW make_conv_pair(int dt, Uploader& up, const HostTensor& a, const HostTensor& b, bool concat_n, int cout_pad = 0, float fold = 1.f, int pos = -1) {
    ConvW c;
    c.dt = dt;
    c.fold = fold; c.calib_pos = pos;
    c.Cout = (int)a.shape[0]; c.Cin = (int)a.shape[1]; c.KH = (int)a.shape[2]; c.KW = (int)a.shape[3];
    c.K = c.KH * c.KW * c.Cin;
    c.Kp = round_up(c.K, 32);
    const int co = c.Cout;
    if (cout_pad > c.Cout) c.Cout = cout_pad;                 // per model: zero rows behind each model's own
    std::vector<float> r((size_t)2 * c.Cout * c.Kp, 0.f);
    const HostTensor* ws[2] = {&a, &b};
    for (int g = 0; g < 2; ++g)
        for (int o = 0; o < co; ++o)
            for (int i = 0; i < c.Cin; ++i)
                for (int kh = 0; kh < c.KH; ++kh)
                    for (int kw = 0; kw < c.KW; ++kw)
                        r[((size_t)g * c.Cout + o) * c.Kp + (size_t)(kh * c.KW + kw) * c.Cin + i] =
                            ws[g]->f[(((size_t)o * c.Cin + i) * c.KH + kh) * c.KW + kw] * fold;
    c.w = up.typed(r, dt);
    if (concat_n) c.Cout *= 2;          // shared input (stem): one ordinary conv with the output channels concatenated
    else c.groups = 2;
    return c;
}
static NormW make_norm_pair(hcm_ctx* ctx, Uploader& up, const std::string& p, int pad_to = 0) {
    NormW n;
    std::vector<float> g = T_(ctx, HCM_HIGH, p + ".weight").f, b = T_(ctx, HCM_HIGH, p + ".bias").f;
    std::vector<float> g2 = T_(ctx, HCM_LOW, p + ".weight").f, b2 = T_(ctx, HCM_LOW, p + ".bias").f;
    if ((int)g.size() < pad_to) { g.resize(pad_to, 0.f); b.resize(pad_to, 0.f); g2.resize(pad_to, 0.f); b2.resize(pad_to, 0.f); }
    g.insert(g.end(), g2.begin(), g2.end());
    b.insert(b.end(), b2.begin(), b2.end());
    n.C = (int)g.size();
    n.gamma = up.f32(g);
    n.beta = up.f32(b);
    return n;
}
static TrunkW make_gn_trunk_pair(hcm_ctx* ctx, Uploader& up, const std::string& pre) {
    TrunkW t;
    const int dt = ctx->dt_depth;
    t.gn = true;
    t.pair = true;
    t.groups = ctx->cfg.depth_baseplanes / 2;
    t.cin1 = 1;
    // (a position's fold is shared by the hi, the lo and the pair trunk: its range slot sees whichever of them runs)
    auto W2 = [&](const std::string& k, bool cat, int pos, int pad = 0) {
        return make_conv_pair(dt, up, T_(ctx, HCM_HIGH, k), T_(ctx, HCM_LOW, k), cat, pad, gn_fold(ctx, pos), pos);
    };
    const std::string bb = pre + "backbone.";
    t.conv1 = W2(bb + "conv1.0.weight", true, 0);
    if (dt != DT_F32) t.conv1_packed = make_depth_stem_packed(dt, up, {&T_(ctx, HCM_HIGH, bb + "conv1.0.weight"), &T_(ctx, HCM_LOW, bb + "conv1.0.weight")}, gn_fold(ctx, 0));
    t.n_conv1 = make_norm_pair(ctx, up, bb + "conv1.1");
    int blk = 0;
    for (int li = 0; li < 4; ++li)
        for (int bi = 0; bi < RESNET50_BLOCKS[li]; ++bi, ++blk) {
            const std::string p = bb + "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
            BottleneckW b;
            b.stride = (li > 0 && bi == 0) ? 2 : 1;
            b.c1 = W2(p + "convs.0.weight", false, gn_pos(blk, 0)); b.n1 = make_norm_pair(ctx, up, p + "convs.1");
            b.c2 = W2(p + "convs.3.weight", false, gn_pos(blk, 1)); b.n2 = make_norm_pair(ctx, up, p + "convs.4");
            b.c3 = W2(p + "convs.6.weight", false, gn_pos(blk, 2)); b.n3 = make_norm_pair(ctx, up, p + "convs.7");
            if (bi == 0) {
                b.has_ds = true;
                b.ds = W2(p + "downsample.0.weight", false, gn_pos(blk, 3));
                b.nds = make_norm_pair(ctx, up, p + "downsample.1");
            }
            t.blocks.push_back(b);
        }
    const int cc = depth_compress_channels(ctx->cfg), ccp = depth_compress_padded(ctx->cfg);
    t.compress = W2(pre + "compression.0.weight", false, hcm_ctx::kDepthPos - 1, ccp);
    t.n_compress = make_norm_pair(ctx, up, pre + "compression.1", ccp);
    t.compress_true = ccp != cc ? cc : 0;
    t.out_c = t.compress.Cout;      // per model
    return t;
}

// ---- hi|lo pair of the torchvision (BatchNorm-folded) RGB trunk: grouped convs with per-model folded scale/bias
// eval-mode BatchNorm2d of the RGB trunk folded into the preceding bias-free conv: y = conv(x) * scale + bias.  fp16 range folding
// (hcm_ctx::rgb_fold = s, a power of two, 1 unless a calibration asked for it): every activation of the trunk carries the factor s -- the stem
// multiplies its weights by s (its input, the frame, is unscaled), every later conv sees inputs that already carry s and only scales its bias.
static void bn_fold(hcm_ctx* ctx, int model, const std::string& bn, int C, std::vector<float>& scale, std::vector<float>& bias, bool stem) {
    const HostTensor& g = T_(ctx, model, bn + ".weight");
    const HostTensor& b = T_(ctx, model, bn + ".bias");
    const HostTensor& m = T_(ctx, model, bn + ".running_mean");
    const HostTensor& v = T_(ctx, model, bn + ".running_var");
    const float f = ctx->dt_rgb == DT_F16 ? ctx->rgb_fold : 1.f;
    scale.resize(C); bias.resize(C);
    for (int o = 0; o < C; ++o) {
        const float s = g.f[o] / std::sqrt(v.f[o] + 1e-5f);
        scale[o] = stem ? s * f : s;
        bias[o] = (b.f[o] - m.f[o] * s) * f;
    }
}
// Expansion conv3 and the block's 1x1 down-sample conv (both BN-folded) K-concatenated for the fused bottleneck launch that folds the
// down-sample into the expansion GEMM: rows [W3 * s3 | Wds * sds] of K = C1 + Cd, bias b3 + bds, one block per model (group).
static ConvW make_c3ds(hcm_ctx* ctx, int dt, Uploader& up, const std::vector<int>& models, const std::string& p) {
    ConvW c;
    c.dt = dt;
    c.groups = (int)models.size();
    const HostTensor& w3 = T_(ctx, models[0], p + "conv3.weight");
    const HostTensor& wd = T_(ctx, models[0], p + "downsample.0.weight");
    const int C3 = (int)w3.shape[0], C1 = (int)w3.shape[1], Cd = (int)wd.shape[1];
    c.Cout = C3; c.Cin = C1 + Cd; c.KH = c.KW = 1; c.K = c.Kp = C1 + Cd;
    std::vector<float> r((size_t)models.size() * C3 * c.Kp, 0.f), bias_all;
    for (size_t g = 0; g < models.size(); ++g) {
        const HostTensor& a = T_(ctx, models[g], p + "conv3.weight");
        const HostTensor& d = T_(ctx, models[g], p + "downsample.0.weight");
        std::vector<float> s3, b3, sd, bd;
        bn_fold(ctx, models[g], p + "bn3", C3, s3, b3);
        bn_fold(ctx, models[g], p + "downsample.1", C3, sd, bd);
        for (int o = 0; o < C3; ++o) {
            float* row = &r[(g * C3 + o) * c.Kp];
            for (int i = 0; i < C1; ++i) row[i] = a.f[(size_t)o * C1 + i] * s3[o];
            for (int i = 0; i < Cd; ++i) row[C1 + i] = d.f[(size_t)o * Cd + i] * sd[o];
            bias_all.push_back(b3[o] + bd[o]);
        }
    }
    c.w = up.typed(r, dt);
    c.bias = up.f32(bias_all);
    return c;
}
static ConvW make_conv_bn_pair(hcm_ctx* ctx, int dt, Uploader& up, const std::string& wkey, const std::string& bn) {
    const HostTensor* ws[2] = {&T_(ctx, HCM_HIGH, wkey), &T_(ctx, HCM_LOW, wkey)};
    ConvW c;
    c.dt = dt;
    c.groups = 2;
    c.Cout = (int)ws[0]->shape[0]; c.Cin = (int)ws[0]->shape[1]; c.KH = (int)ws[0]->shape[2]; c.KW = (int)ws[0]->shape[3];
    c.K = c.KH * c.KW * c.Cin;
    c.Kp = round_up(c.K, 32);
    std::vector<float> r((size_t)2 * c.Cout * c.Kp, 0.f), bias_all;
    for (int g = 0; g < 2; ++g) {
        std::vector<float> scale, bias;
        bn_fold(ctx, g == 0 ? HCM_HIGH : HCM_LOW, bn, c.Cout, scale, bias);
        bias_all.insert(bias_all.end(), bias.begin(), bias.end());
        for (int o = 0; o < c.Cout; ++o)
            for (int i = 0; i < c.Cin; ++i)
                for (int kh = 0; kh < c.KH; ++kh)
                    for (int kw = 0; kw < c.KW; ++kw)
                        r[((size_t)g * c.Cout + o) * c.Kp + (size_t)(kh * c.KW + kw) * c.Cin + i] =
                            ws[g]->f[(((size_t)o * c.Cin + i) * c.KH + kh) * c.KW + kw] * scale[o];
    }
    c.w = up.typed(r, dt);
    c.bias = up.f32(bias_all);
    return c;
}
// stem of the pair: shared RGB frame -> ONE conv with 2x64 output channels, in both K layouts (see make_stem_rowrun)
static void make_stem_pair(hcm_ctx* ctx, int dt, Uploader& up, const std::string& wkey, const std::string& bn, ConvW& plain, ConvW& rowrun,
                           ConvW& packed) {
    const HostTensor* ws[2] = {&T_(ctx, HCM_HIGH, wkey), &T_(ctx, HCM_LOW, wkey)};
    const int Co = (int)ws[0]->shape[0];
    plain = ConvW(); rowrun = ConvW(); packed = ConvW();
    plain.dt = rowrun.dt = packed.dt = dt;
    plain.Cout = rowrun.Cout = packed.Cout = 2 * Co; plain.Cin = rowrun.Cin = packed.Cin = 3;
    plain.KH = plain.KW = rowrun.KH = rowrun.KW = packed.KH = packed.KW = 7;
    plain.K = 147; plain.Kp = 160; rowrun.K = 7 * 24; rowrun.Kp = 192; packed.K = packed.Kp = 224;
    std::vector<float> rp((size_t)2 * Co * plain.Kp, 0.f), rr((size_t)2 * Co * rowrun.Kp, 0.f), rk((size_t)2 * Co * packed.Kp, 0.f), bias_all;
    for (int g = 0; g < 2; ++g) {
        std::vector<float> scale, bias;
        bn_fold(ctx, g == 0 ? HCM_HIGH : HCM_LOW, bn, Co, scale, bias, true);
        bias_all.insert(bias_all.end(), bias.begin(), bias.end());
        for (int o = 0; o < Co; ++o)
            for (int ci = 0; ci < 3; ++ci)
                for (int kh = 0; kh < 7; ++kh)
                    for (int kw = 0; kw < 7; ++kw) {
                        const float v = ws[g]->f[(((size_t)o * 3 + ci) * 7 + kh) * 7 + kw] * scale[o];
                        rp[((size_t)g * Co + o) * plain.Kp + (kh * 7 + kw) * 3 + ci] = v;
                        rr[((size_t)g * Co + o) * rowrun.Kp + kh * 24 + kw * 3 + ci] = v;
                        rk[((size_t)g * Co + o) * packed.Kp + kh * 32 + kw * 4 + ci] = v;
                    }
    }
    plain.w = up.typed(rp, dt); rowrun.w = up.typed(rr, dt);
    plain.bias = up.f32(bias_all); rowrun.bias = up.f32(bias_all);
    if (dt != DT_F32) { packed.w = up.typed(rk, dt); packed.bias = up.f32(bias_all); }
}
static TrunkW make_tv_trunk_pair(hcm_ctx* ctx, Uploader& up, const std::string& pre) {
    TrunkW t;
    const int dt = ctx->dt_rgb;
    t.gn = false;
    t.pair = true;
    t.cin1 = 3;
    make_stem_pair(ctx, dt, up, pre + "conv1.weight", pre + "bn1", t.conv1, t.conv1_rowrun, t.conv1_packed);
    for (int li = 0; li < 4; ++li)
        for (int bi = 0; bi < RESNET50_BLOCKS[li]; ++bi) {
            const std::string p = pre + "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
            BottleneckW b;
            b.stride = (li > 0 && bi == 0) ? 2 : 1;
            b.c1 = make_conv_bn_pair(ctx, dt, up, p + "conv1.weight", p + "bn1");
            b.c2 = make_conv_bn_pair(ctx, dt, up, p + "conv2.weight", p + "bn2");
            b.c3 = make_conv_bn_pair(ctx, dt, up, p + "conv3.weight", p + "bn3");
            if (bi == 0) { b.has_ds = true; b.ds = make_conv_bn_pair(ctx, dt, up, p + "downsample.0.weight", p + "downsample.1"); }
            if (li == 0 && bi == 0 && dt != DT_F32) b.c3ds = make_c3ds(ctx, dt, up, {HCM_HIGH, HCM_LOW}, p);
            t.blocks.push_back(b);
        }
    t.out_c = 2048;
    return t;
}

static SimpleCnnW make_simple_cnn(hcm_ctx* ctx, int dt, Uploader& up, int model, const std::string& pre, int cin, int h, int w) {
    SimpleCnnW s;
    s.cin = cin; s.h = h; s.w = w; s.h3 = simple_cnn_out_hw(h); s.w3 = simple_cnn_out_hw(w);
    s.c0 = make_conv(dt, up, T_(ctx, model, pre + "cnn.0.weight"), nullptr, &T_(ctx, model, pre + "cnn.0.bias").f);
    if (dt != DT_F32 && (cin == 1 || cin == 3) && w % 4 == 0) {
        // packed-frame layout of the 8x8/4 conv (forward.cpp simple_cnn): a kernel row is one contiguous run of 8 pixels
        const HostTensor& w0 = T_(ctx, model, pre + "cnn.0.weight");       // (32, cin, 8, 8)
        const int cp = cin == 3 ? 4 : 1, KR = 8 * cp;
        ConvW c;
        c.dt = dt; c.Cout = (int)w0.shape[0]; c.Cin = cin; c.KH = 8; c.KW = 8; c.K = 8 * KR; c.Kp = c.K;
        std::vector<float> r((size_t)c.Cout * c.Kp, 0.f);
        for (int o = 0; o < c.Cout; ++o)
            for (int ci = 0; ci < cin; ++ci)
                for (int kh = 0; kh < 8; ++kh)
                    for (int kw = 0; kw < 8; ++kw)
                        r[(size_t)o * c.Kp + kh * KR + kw * cp + ci] = w0.f[(((size_t)o * cin + ci) * 8 + kh) * 8 + kw];
        c.w = up.typed(r, dt);
        c.bias = up.f32(T_(ctx, model, pre + "cnn.0.bias").f);
        s.c0_packed = c;
    }
    s.c1 = make_conv(dt, up, T_(ctx, model, pre + "cnn.2.weight"), nullptr, &T_(ctx, model, pre + "cnn.2.bias").f);
    s.c2 = make_conv(dt, up, T_(ctx, model, pre + "cnn.4.weight"), nullptr, &T_(ctx, model, pre + "cnn.4.bias").f);
    if (dt != DT_F32 && cin == 1 && s.c0_packed.w) {
        s.c1_frag = make_conv_frag(dt, up, T_(ctx, model, pre + "cnn.2.weight"));
        s.c2_frag = make_conv_frag(dt, up, T_(ctx, model, pre + "cnn.4.weight"));
    }
    // Flatten() of the NCHW (B,32,h,w) tensor: source column c*S + s; ours is NHWC: s*32 + c
    const int S = s.h3 * s.w3;
    std::vector<int> perm((size_t)S * 32);
    for (int sp = 0; sp < S; ++sp)
        for (int c = 0; c < 32; ++c) perm[(size_t)sp * 32 + c] = c * S + sp;
    s.fc = make_linear(up, {&T_(ctx, model, pre + "cnn.7.weight")}, {&T_(ctx, model, pre + "cnn.7.bias")}, dt, &perm);
    return s;
}

static RnnW make_rnn(hcm_ctx* ctx, Uploader& up, int model, const std::string& pre, int early = -1) {
    RnnW r;
    const HostTensor& wih = T_(ctx, model, pre + "weight_ih_l0");
    const HostTensor& whh = T_(ctx, model, pre + "weight_hh_l0");
    const HostTensor& bih = T_(ctx, model, pre + "bias_ih_l0");
    const HostTensor& bhh = T_(ctx, model, pre + "bias_hh_l0");
    r.in = (int)wih.shape[1];
    const int H = ctx->cfg.hidden;
    r.early = r.in;
    if (ctx->cfg.rnn_type == HCM_LSTM) {
        if (early >= 0 && early <= r.in && early % 4 == 0 && (r.in - early) % 4 == 0) r.early = early;
        const int N = 4 * H, K = r.in + H, e = r.early;
        HostTensor cat;
        cat.shape = {N, K};
        cat.f.resize((size_t)N * K);
        for (int n = 0; n < N; ++n) {         // [W_ih[:, :e] | W_hh | W_ih[:, e:]]
            std::memcpy(&cat.f[(size_t)n * K], &wih.f[(size_t)n * r.in], e * 4);
            std::memcpy(&cat.f[(size_t)n * K + e], &whh.f[(size_t)n * H], H * 4);
            std::memcpy(&cat.f[(size_t)n * K + e + H], &wih.f[(size_t)n * r.in + e], (r.in - e) * 4);
        }
        HostTensor b;
        b.shape = {N};
        b.f.resize(N);
        for (int n = 0; n < N; ++n) b.f[n] = bih.f[n] + bhh.f[n];
        r.cat = make_linear(up, {&cat}, {&b}, DT_F32);
    } else {
        r.ih = make_linear(up, {&wih}, {&bih}, DT_F32);
        r.hh = make_linear(up, {&whh}, {&bhh}, DT_F32);
    }
    return r;
}

// spatial_embeddings (S,64) viewed as (1,64,h,w): channel c, token s reads E.flat[c*S + s]  (resnet_encoders.py:91-104,:218-231)
static float* make_pe_view(Uploader& up, const HostTensor& E) {
    const int S = (int)E.shape[0], C = (int)E.shape[1];
    std::vector<float> t((size_t)S * C);
    for (int s = 0; s < S; ++s)
        for (int c = 0; c < C; ++c) t[(size_t)s * C + c] = E.f[(size_t)c * S + s];
    return up.f32(t);
}

void prepare_high(hcm_ctx* ctx) {
    const hcm_config& c = ctx->cfg;
    const int M = HCM_HIGH;
    Uploader up{ctx};
    HighW& h = ctx->hi;
    h.rgb = make_tv_trunk(ctx, up, M, "rgb_encoder.cnn.");
    h.depth = make_gn_trunk(ctx, up, M, "depth_encoder.visual_encoder.");
    h.rgb_pe = make_pe_view(up, T_(ctx, M, "rgb_encoder.spatial_embeddings.weight"));
    h.depth_pe = make_pe_view(up, T_(ctx, M, "depth_encoder.spatial_embeddings.weight"));
    const int fs = depth_final_spatial(c);
    h.depth_S = fs * fs;
    const int cc = depth_compress_channels(c), ccp = depth_compress_padded(c);
    h.depth_C = ccp + 64;                       // token row: [compression channels | padding | position embedding]
    // BERT
    const std::string e = "embedding_layer.embeddings.";
    h.bert.word = up.f32(T_(ctx, M, e + "word_embeddings.weight").f);
    h.bert.pos = up.f32(T_(ctx, M, e + "position_embeddings.weight").f);
    {
        const HostTensor& tt = T_(ctx, M, e + "token_type_embeddings.weight");
        std::vector<float> t0(tt.f.begin(), tt.f.begin() + c.bert_hidden);
        h.bert.type0 = up.f32(t0);
    }
    h.bert.ln = make_norm(ctx, up, M, e + "LayerNorm");
    for (int i = 0; i < c.bert_layers; ++i) {
        const std::string p = "embedding_layer.encoder.layer." + std::to_string(i) + ".";
        BertLayerW L;
        L.qkv = make_linear(up, {&T_(ctx, M, p + "attention.self.query.weight"), &T_(ctx, M, p + "attention.self.key.weight"), &T_(ctx, M, p + "attention.self.value.weight")},
                            {&T_(ctx, M, p + "attention.self.query.bias"), &T_(ctx, M, p + "attention.self.key.bias"), &T_(ctx, M, p + "attention.self.value.bias")}, ctx->dt_bert);
        L.o = make_linear(up, {&T_(ctx, M, p + "attention.output.dense.weight")}, {&T_(ctx, M, p + "attention.output.dense.bias")}, ctx->dt_bert);
        static const bool want_frag = dev_env("HCM_BERT_FUSE") != nullptr;      // (development build: the fused attention-block experiment, forward.cpp bert())
        if (want_frag && (ctx->dt_bert == DT_F16 || ctx->dt_bert == DT_BF16) && L.o.N % 16 == 0 && L.o.K % 32 == 0 && L.o.K == L.o.Kp) {
            // the same values in MFMA-fragment order (bert_block.hip phase B; device twin: pack_frag_kernel): chunk of 8 at
            // ((ks * (N / 16) + ct) * 64 + fg * 16 + fr) * 8  <-  W[ct * 16 + fr][ks * 32 + fg * 8 ..]
            const std::vector<float>& wsrc = T_(ctx, M, p + "attention.output.dense.weight").f;
            const int N = L.o.N, K = L.o.K;
            std::vector<float> fo((size_t)N * K);
            for (int ks = 0; ks < K / 32; ++ks)
                for (int ct = 0; ct < N / 16; ++ct)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e)
                            fo[(((size_t)ks * (N / 16) + ct) * 64 + l) * 8 + e] = wsrc[(size_t)(ct * 16 + (l & 15)) * K + ks * 32 + (l >> 4) * 8 + e];
            L.o_frag = up.typed(fo, ctx->dt_bert);
        }
        L.ln1 = make_norm(ctx, up, M, p + "attention.output.LayerNorm");
        L.ff1 = make_linear(up, {&T_(ctx, M, p + "intermediate.dense.weight")}, {&T_(ctx, M, p + "intermediate.dense.bias")}, ctx->dt_bert);
        L.ff2 = make_linear(up, {&T_(ctx, M, p + "output.dense.weight")}, {&T_(ctx, M, p + "output.dense.bias")}, ctx->dt_bert);
        L.ln2 = make_norm(ctx, up, M, p + "output.LayerNorm");
        if (ctx->dt_bert == DT_F16 && dev_env("HCM_LN_FOLD")) {      // (opt-in experiment of the development build: forward.cpp bert())
            // LayerNorm folded into the GEMM that consumes it: W' = W diag(gamma), s = row sums of the fp16-ROUNDED W' (so that mean * s cancels exactly
            // what the MFMAs accumulate for a constant row), t = W beta + b
            auto fold = [&](const std::vector<const HostTensor*>& ws, const std::vector<const HostTensor*>& bs, const std::string& lnkey, LinW& out, float*& s_out, float*& t_out) {
                const std::vector<float>& ga = T_(ctx, M, lnkey + ".weight").f;
                const std::vector<float>& be = T_(ctx, M, lnkey + ".bias").f;
                const int K = (int)ws[0]->shape[1];
                int N = 0;
                for (auto* w : ws) N += (int)w->shape[0];
                out.N = N; out.K = K; out.Kp = round_up(K, 32); out.dt = DT_F16;
                std::vector<float> r((size_t)N * out.Kp, 0.f), sv((size_t)N, 0.f), tv((size_t)N, 0.f);
                int row = 0;
                for (size_t wi = 0; wi < ws.size(); ++wi) {
                    const int n = (int)ws[wi]->shape[0];
                    for (int i = 0; i < n; ++i, ++row) {
                        double sa = 0.0, ta = 0.0;
                        for (int j = 0; j < K; ++j) {
                            const float w = ws[wi]->f[(size_t)i * K + j];
                            const float wf = w * ga[j];
                            r[(size_t)row * out.Kp + j] = wf;
                            sa += (double)(float)(_Float16)wf;
                            ta += (double)w * (double)be[j];
                        }
                        sv[row] = (float)sa;
                        tv[row] = (float)(ta + (double)bs[wi]->f[i]);
                    }
                }
                out.w = up.typed(r, DT_F16);
                out.bias = nullptr;
                s_out = up.f32(sv);
                t_out = up.f32(tv);
            };
            fold({&T_(ctx, M, p + "intermediate.dense.weight")}, {&T_(ctx, M, p + "intermediate.dense.bias")}, p + "attention.output.LayerNorm", L.ff1_f, L.ff1_s, L.ff1_t);
            if (i > 0) {
                const std::string pp = "embedding_layer.encoder.layer." + std::to_string(i - 1) + ".";
                fold({&T_(ctx, M, p + "attention.self.query.weight"), &T_(ctx, M, p + "attention.self.key.weight"), &T_(ctx, M, p + "attention.self.value.weight")},
                     {&T_(ctx, M, p + "attention.self.query.bias"), &T_(ctx, M, p + "attention.self.key.bias"), &T_(ctx, M, p + "attention.self.value.bias")},
                     pp + "output.LayerNorm", L.qkv_f, L.qkv_s, L.qkv_t);
            }
        }
        h.bert.layers.push_back(L);
    }
    // Conv1d(k=1) weights (out,in,1) are linear layers over the token axis
    // (behind a range-folded RGB trunk the 2048 feature columns of a token row carry rgb_fold, the 64 position channels do not)
    const float rgb_unfold = ctx->dt_rgb == DT_F16 ? 1.f / ctx->rgb_fold : 1.f;
    h.rgb_kv = make_linear(up, {&T_(ctx, M, "rgb_kv.weight")}, {&T_(ctx, M, "rgb_kv.bias")}, ctx->dt_vla, nullptr, 2048, rgb_unfold);
    {
        std::vector<int> perm((size_t)h.depth_C);
        for (int n = 0; n < h.depth_C; ++n) perm[n] = depth_tok_src(n, cc, ccp);
        h.depth_kv = make_linear(up, {&T_(ctx, M, "depth_kv.weight")}, {&T_(ctx, M, "depth_kv.bias")}, ctx->dt_vla, &perm);
    }
    h.rgb_linear = make_linear(up, {&T_(ctx, M, "rgb_linear.2.weight")}, {&T_(ctx, M, "rgb_linear.2.bias")}, ctx->dt_vla, nullptr, 2048, rgb_unfold);
    {
        // depth_linear: Flatten of (B, dC, S) -> column c*S + s; ours [B][S][dC] -> s*dC + c
        const int S = h.depth_S, dC = h.depth_C;
        std::vector<int> perm((size_t)S * dC);
        for (int s = 0; s < S; ++s)
            for (int ch = 0; ch < dC; ++ch) {
                const int src = depth_tok_src(ch, cc, ccp);
                perm[(size_t)s * dC + ch] = src < 0 ? -1 : src * S + s;
            }
        h.depth_linear = make_linear(up, {&T_(ctx, M, "depth_linear.1.weight")}, {&T_(ctx, M, "depth_linear.1.bias")}, ctx->dt_vla, &perm);
    }
    // Visual_Ling_Attn
    VlaW& v = h.vla;
    v.vis_fc = make_linear(up, {&T_(ctx, M, "image_cm_encoder.vis_fc.weight")}, {&T_(ctx, M, "image_cm_encoder.vis_fc.bias")}, ctx->dt_vla);
    v.ins_fc = make_linear(up, {&T_(ctx, M, "image_cm_encoder.ins_fc.weight")}, {&T_(ctx, M, "image_cm_encoder.ins_fc.bias")}, ctx->dt_vla);
    v.ln = make_norm(ctx, up, M, "image_cm_encoder.layer_norm");
    for (int i = 0; i < c.vla_layers; ++i) {
        const std::string p = "image_cm_encoder.layers." + std::to_string(i) + ".";
        const std::string a = p + "enc_att.attention.";
        VlaLayerW L;
        L.q = make_linear(up, {&T_(ctx, M, a + "fc_q.weight")}, {&T_(ctx, M, a + "fc_q.bias")}, ctx->dt_vla);
        L.kv = make_linear(up, {&T_(ctx, M, a + "fc_k.weight"), &T_(ctx, M, a + "fc_v.weight")}, {&T_(ctx, M, a + "fc_k.bias"), &T_(ctx, M, a + "fc_v.bias")}, ctx->dt_vla);
        L.o = make_linear(up, {&T_(ctx, M, a + "fc_o.weight")}, {&T_(ctx, M, a + "fc_o.bias")}, ctx->dt_vla);
        L.ln_att = make_norm(ctx, up, M, p + "enc_att.layer_norm");
        L.ff1 = make_linear(up, {&T_(ctx, M, p + "pwff.fc1.weight")}, {&T_(ctx, M, p + "pwff.fc1.bias")}, ctx->dt_vla);
        L.ff2 = make_linear(up, {&T_(ctx, M, p + "pwff.fc2.weight")}, {&T_(ctx, M, p + "pwff.fc2.bias")}, ctx->dt_vla);
        L.ln_ff = make_norm(ctx, up, M, p + "pwff.layer_norm");
        if ((ctx->dt_vla == DT_F16 || ctx->dt_vla == DT_BF16) && c.d_model == 256 && c.d_ff % 256 == 0) {
            L.o_f = make_linear_frag(ctx->dt_vla, up, T_(ctx, M, a + "fc_o.weight"));
            L.ff1_f = make_linear_frag(ctx->dt_vla, up, T_(ctx, M, p + "pwff.fc1.weight"));
            L.ff2_f = make_linear_frag(ctx->dt_vla, up, T_(ctx, M, p + "pwff.fc2.weight"));
        }
        v.layers.push_back(L);
    }
    {
        // sinusoid_encoding_table(L, d): pe[p,2i] = sin(p / 10000^(2i/d)), pe[p,2i+1] = cos(same)  (common/utils.py:167-185)
        const int L = c.instr_len, d = c.d_model;
        std::vector<float> pe((size_t)L * d);
        for (int p = 0; p < L; ++p)
            for (int i = 0; i < d / 2; ++i) {
                const float div = std::pow(10000.0f, (2.0f * (float)i) / (float)d);
                const float ang = (float)p / div;
                pe[(size_t)p * d + 2 * i] = std::sin(ang);
                pe[(size_t)p * d + 2 * i + 1] = std::cos(ang);
            }
        v.pe = up.f32(pe);
    }
    // identical trunk weights in both state_dicts (the reference freezes the pretrained encoders of both models)?
    auto same_trunk = [&](const std::string& prefix) {
        if (!c.build_low || c.reserved[5]) return false;          // hcm_config.reserved[5]: run both models' trunks even when their weights are identical
        size_t n = 0;
        for (const auto& kv : ctx->sd[HCM_HIGH]) {
            if (kv.first.compare(0, prefix.size(), prefix) != 0) continue;
            auto it = ctx->sd[HCM_LOW].find(kv.first);
            if (it == ctx->sd[HCM_LOW].end() || it->second.f != kv.second.f) return false;
            ++n;
        }
        return n > 0;
    };
    h.rgb_shared = c.rgb_encoder == HCM_ENC_RESNET && same_trunk("rgb_encoder.cnn.");
    h.depth_shared = c.depth_encoder == HCM_ENC_RESNET && same_trunk("depth_encoder.visual_encoder.");
    if (c.build_low && c.depth_encoder == HCM_ENC_RESNET && !h.depth_shared && !(dev_env("HCM_NO_DEPTH_PAIR") && atoi(dev_env("HCM_NO_DEPTH_PAIR")))) {
        h.depth_pair = make_gn_trunk_pair(ctx, up, "depth_encoder.visual_encoder.");
        h.has_depth_pair = true;
    }
    if (c.build_low && c.rgb_encoder == HCM_ENC_RESNET && !h.rgb_shared && !(dev_env("HCM_NO_RGB_PAIR") && atoi(dev_env("HCM_NO_RGB_PAIR")))) {
        h.rgb_pair = make_tv_trunk_pair(ctx, up, "rgb_encoder.cnn.");
        h.has_rgb_pair = true;
    }
    h.rnn = make_rnn(ctx, up, M, "state_encoder.rnn.", c.rgb_out + c.depth_out);     // early: rgb_in | depth_in
    h.head_w = up.f32(T_(ctx, M, "linear.weight").f);
    h.head_b = up.f32(T_(ctx, M, "linear.bias").f);
}

void prepare_low(hcm_ctx* ctx) {
    const hcm_config& c = ctx->cfg;
    const int M = HCM_LOW;
    Uploader up{ctx};
    LowW& l = ctx->lo;
    l.depth_simple = c.depth_encoder == HCM_ENC_SIMPLECNN;
    l.rgb_simple = c.rgb_encoder == HCM_ENC_SIMPLECNN;
    if (!l.depth_simple) {
        l.depth = make_gn_trunk(ctx, up, M, "depth_encoder.visual_encoder.");
        // visual_fc: Flatten of (B, cc, fs, fs) -> c*S + s ; ours s*cc + c
        const int fs = depth_final_spatial(c), S = fs * fs, cc = depth_compress_channels(c), ccp = depth_compress_padded(c);
        std::vector<int> perm((size_t)S * ccp);
        for (int s = 0; s < S; ++s)
            for (int ch = 0; ch < ccp; ++ch) perm[(size_t)s * ccp + ch] = ch < cc ? ch * S + s : -1;
        l.depth_fc = make_linear(up, {&T_(ctx, M, "depth_encoder.visual_fc.1.weight")}, {&T_(ctx, M, "depth_encoder.visual_fc.1.bias")}, ctx->dt_depth, &perm);
    } else {
        l.depth_s = make_simple_cnn(ctx, ctx->dt_depth, up, M, "depth_encoder.", 1, c.depth_h, c.depth_w);
    }
    if (!l.rgb_simple) {
        l.rgb = make_tv_trunk(ctx, up, M, "rgb_encoder.cnn.");
        l.rgb_fc = make_linear(up, {&T_(ctx, M, "rgb_encoder.fc.weight")}, {&T_(ctx, M, "rgb_encoder.fc.bias")}, ctx->dt_rgb, nullptr, 2048,
                               ctx->dt_rgb == DT_F16 ? 1.f / ctx->rgb_fold : 1.f);
    } else {
        l.rgb_s = make_simple_cnn(ctx, ctx->dt_rgb, up, M, "rgb_encoder.", 3, c.rgb_h, c.rgb_w);
    }
    l.subtask_emb = up.f32(T_(ctx, M, "sub_task_embedding.weight").f);
    l.rnn = make_rnn(ctx, up, M, "state_encoder.rnn.", ctx->cfg.depth_out + ctx->cfg.rgb_out);   // early: depth | rgb; late: sub-task embedding
    l.lin_w = up.f32(T_(ctx, M, "linear.weight").f);
    l.lin_b = up.f32(T_(ctx, M, "linear.bias").f);
    l.stop_w = up.f32(T_(ctx, M, "stop_linear.weight").f);
    l.stop_b = up.f32(T_(ctx, M, "stop_linear.bias").f);
}

}  // namespace hcm

namespace hcm {
void prepare_cma(hcm_ctx* ctx) {
    const hcm_config& c = ctx->cfg;
    const hcm_cma_config& m = ctx->cma_cfg;
    const int M = HCM_CMA;
    Uploader up{ctx};
    CmaW& w = ctx->cma;
    w.rgb = make_tv_trunk(ctx, up, M, "rgb_encoder.cnn.");
    w.depth = make_gn_trunk(ctx, up, M, "depth_encoder.visual_encoder.");
    w.rgb_pe = make_pe_view(up, T_(ctx, M, "rgb_encoder.spatial_embeddings.weight"));
    w.depth_pe = make_pe_view(up, T_(ctx, M, "depth_encoder.spatial_embeddings.weight"));
    const int fs = depth_final_spatial(c);
    w.depth_S = fs * fs;
    const int cc = depth_compress_channels(c), ccp = depth_compress_padded(c);
    w.depth_C = ccp + 64;
    w.emb = up.f32(T_(ctx, M, "instruction_encoder.embedding_layer.weight").f);
    w.dirs = m.bidirectional ? 2 : 1;
    for (int d = 0; d < w.dirs; ++d) {
        const std::string sfx = d ? "_reverse" : "";
        const std::string p = "instruction_encoder.encoder_rnn.";
        const HostTensor& bih = T_(ctx, M, p + "bias_ih_l0" + sfx);
        const HostTensor& bhh = T_(ctx, M, p + "bias_hh_l0" + sfx);
        if (m.instr_rnn == HCM_GRU) {
            // GRU: n = tanh(W_in x + b_in + r * (W_hn h + b_hn)) -- b_hh stays with the recurrent product (it sits inside the reset gate's product)
            w.ih[d] = make_linear(up, {&T_(ctx, M, p + "weight_ih_l0" + sfx)}, {&bih}, DT_F32);
            w.ih[d].K = w.ih[d].Kp;
            w.hh[d] = make_linear(up, {&T_(ctx, M, p + "weight_hh_l0" + sfx)}, {&bhh}, DT_F32);
            continue;
        }
        HostTensor b;
        b.shape = bih.shape;
        b.f.resize(bih.f.size());
        for (size_t i = 0; i < b.f.size(); ++i) b.f[i] = bih.f[i] + bhh.f[i];
        w.ih[d] = make_linear(up, {&T_(ctx, M, p + "weight_ih_l0" + sfx)}, {&b}, DT_F32);
        w.ih[d].K = w.ih[d].Kp;          // the embedded tokens are stored zero-padded to Kp columns (E = 50 is not a vector multiple)
        w.hh[d] = make_linear(up, {&T_(ctx, M, p + "weight_hh_l0" + sfx)}, {}, DT_F32);
        {
            const HostTensor& whh = T_(ctx, M, p + "weight_hh_l0" + sfx);       // (4H, H)
            const int H4 = (int)whh.shape[0], H = (int)whh.shape[1];
            std::vector<float> t((size_t)H * H4);
            // [k][unit j][gate g]: the four gate weights of (input k, unit j) are one 16-byte load of the scan kernel
            for (int n = 0; n < H4; ++n)
                for (int k = 0; k < H; ++k) t[((size_t)k * H + (n % H)) * 4 + n / H] = whh.f[(size_t)n * H + k];
            w.hh_t[d] = up.f32(t);
        }
    }
    const float rgb_unfold = ctx->dt_rgb == DT_F16 ? 1.f / ctx->rgb_fold : 1.f;
    w.rgb_linear = make_linear(up, {&T_(ctx, M, "rgb_linear.2.weight")}, {&T_(ctx, M, "rgb_linear.2.bias")}, ctx->dt_vla, nullptr, 2048, rgb_unfold);
    {
        const int S = w.depth_S, dC = w.depth_C;          // Flatten of (B, dC, S): column c*S + s; ours [B][S][dC]
        std::vector<int> perm((size_t)S * dC);
        for (int s = 0; s < S; ++s)
            for (int ch = 0; ch < dC; ++ch) {
                const int src = depth_tok_src(ch, cc, ccp);
                perm[(size_t)s * dC + ch] = src < 0 ? -1 : src * S + s;
            }
        w.depth_linear = make_linear(up, {&T_(ctx, M, "depth_linear.1.weight")}, {&T_(ctx, M, "depth_linear.1.bias")}, ctx->dt_vla, &perm);
    }
    w.rgb_kv = make_linear(up, {&T_(ctx, M, "rgb_kv.weight")}, {&T_(ctx, M, "rgb_kv.bias")}, ctx->dt_vla, nullptr, 2048, rgb_unfold);
    {
        std::vector<int> perm((size_t)w.depth_C);
        for (int n = 0; n < w.depth_C; ++n) perm[n] = depth_tok_src(n, cc, ccp);
        w.depth_kv = make_linear(up, {&T_(ctx, M, "depth_kv.weight")}, {&T_(ctx, M, "depth_kv.bias")}, ctx->dt_vla, &perm);
    }
    w.state_q = make_linear(up, {&T_(ctx, M, "state_q.weight")}, {&T_(ctx, M, "state_q.bias")}, DT_F32);
    w.text_k = make_linear(up, {&T_(ctx, M, "text_k.weight")}, {&T_(ctx, M, "text_k.bias")}, DT_F32);
    w.text_q = make_linear(up, {&T_(ctx, M, "text_q.weight")}, {&T_(ctx, M, "text_q.bias")}, DT_F32);
    w.compress = make_linear(up, {&T_(ctx, M, "second_state_compress.0.weight")}, {&T_(ctx, M, "second_state_compress.0.bias")}, DT_F32);
    w.rnn1 = make_rnn(ctx, up, M, "state_encoder.rnn.");
    w.rnn2 = make_rnn(ctx, up, M, "second_state_encoder.rnn.");
    w.scale = T_(ctx, M, "_scale").f[0];                  // registered buffer: 1 / sqrt(hidden / 2) (cma.py:147)
    w.lin_w = up.f32(T_(ctx, M, "linear.weight").f);
    w.lin_b = up.f32(T_(ctx, M, "linear.bias").f);
    w.stop_w = up.f32(T_(ctx, M, "stop_linear.weight").f);
    w.stop_b = up.f32(T_(ctx, M, "stop_linear.bias").f);
}
}  // namespace hcm
