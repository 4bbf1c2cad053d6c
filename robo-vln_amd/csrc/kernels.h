// Launch-level interface between the model code (model.cpp / api.cpp) and the HIP kernels.
// All pointers are device pointers.  `dt` is HCM_F32 (T=float) or HCM_BF16 (T=bf16 storage).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include <cstdlib>

namespace hcm {

// A/B, tuning and profiling knobs (HCM_NO_*, HCM_IGEMM_*, HCM_GN_*, ...) exist in `make DEV=1` builds only (libhcm_dev.so, loaded with
// HCM_DEV_LIB=1): the shipped library reads three operational variables -- HCM_GRAPH, HCM_SERIAL, HCM_NO_CALIB -- and nothing else, so no
// stray environment variable can change which kernel a production step runs.
#ifdef HCM_DEV_KNOBS
inline const char* dev_env(const char* name) { return getenv(name); }
#else
inline const char* dev_env(const char*) { return nullptr; }
#endif

// One-time kernel-attribute setup is PER DEVICE (hipFuncSetAttribute acts on the current device): a process-wide flag would leave the
// second GPU of a process with the default 64 KB dynamic-LDS limit and its 130-160 KB launches failing.
struct DeviceOnce {
    std::atomic<unsigned long long> mask{0ull};
    static unsigned long long bit() { int d = 0; (void)hipGetDevice(&d); return 1ull << (d & 63); }
    bool need() const { return !(mask.load(std::memory_order_acquire) & bit()); }
    void done() { mask.fetch_or(bit(), std::memory_order_release); }
};

enum { DT_F32 = 0, DT_BF16 = 1, DT_I32 = 2, DT_I64 = 3, DT_U8 = 4, DT_F16 = 5 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

inline size_t dt_size(int dt) { return (dt == DT_BF16 || dt == DT_F16) ? 2 : 4; }
inline int dt_chunk(int dt) { return (dt == DT_BF16 || dt == DT_F16) ? 8 : 4; }

// Implicit-GEMM convolution / linear layer:  y[m][n] = act( sum_k A[m][k] * w[n][k] + bias[n] + res[m][n] )
//   A[m][k] is gathered on the fly from the NHWC activation x: m=(b,oy,ox), k=(kh,kw,ci).
//   A plain row-major GEMM is the degenerate case H=W=KH=KW=1, B=M, Cin=K.
struct IGemm {
    const void* x = nullptr;     // [B,H,W,*] channels at pixel stride xC (elements of T)
    const void* w = nullptr;     // [N][Kp] (T), k = (kh*KW+kw)*Cin + ci, rows zero-padded from K to Kp
    const float* bias = nullptr; // [N] f32 or null
    const void* res = nullptr;   // [M][ldr] (T) or null
    void* y = nullptr;           // [M][ldy] T, or f32 when out_f32
    int B = 1, H = 1, W = 1, Cin = 0, xC = 0;
    int Ho = 1, Wo = 1, KH = 1, KW = 1, stride = 1, pad = 0;
    int M = 0, N = 0, K = 0, Kp = 0;
    int ldy = 0, ldr = 0;
    int act = ACT_NONE;
    int out_f32 = 0;
    int res_f32 = 0;             // `res` is f32 [M][ldr] although T is a 16-bit type (launch_igemm's own kernels only: not gemm256, not grouped launches)
    // narrow-channel first layers (Cin = 1 or 3): x is the RAW frame of dtype x_src_dt (DT_F32 / DT_U8 / the storage
    // type), gathered element-wise and multiplied by x_scale; -1 = x is an ordinary T activation
    int x_src_dt = -1;
    float x_scale = 1.0f;
    int groups = 1;              // grouped launch: group g adds g*g_x / g*g_w / g*g_b / g*g_y ELEMENTS to x / w / bias / (y, res)
    long long g_x = 0, g_w = 0, g_b = 0, g_y = 0;
    int stride_w = 0;            // horizontal stride when it differs from `stride` (0 = same); packed-frame stem only
    // fused GroupNorm epilogue (launch_igemm checks: map of gn_hw | 64 pixels per sample, gn_cg channels per group with 8 | gn_cg | 128,
    // no bias): y = (conv - mean) * rstd * gamma + beta, then residual / activation as usual; 64x128 tiles are forced
    const float* gn_gamma = nullptr; const float* gn_beta = nullptr; int gn_cg = 0, gn_hw = 0; float gn_eps = 1e-5f;
    int x_rowrun = 0;            // f32 RGB stem: weights/K laid out as KH runs of 24 floats (21 taps + 3 zero pads)
    // horizontal half of MaxPool2d(3, 2, 1) fused into the epilogue: y is [B*Ho][Wo/2][ldy] (16-bit types, Wo a power of two <= 128, N % 64 == 0,
    // no residual); launch_vpool3s2 finishes the pool
    int hpool = 0;
    // GroupNorm statistics of the (bias-free) conv output taken from the f32 tile image in the epilogue: sum / sum of squares per
    // (sample, 64-row pixel block, group) -> cs_part[((b * cs_hw / 64 + block) * cs_G + group) * 2]; cs_hw % 64 == 0, 32 % cs_cg == 0.
    // launch_groupnorm_apply then normalises without a statistics pass over the map.
    float* cs_part = nullptr; int cs_cg = 0, cs_hw = 0, cs_G = 0;
    // GroupNorm ON LOAD (round 4; 16-bit GroupNorm trunks, large maps): x is the UN-normalised output of the producing conv, whose epilogue left
    // sum / sum of squares per (sample, 64-pixel block, group) in gi_stats (the cs_part layout, gi_ps blocks per sample).  The register-staged
    // kernel igemm_gnin_kernel normalises while it stages the operand -- relu?(x * scale[s][c] - shift[s][c] (+ gi_res)), rounded to T exactly as
    // gn_apply_kernel would have stored it -- so the stand-alone apply pass and its launch disappear.  gi_out (1x1 stride-1 convs with one channel
    // tile only): the normalised values are also written back (in place is fine: a pixel row is staged by exactly one workgroup) -- the block output
    // that the NEXT residual add needs.  x, gi_res and gi_out share the pixel stride xC and the group offset g_x.
    const float* gi_stats = nullptr; const float* gi_gamma = nullptr; const float* gi_beta = nullptr;
    int gi_ps = 0, gi_cg = 0, gi_G = 0, gi_hw = 0, gi_relu = 0; float gi_eps = 1e-5f;
    const void* gi_res = nullptr; void* gi_out = nullptr;
    // LayerNorm folded into the GEMMs around it (round 4, BERT in fp16 mode; forward.cpp bert()).  The tensor BETWEEN two GEMMs stays the pre-LayerNorm
    // sum u; the LayerNorm h = (u - mean) * rstd * gamma + beta is never materialised:
    //   consumer (gemm256f_kernel, QKV / FFN1): y = act(rstd[m] * (u W'^T - mean[m] * ln_s[n]) + ln_t[n]), W' = W diag(gamma) (the weight matrix passed
    //     in), ln_s[n] = sum_k W'[n][k], ln_t[n] = sum_k W[n][k] beta[k] + bias[n]; the row statistics are taken from the operand fragments inside the K
    //     loop (fdot2) and, with ln_stats_out, written as (mean, rstd) per row for ...
    //   ... the residual of the NEXT projection (igemm epilogues, attention-output / FFN2): res holds u, the epilogue adds
    //     (u - mean) * rstd * rln_gamma + rln_beta with (mean, rstd) from rln_stats.
    const float* ln_s = nullptr; const float* ln_t = nullptr; float* ln_stats_out = nullptr; float ln_eps = 1e-12f;
    // where the consumer's row statistics come from: ln_part_in = [M][ln_part_P][2] partial (sum, sum of squares) per 32-column slice of the row, left
    // by the PRODUCER's register epilogue (ln_part_out, igemm 8-wave tiles with 32-channel wave columns: force_choice) and combined in fixed order in
    // the consumer's prologue; nullptr = taken inside the consumer's own K loop (measured slower: 128 v_dot2 per K tile and wave)
    const float* ln_part_in = nullptr; int ln_part_P = 0; float* ln_part_out = nullptr;
    int force_gemm256 = 0;       // 1: gemm256f_kernel whatever the row count (layout constraints only; the folded-LayerNorm consumers)
    int force_choice = -1;       // >= 0: launch_igemm's (tile, staging variant) choice for this launch (the folded-LayerNorm producers)
    const float* rln_stats = nullptr; const float* rln_gamma = nullptr; const float* rln_beta = nullptr;
    int impl = 0;                // 0: launch_igemm chooses; 1: igemm_dma_kernel / igemm_kernel only; 2: the 256 x 256-tile kernel of gemm256.hip only; 3: the few-row kernel of skinny.hip only
};
hipError_t launch_igemm(const IGemm& g, int dt, hipStream_t s);
bool igemm_gnin_ok(const IGemm& g, int dt);          // can launch_igemm take this conv with a GroupNorm-on-load operand (gi_*)?
// 256 x 256-tile, 8-phase GEMM for wide token-major linear layers (gemm256.hip); bit-identical to launch_igemm's kernels
bool gemm256_applicable(const IGemm& g, int dt);
// few-row GEMM, a wave per 16 x 16 output tile with its operands straight from L2 into registers (skinny.hip); bit-identical to launch_igemm's kernels
bool skinny_applicable(const IGemm& g, int dt);
hipError_t launch_skinny(const IGemm& g, int dt, hipStream_t s);
hipError_t launch_gemm256(const IGemm& g, int dt, hipStream_t s);
hipError_t gemm256_prof_read(unsigned long long* host, bool reset);   // phase counters of the profiled 256 x 256 GEMM builds: [16 workgroups][8 waves][8]
// Fused 3x3 conv (C1 -> C1, bias, ReLU) + 1x1 expansion (C1 -> 4*C1, bias, + identity, ReLU) of a BN-folded bottleneck, 16-bit types,
// C1 = 64 or 128 (igemm.hip: bneck23_kernel).  Weights as for launch_igemm: w2 [C1][9*C1] (k = (kh*3+kw)*C1 + ci), w3 [4*C1][C1].
struct Bneck23 {
    const void* x = nullptr;     // [B,H,W,*] C1 channels at pixel stride xC
    const void* w2 = nullptr; const float* b2 = nullptr;
    const void* w3 = nullptr; const float* b3 = nullptr;
    const void* res = nullptr;   // identity [M][ldr]
    void* y = nullptr;           // [M][ldy]
    int B = 1, H = 1, W = 1, C1 = 0, xC = 0, stride = 1, ldy = 0, ldr = 0;
    int groups = 1;              // hi|lo pair: element offsets per group
    long long g_x = 0, g_w2 = 0, g_b2 = 0, g_w3 = 0, g_b3 = 0, g_y = 0;
    // optional: the NEXT block's 1x1 reduction relu(y @ w1^T + b1) from the output tile in the same launch (bneck231_kernel):
    // w1 [CN][4*C1], CN = 64 or 128, o1 [M][ldo]
    const void* w1 = nullptr; const float* b1 = nullptr; void* o1 = nullptr;
    int CN = 0, ldo = 0;
    long long g_w1 = 0, g_b1 = 0, g_o1 = 0;
    // optional (with w1): the block's own 1x1 down-sample conv folded into the expansion GEMM -- w3 is then [4*C1][C1 + 64*KD] = [W3 | Wds],
    // b3 = b3 + bds, res is unused; xd = block input [B,H,W,*] (64*KD channels at pixel stride xdC), same stride as the 3x3 conv
    const void* xd = nullptr; int xdC = 0, KD = 0; long long g_xd = 0;
};
// A run of up to 6 identity bottlenecks (no down-sample, stride 1) of a GroupNorm trunk on 8 x 8 maps with 512 / 128 channels and 16 groups
// per trunk -- the depth encoder's layer3 behind its first block -- as ONE launch, a workgroup per (sample, trunk): igemm.hip depth_l3_kernel.
struct DepthL3 {
    const void* x = nullptr; void* y = nullptr;      // [B][64][ld] activations (block input / output of the run), trunk g at channels [g * 512, +512)
    int ld = 0, B = 0, groups = 1, nblocks = 0;
    const void* w1[6] = {}; const void* w2[6] = {}; const void* w3[6] = {};          // [groups][128][512], [groups][128][1152], [groups][512][128]
    const float* g1[6] = {}; const float* b1[6] = {}; const float* g2[6] = {}; const float* b2[6] = {}; const float* g3[6] = {}; const float* b3[6] = {};
    float eps1[6] = {}, eps2[6] = {}, eps3[6] = {};
};
hipError_t launch_depth_l3(const DepthL3& d, int dt, hipStream_t s);
// Identity bottlenecks of the depth GroupNorm trunk's layer1 (side 32, C 128, CM 32) / layer2 (side 16, C 256, CM 64), a whole sample of one trunk
// per workgroup (depth_blk.hip, round 4).  x / y: [B][side*side][ld] activations, trunk g at channels [g * C, +C); weights [groups][CM][C],
// [groups][CM][9*CM] (k = tap * CM + ci), [groups][C][CM]; no biases; 16 GroupNorm groups per trunk.  nblocks > 1 needs x != y.
struct DepthBlk {
    const void* x = nullptr; void* y = nullptr;
    int ld = 0, B = 0, groups = 1, nblocks = 0, side = 0, C = 0, CM = 0;
    const void* w1[4] = {}; const void* w2[4] = {}; const void* w3[4] = {};
    const float* g1[4] = {}; const float* b1[4] = {}; const float* g2[4] = {}; const float* b2[4] = {}; const float* g3[4] = {}; const float* b3[4] = {};
    float eps1[4] = {}, eps2[4] = {}, eps3[4] = {};
};
hipError_t launch_depth_blk(const DepthBlk& d, int dt, hipStream_t s);
hipError_t launch_bneck23(const Bneck23& b, int dt, hipStream_t s);

// While tuning is on, the first launch of every new (shape, dtype) times all tile/staging variants on the real
// operands and caches the fastest (process-wide); hcm_finalize() runs one tuning step at max_batch.
void igemm_set_tuning(bool on);
size_t igemm_tuned_shapes();
hipError_t launch_spin(unsigned long long ticks, hipStream_t s);
hipError_t launch_mark(unsigned long long* slot, hipStream_t s);     // development aid: wall-clock stamp in stream order (HCM_MARKS=1)
hipError_t igemm_prof_read(unsigned long long* host8, bool reset);   // HCM_IGEMM_PROF=1 phase counters

// 7x7/2 pad-3 stem on a 16-bit trunk: the raw RGB frame (f32 or uint8, NHWC3) is first packed ONCE into a zero-bordered
// 4-channel frame of the storage type, [B][H+6][W+8][4] (3 px border left/top, 5 right, 3 bottom; channel 3 = 0; values
// multiplied by `scale`).  In that frame a kernel row of an output pixel is ONE contiguous, 16-byte-aligned run of
// 8 px x 4 ch = 32 elements (7 real taps + 1 whose weights are zero), so the stem becomes an ordinary LDS-DMA implicit
// GEMM over "virtual pixels" of 8 elements: H' = H+6, W' = (W+8)/2, xC = 8, Cin = 32, KH = 7, KW = 1, stride 2 down /
// 1 across, pad 0, K = 7*32 -- no element-wise gather, no conversion in the GEMM, no bounds cases.
hipError_t launch_pack_frame(const void* x, int src_dt, void* y, int dt, int B, int H, int W, float scale, hipStream_t s, int border = 1);
// border = 0: plain [B][H][W][4] packing (SimpleCNN's un-padded 8x8/4 first conv: a kernel row is 8 px x 4 ch = 32 elements,
// a "virtual pixel" of the GEMM the 4-pixel stride = 16 elements)
inline size_t pack_frame_elems(int B, int H, int W) { return (size_t)B * (H + 6) * (W + 8) * 4 + 64; }
// round 6 (stem.hip): conv1 (7x7/2, BN folded) + ReLU + MaxPool2d(3, 2, 1) on the packed frame as ONE launch, weights in registers, the frame
// streamed through an LDS ring; y = pooled map [B][H/4][W/4][C].  W == 256, H % 4 == 0, C % 64 == 0; bit-identical to the packed stem conv
// with the horizontal pool epilogue + launch_vpool3s2.
bool rgb_stem_pool_ok(int dt, int H, int W, int C, int Kp);
// w1 / b1 / o1 non-null: also layer1 block 0's 1x1 reduction of the pooled map, 64 -> 64 per 64-channel group (w1 [C][64], k = the group's own channels),
// + bias + ReLU -> o1 [B][H/4][W/4][C]; bit-identical to the grouped 1x1 launch.
hipError_t launch_rgb_stem_pool(const void* pk, const void* w, const float* bias, void* y, int dt, int B, int H, int W, int C, hipStream_t s,
                                const void* w1 = nullptr, const float* b1 = nullptr, void* o1 = nullptr);
hipError_t launch_avgpool2_f32(const float* x, void* y, int dt, int B, int H, int W, hipStream_t s);
hipError_t launch_avgpool2_f32_padded(const float* x, void* y, int dt, int B, int H, int W, hipStream_t s);   // -> [B][H/2+6][W/2+8]
// vertical half of MaxPool2d(3, 2, 1): x [B][H][W][C] -> y [B][(H+1)/2][W][C] (rows 2p-1, 2p, 2p+1)
hipError_t launch_vpool3s2(const void* x, void* y, int dt, int B, int H, int W, int C, hipStream_t s);
hipError_t launch_maxpool3x3s2(const void* x, void* y, int dt, int B, int H, int W, int C, int Ho, int Wo, hipStream_t s);
// ... over relu(GroupNorm(x)) with the normalisation applied on load from the producing conv's epilogue statistics (bit-identical to the apply pass + the pool)
bool maxpool_gn_ok(int dt, int C, int G);
hipError_t launch_maxpool3x3s2_gn(const void* x, void* y, const float* gamma, const float* beta, const float* part, int PS, float eps, int G, int dt, int B, int H,
                                  int W, int C, int Ho, int Wo, hipStream_t s);
// adaptive average pool NHWC [B,H,W,C] -> [B,OH,OW,*] written with row stride ldy (elements) per output pixel
hipError_t launch_adaptive_avgpool(const void* x, void* y, int dt, int B, int H, int W, int C, int OH, int OW, int ldy, hipStream_t s,
                                   int ldx = 0 /* input pixel stride in elements, 0 = C */);
// mean over S rows: x [B,S,ldx(>=C)] (T) -> y [B, ldy] (T or f32 when out_f32) columns [0,C)
// lens (optional, device, [B]): sample b averages its first clamp(lens[b], 1, S) rows only (ragged instruction batches)
hipError_t launch_mean_rows(const void* x, void* y, int dt, int B, int S, int C, int ldx, int ldy, int out_f32, hipStream_t s,
                            const int* lens = nullptr);
// write a constant f32 table tab[S][C] into columns of y [B,S,ldy] (T)
hipError_t launch_fill_cols(const float* tab, void* y, int dt, int B, int S, int C, int ldy, hipStream_t s);

// GroupNorm over NHWC x [B,HW,C] in place: x = relu?( (x-mean)*rstd*gamma+beta (+res) ); stats scratch [B*G*2] f32
// pixel chunks per sample of the two-launch GroupNorm (0: single-launch slab kernel); `stats` must hold gn_stats_floats()
inline int gn_partials(int HW) { return HW >= 256 ? (HW / 64 < 16 ? HW / 64 : 16) : 0; }
inline size_t gn_stats_floats(int B, int HW, int G) {
    const int P = gn_partials(HW), PS = HW % 64 == 0 ? HW / 64 : 0;          // two-launch chunks / 64-row blocks of the epilogue statistics
    const int n = P > PS ? P : PS;
    return (size_t)B * (n > 0 ? n : 1) * G * 2;
}
// second half of the two-launch GroupNorm alone: statistics come from `part` = PS partial sums per (sample, group) (launch_igemm's cs_part)
hipError_t launch_groupnorm_apply(void* x, const void* res, const float* gamma, const float* beta, const float* part, int PS, int dt, int B,
                                  int HW, int C, int G, float eps, int relu, hipStream_t s);
// ... with an UN-normalised residual: x = relu?(GN(x) + round_T(GN2(res))) (the stage-first bottlenecks' down-sample branch; bit-identical to two apply passes)
hipError_t launch_groupnorm_apply2(void* x, const void* res, const float* gamma, const float* beta, const float* part, const float* gamma2, const float* beta2,
                                   const float* part2, int PS, int dt, int B, int HW, int C, int G, float eps, float eps2, int relu, hipStream_t s);
bool groupnorm_apply_ok(int dt, int HW, int C, int G);
hipError_t launch_groupnorm(void* x, const void* res, const float* gamma, const float* beta, float* stats,
                            int dt, int B, int HW, int C, int G, float eps, int relu, hipStream_t s, int cg_true = 0);   // cg_true > 0: real channels per group, the rest are zero padding
// LayerNorm rows: y = LN(x (+res)) * gamma + beta (+ post[row % post_rows][:])
hipError_t launch_layernorm(const void* x, const void* res, const float* gamma, const float* beta,
                            const float* post, int post_rows, void* y, int dt, int rows, int D, float eps, hipStream_t s);
// LayerNorm of an f32 tensor with two outputs: y16 (T: the next GEMM's operand) and y32 (f32: the residual stream); D = 768 / 256 / 512
hipError_t launch_layernorm_f32in(const float* x, const float* gamma, const float* beta, void* y16, float* y32, int dt, int rows, int D, float eps,
                                  hipStream_t s, float* stats = nullptr);   // stats: (mean, rstd) per row out; y32 may be null then
// BERT embeddings: y[b,l,:] = LN(word[id] + pos[l] + type0) ; tables f32
hipError_t launch_bert_embed(const void* ids, int ids_dt, const float* word, const float* pos, const float* type0,
                             const float* gamma, const float* beta, void* y, int dt, int B, int L, int D, int vocab,
                             float eps, hipStream_t s);
// softmax(Q K^T / sqrt(64)) V, head dim 64.  q batch index = b % q_batch_mod (shared queries)
// klens (optional, device, [B]): sample b attends over its first clamp(klens[b], 1, Lk) keys only; rows keep the stride Lk.  The result
// for those queries equals a launch with Lk = klens[b] on the unpadded tensors bit for bit.
hipError_t launch_attention(const void* q, const void* k, const void* v, void* out, int dt, int B, int heads,
                            int Lq, int Lk, int ldq, int ldk, int ldv, int ldo, int q_batch_mod, hipStream_t s, const int* klens = nullptr);

// BERT attention block tail in one launch (bert_block.hip): y = LayerNorm(softmax(Q K^T / 8) V Wo^T + bo + residual) per sample, bit-identical to
// launch_attention + launch_igemm(+ residual) + launch_layernorm[_f32in].  res32 / y32: the bf16 mode's f32 residual stream.
bool bert_attn_block_ok(int dt, int D, int heads, int L, int Kp, int ldq);
// wo_frag: W_o in fragment order (launch_pack_frag; host-side twin: weights.cpp frag_order) -- a wave's 16-column x 32-k MFMA operand fragment is
// 1 KB contiguous, so that fragment loads straight from L2 cost 8 cache-line look-ups instead of 64
hipError_t launch_pack_frag(const void* w, void* out, int dt, int N, int K, hipStream_t s);
hipError_t launch_bert_attn_block(const void* qkv, int ldq, const void* wo_frag, const float* bo, const void* res, const float* res32, const float* gamma,
                                  const float* beta, void* y, float* y32, int dt, int B, int L, const int* klens, float eps, hipStream_t s);

// One cross-modal layer after the projections, both visual streams in one launch (vla_fused.hip): [attention when Lk <= 32] ->
// fc_o + residual I -> LayerNorm -> fc1 + ReLU -> fc2 + residual -> LayerNorm [-> mean over the instruction's tokens].  16-bit storage types,
// d_model 256, 4 heads, d_ff a multiple of 256.  Per-stream pointers are indexed by blockIdx.y.
struct VlaPost {
    const void* q = nullptr;                 // fc_q(I)  [B][L][256]   (in-kernel attention only)
    const void* I = nullptr;                 // query stream = residual [B][L][256]
    const void* kv[2] = {nullptr, nullptr};  // fc_k | fc_v of the keys  [B][Lk][512]   (in-kernel attention only)
    const void* att[2] = {nullptr, nullptr}; // attention output [B][L][256]            (when the attention ran as its own launch)
    void* out[2] = {nullptr, nullptr};       // layer output [B][L][256]
    float* pooled[2] = {nullptr, nullptr};   // mean over the tokens -> pooled[b * ld_pool + c] (only when L <= 80), or null
    const void *wo = nullptr, *w1 = nullptr, *w2 = nullptr;       // [256][256], [d_ff][256], [256][d_ff] (K contiguous), or all three in fragment order:
    int wfrag = 0;               // 1: wo / w1 / w2 as launch_pack_frag writes them -- the weights go straight from L2 into registers (vla_post_wf_kernel)
    const float *bo = nullptr, *b1 = nullptr, *b2 = nullptr, *g1 = nullptr, *be1 = nullptr, *g2 = nullptr, *be2 = nullptr;
    const int* lens = nullptr;               // per-environment token counts (ragged batches) or null
    int B = 0, L = 0, Lk[2] = {0, 0}, d_ff = 1024, fuse_att = 0, ld_pool = 0, streams = 2;
    int dbg = 0;
    unsigned* calib = nullptr;   // calibration forward: {max |x| bits, non-finite count} over everything the kernel rounds to the storage type
};
bool vla_post_ok(int dt, int d_model, int heads, int d_ff);
hipError_t launch_vla_post(const VlaPost& p, int dt, hipStream_t s);

// RNN input assembly: xh[b][x_cols + j] = h_in[0][b][j] * mask[b]   (f32)
hipError_t launch_rnn_prep(const float* h_in, const float* mask, float* xh, int B, int Hd, int ld, int col0, hipStream_t s);
struct Heads {            // up to two small linear heads on the new hidden state
    const float* w0 = nullptr; const float* b0 = nullptr; float* out0 = nullptr; int r0 = 0; int ld0 = 0;
    const float* w1 = nullptr; const float* b1 = nullptr; float* out1 = nullptr; int r1 = 0; int ld1 = 0;
    // optional, fused act() step: pred[b] = argmax(out0 row) (first maximal index, hierarchical_trainer.py:1098) and the low-level model's
    // sub-task embedding row emb[pred] written to emb_out[b*emb_ld ..] -- saves two launches on the step's serial tail
    int64_t* pred = nullptr; const float* emb = nullptr; float* emb_out = nullptr; int emb_dim = 0, emb_ld = 0, emb_rows = 0;    // optional overflow guard: incremented once per sample whose gate pre-activations are not all finite (an fp16 overflow or a NaN
    // anywhere upstream ends up there: the squashing cell would turn it into finite garbage) -- hcm_query(HCM_STEP_NONFINITE)
    unsigned* bad = nullptr;
};
// LSTM cell from pre-activations gates [B][4H] (i,f,g,o), c_in = h_in[1]*mask; writes h_out (2,B,H)
hipError_t launch_lstm_cell(const float* gates, const float* h_in, const float* mask, float* h_out, int B, int Hd,
                            const Heads& heads, hipStream_t s);
// GRU cell from gi, gh [B][3H] (r,z,n); h = h_in[0]*mask; writes h_out (1,B,H)
hipError_t launch_gru_cell(const float* gi, const float* gh, const float* h_in, const float* mask, float* h_out,
                           int B, int Hd, const Heads& heads, hipStream_t s);
// split-K: fixed-order sum of S f32 partial results [S][M][N] + bias + activation (see Fwd::linear)
// How many K slices a skinny long-K linear layer is cut into (forward.cpp Fwd::linear and hcm_op_linear use the same rule, so an operator call
// reproduces the model path bit for bit): powers of two while the (64 x 32-tile) grid stays under 256 workgroups and a slice keeps >= 256 columns
// in whole 64-column K tiles.  (Round 6 tried one odd factor on top where the doubling stops on divisibility -- SimpleCNN's 25088 = 2^9 x 49
// columns stop at 8 slices -- and measured the GEMM at 15-19 us for 8, 14, 28 and 56 slices alike: it is not a per-workgroup chain, DESIGN_LOG R6.14.)
static inline int splitk_slices(int M, int N, int K, int CHw, bool has_res) {
    int S = 1;
    if (has_res || M > 256 || K < 2048) return 1;
    const long blocks = (long)((M + 63) / 64) * ((N + 31) / 32);
    while (S < 16 && blocks * S < 256 && K % (2 * S * 64) == 0 && K / (2 * S) >= 256) S *= 2;
    if (K % (S * CHw)) S = 1;
    return S;
}
hipError_t launch_splitk_reduce(const float* part, const float* bias, void* y, int dt, int S, int M, int N, int ldy, int act, int out_f32,
                                hipStream_t s);
// CMANet (models/cma.py) pieces: instruction embedding + lengths, one packed-LSTM time step, single-query attention
hipError_t launch_instr_embed(const void* ids, int ids_dt, const float* table, float* x, int* lengths, int B, int L, int E, int ldx,
                              int vocab, hipStream_t s);
hipError_t launch_instr_lstm_cell(const float* pre, const float* gh, float* h, float* c, const int* lengths, float* out, int t, int B,
                                  int L, int Hd, int ld_out, int col0, hipStream_t s);
hipError_t launch_instr_gru_cell(const float* pre, const float* gh, float* h, const int* lengths, float* out, int t, int B, int L, int Hd, int ld_out,
                                 int col0, hipStream_t s);   // nn.GRU step of the packed instruction encoder (gate order r, z, n; gh carries b_hh)
// all L steps of both directions in one launch (H == 256): wt = W_hh transposed [H][4H]
hipError_t launch_instr_lstm_scan(const float* pre0, const float* pre1, const float* wt0, const float* wt1, const int* lengths, float* out,
                                  int B, int L, int H, int dirs, int ld_out, hipStream_t s);
hipError_t launch_attn1q(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const int* lengths, float* out,
                         int ldo, int B, int S, int D, int Dv, float scale, hipStream_t s);
// pred[b] = argmax_j logits[b*ld + j] (first max), int64
hipError_t launch_argmax(const float* logits, int64_t* pred, int B, int n, int ld, hipStream_t s);
// xh[b][col0 + j] = emb[subtask[b]][j]
hipError_t launch_embed_rows(const float* emb, const int64_t* idx, float* y, int B, int D, int ld, int col0, int nrows, hipStream_t s);
// fp16 range calibration: slot[0] = max(slot[0], bits of max |x|), slot[1] += number of non-finite elements; x is [rows][ld], cols used
hipError_t launch_absmax(const void* x, int dt, int rows, int cols, int ld, unsigned* slot, hipStream_t s);
// generic converts
hipError_t launch_convert_to_f32(const void* x, int dt, float* y, size_t n, hipStream_t s);
hipError_t launch_convert_from_f32(const float* x, void* y, int dt, size_t n, hipStream_t s);
// simplecnn.hip: Conv2d(1, 32, 8, stride 4) + bias + activation from the raw f32 depth frame (B,H,H,1), w = [32][64] in the 16-bit storage type
// SimpleDepthCNN's three convolutions in one launch (simplecnn.hip): the two intermediate maps live in LDS; w1f / w2f in fragment order (launch_pack_frag)
bool simplecnn3_ok(int dt, int H);
hipError_t launch_simplecnn3(const float* x, const void* w0, const float* b0, const void* w1f, const float* b1, const void* w2f, const float* b2, void* y,
                             int dt, int B, int H, hipStream_t s);
bool depth_conv8x8s4_ok(int dt, int H, int act);
hipError_t launch_depth_conv8x8s4(const float* x, const void* w, const float* bias, void* y, int dt, int B, int H, int act, hipStream_t s);
// storage-type conversion between sub-networks (e.g. fp16 depth tokens -> bf16 cross-modal block)
hipError_t launch_convert(const void* x, int dt_in, void* y, int dt_out, size_t n, hipStream_t s);

}  // namespace hcm
