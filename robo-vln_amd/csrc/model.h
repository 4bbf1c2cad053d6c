// Host-side model state for libhcm: strict state_dict loader, device weights, workspace arena.
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include <stdint.h>

#include "../../include/hcm.h"
#include "kernels.h"

namespace hcm {

using Shape = std::vector<int64_t>;

struct HostTensor {
    std::vector<float> f;
    Shape shape;
    bool loaded = false;
};

struct ConvW {                 // [Cout][Kp] in the sub-network's storage dtype, k = (kh*KW+kw)*Cin + ci
    void* w = nullptr;
    int dt = 0;
    float* bias = nullptr;     // f32 [Cout] or null
    int Cout = 0, Cin = 0, KH = 1, KW = 1, K = 0, Kp = 0;
    int groups = 1;            // 2: hi|lo pair trunk -- w holds [2][Cout][Kp], bias [2][Cout]; Cout / Cin are per group
    // fp16 range folding (api.cpp calibrate_run): a GroupNorm-trunk conv whose un-normalised output left the fp16 guard band has the
    // power of two `fold` multiplied into its weights; the GroupNorm behind it runs with eps * fold^2 and returns exactly what it would
    // have returned for the unscaled conv.  calib_pos = position of the conv in the trunk topology (the per-position range slot).
    float fold = 1.f;
    int calib_pos = -1;
};
struct NormW { float* gamma = nullptr; float* beta = nullptr; int C = 0; };
struct LinW {                  // [N][Kp]; dt = compute dtype or f32 (recurrent weights)
    void* w = nullptr;
    float* bias = nullptr;
    int N = 0, K = 0, Kp = 0, dt = 0;
};

struct BottleneckW {
    ConvW c1, c2, c3, ds;
    ConvW c3ds;                // layer1 block 0 of a 16-bit BN-folded trunk: [W3 | Wds] K-concatenated, bias b3 + bds (bneck231_kernel, KD = 1)
    NormW n1, n2, n3, nds;     // GroupNorm variant only
    bool has_ds = false;
    int stride = 1;
};
struct TrunkW {
    bool gn = false;           // false: BN folded into conv (torchvision); true: GroupNorm (habitat)
    bool pair = false;         // hi|lo channel-concatenated pair (grouped convs, 2x GroupNorm groups)
    int groups = 0;
    ConvW conv1;               // 7x7/2 stem, K = (kh,kw,ci) order (element-wise gather path: uint8 / 16-bit frames)
    ConvW conv1_rowrun;        // same stem in the row-run K layout of the f32 RGB fast gather (torchvision trunk only)
    ConvW conv1_packed;        // same stem for the packed-frame path (16-bit trunks): k = kh*32 + kw*4 + ci, K = 224
    int k1 = 7, s1 = 2, p1 = 3, cin1 = 3;
    NormW n_conv1;
    std::vector<BottleneckW> blocks;
    ConvW compress;            // habitat: 3x3 compression conv
    NormW n_compress;
    int compress_true = 0;     // > 0: real channels (per model) of the compression conv when its output is padded to a power of two
    int out_c = 0;
};
struct SimpleCnnW {
    ConvW c0, c1, c2;          // 8x8/4 (narrow-channel gather), 4x4/2, 3x3/1
    ConvW c0_packed;           // 16-bit path: k = kh*KR + kw*cp + ci with cp = 4 (RGB, KR = 32) or 1 (depth, KR = 8)
    void* c1_frag = nullptr;   // c1.w / c2.w in MFMA-fragment order (depth, 16-bit): the three convolutions in one launch (simplecnn.hip)
    void* c2_frag = nullptr;
    LinW fc;
    int cin = 1, h = 0, w = 0, h3 = 0, w3 = 0;      // frame H x W, final map h3 x w3
};
struct BertLayerW {
    LinW qkv, o, ff1, ff2;
    void* o_frag = nullptr;    // o.w in MFMA-fragment order for the fused attention-block launch (bert_block.hip); 16-bit 768 x 768 only
    NormW ln1, ln2;
    // folded LayerNorm (fp16 BERT; forward.cpp bert(), IGemm::ln_*): ff1_f = W_ff1 diag(gamma_ln1) with ff1_s[n] = sum_k W'[n][k] (of the ROUNDED
    // weights) and ff1_t[n] = sum_k W[n][k] beta_ln1[k] + b[n]; qkv_f likewise with the PREVIOUS layer's ln2 (absent in layer 0)
    LinW qkv_f, ff1_f;
    float* qkv_s = nullptr; float* qkv_t = nullptr; float* ff1_s = nullptr; float* ff1_t = nullptr;
};
struct BertW {
    float* word = nullptr; float* pos = nullptr; float* type0 = nullptr;
    NormW ln;
    std::vector<BertLayerW> layers;
};
struct VlaLayerW {
    LinW q, kv, o, ff1, ff2; NormW ln_att, ln_ff;
    void *o_f = nullptr, *ff1_f = nullptr, *ff2_f = nullptr;      // o.w / ff1.w / ff2.w in MFMA-fragment order (16-bit, d_model 256): vla_post_wf_kernel reads its weights straight into registers
};
struct VlaW {
    LinW vis_fc, ins_fc;
    NormW ln;
    std::vector<VlaLayerW> layers;
    float* pe = nullptr;       // sinusoid table [L][d_model] f32, built once (common/utils.py:167-185)
};
struct RnnW {
    LinW cat;                  // LSTM: [4H][in+H] f32, bias = b_ih + b_hh; columns ordered like the input row (below)
    LinW ih, hh;               // GRU: separate
    int in = 0;
    // input row layout: [x[0:early) | h*mask (H) | x[early:in)].  The gate pre-activations of the first early+H columns
    // depend only on the encoders' own projections and the previous state, so that GEMM runs beside the cross-modal block
    // and only the `in - early` late columns are multiplied in the serial tail (LSTM only; GRU keeps early = in).
    int early = 0;
    int xcol(int j) const { return j < early ? j : j + (cat.w ? cat.N / 4 : 0); }
};
struct HighW {
    TrunkW rgb, depth;
    float* rgb_pe = nullptr;   // [16][64] transposed view of spatial_embeddings (the `.view` quirk)
    float* depth_pe = nullptr; // [S][64]
    int depth_S = 0, depth_C = 0;
    // hi + lo GroupNorm depth trunks fused into ONE channel-concatenated network (hcm_act only): same input frame, same
    // layer shapes, different weights -> grouped convs (2 groups), GroupNorm over 2C channels with 2G groups
    TrunkW depth_pair;
    bool has_depth_pair = false;
    TrunkW rgb_pair;           // likewise for the two BatchNorm-folded RGB ResNet-50s (shared frame, 2x64-channel stem)
    bool has_rgb_pair = false;
    // the low-level model's trunk weights are bit-identical to the high-level model's (frozen pretrained encoders in both
    // state_dicts): the trunk runs ONCE per step and feeds both models' heads -- bit-identical to running it twice
    bool rgb_shared = false, depth_shared = false;
    BertW bert;
    LinW rgb_kv, depth_kv, rgb_linear, depth_linear;
    VlaW vla;
    RnnW rnn;
    float* head_w = nullptr; float* head_b = nullptr;
};
struct LowW {
    bool rgb_simple = false, depth_simple = false;
    TrunkW rgb, depth;
    SimpleCnnW rgb_s, depth_s;
    LinW rgb_fc, depth_fc;
    float* subtask_emb = nullptr;
    RnnW rnn;
    float* lin_w = nullptr; float* lin_b = nullptr; float* stop_w = nullptr; float* stop_b = nullptr;
};

// CMANet (models/cma.py:28-186)
struct CmaW {
    TrunkW rgb, depth;
    float* rgb_pe = nullptr; float* depth_pe = nullptr;
    int depth_S = 0, depth_C = 0;
    float* emb = nullptr;          // word embedding table [vocab][E] f32
    LinW ih[2], hh[2];             // per direction: W_ih (bias = b_ih + b_hh) and W_hh, f32
    float* hh_t[2] = {nullptr, nullptr};   // W_hh transposed and gate-interleaved [H (k)][H (unit)][4 gates] for the one-launch scan
    int dirs = 1;
    LinW rgb_linear, depth_linear, rgb_kv, depth_kv;    // token-side projections (dt_vla weights, f32 outputs)
    LinW state_q, text_k, text_q, compress;             // f32
    RnnW rnn1, rnn2;
    float scale = 0.f;
    float* lin_w = nullptr; float* lin_b = nullptr; float* stop_w = nullptr; float* stop_b = nullptr;
};

struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    bool dry = true;
    void* alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        char* p = (dry ? (char*)0x1000 : base) + off;
        off += bytes;
        if (off > peak) peak = off;
        // the capacity is the maximum over the dry runs of every entry shape (api.cpp: hcm_finalize); a request beyond it is a bug in
        // that sizing -- fail the call instead of writing past the allocation
        if (!dry && off > cap) throw std::runtime_error("workspace arena overflow (" + std::to_string(off) + " > " + std::to_string(cap) + " bytes)");
        return p;
    }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
    void reset() { off = 0; }
};

struct Tap { float* dev = nullptr; size_t cap = 0; size_t n = 0; Shape shape; };

}  // namespace hcm

struct hcm_ctx {
    hcm_config cfg;
    int dt = 0;                     // headline precision: DT_F32 / DT_BF16
    // storage / MFMA input type per sub-network: in bf16 mode the GroupNorm depth trunk runs
    // on fp16 MFMA tiles (same rate, 3 more mantissa bits), everything else on bf16
    int dt_rgb = 0, dt_depth = 0, dt_bert = 0, dt_vla = 0;
    std::map<std::string, hcm::HostTensor> sd[3];       // HCM_HIGH, HCM_LOW, HCM_CMA
    int kind = 0;                   // 0: HCM hi/lo handle, 1: CMANet handle (hcm_cma_create)
    hcm_cma_config cma_cfg;
    hcm::CmaW cma;
    int* len_buf = nullptr;         // CMANet: per-sample instruction lengths
    bool finalized = false;
    int device = -1;                // the HIP device the handle was created on
    // hcm_guard_poll: the overflow-guard word travels to a pinned host word behind the caller's stream; read back without a synchronisation
    unsigned* guard_host = nullptr; hipEvent_t guard_ev = nullptr; bool guard_pending = false; unsigned guard_last = 0u;
    void* comm = nullptr; int comm_world = 1, comm_rank = 0;      // RCCL communicator of hcm_comm_init (comm.cpp)
    int gather_joined = 0;      // the LAST hcm_act_gather / hcm_gather_poison call enqueued its ncclAllGather (hcm_query(HCM_GATHER_JOINED))
    bool unusable = false;          // a re-build after calibration could not get its workspace back
    std::vector<void*> dev_allocs;
    size_t weight_bytes = 0;
    hcm::HighW hi;
    hcm::LowW lo;
    hcm::Arena arena;
    int64_t* pred_buf = nullptr;    // argmax output for hcm_act
    std::string err;
    hcm_ctx() { for (auto& f : depth_fold) f = 1.f; }
    bool taps_on = false;
    // development aid (make DEV=1, HCM_MARKS=1): wall-clock stamps of named points of a step, see Fwd::mark
    unsigned long long* marks_dev = nullptr;
    std::vector<std::string> mark_names;
    std::map<std::string, hcm::Tap> taps;
    hipStream_t stream = nullptr;   // the caller's stream of the current call
    hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr};   // side streams of the encoder chains (forward.cpp step())
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    bool concurrent = true;
    // hipGraph cache of the fused step (hcm_act): keyed by (B, dtypes, every pointer argument).  A key is run eagerly the
    // first time it is seen and captured (all forked streams included) the second time; replays cost one graph launch.
    // Round 4: the fused step is replayed as LINEAR graphs, one per chain, stitched by events outside the graphs.  hipGraphLaunch of a graph
    // captured with forked streams costs the host ~2.1 us per node (0.7-0.8 ms for the step's ~240 nodes: at B = 1 the step was bounded by
    // it and the RGB chain started 0.31 ms late); a graph captured on ONE stream goes down ROCm's batched-submission path at ~0.1 us per
    // node (tools/native/graph_launch_mt.hip).  `prog` = the step's top-level structure as recorded during capture.
    struct SegOp {                                   // kind 0 fork(n aux), 1 launch exec on st, 2 join(n aux), 3 host -> device copy on st (staged host frames)
        int kind = 0; int n = 0; hipGraphExec_t exec = nullptr; hipStream_t st = nullptr;
        void* dst = nullptr; const void* src = nullptr; size_t bytes = 0;
    };
    struct GraphEntry { std::vector<uint64_t> key; hipGraphExec_t exec = nullptr; std::vector<SegOp> prog; hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr}; };
    // Chains only overlap when their streams sit on different hardware queues (ROCm multiplexes streams onto a few; a linear graph is enqueued whole, so two
    // chains on one queue run strictly one after the other).  `pool` = spare streams; before the first segmented capture on a caller stream the side streams
    // are re-picked by a timing probe so that BERT's, the depth chain's and the caller's stream overlap pairwise (api.cpp, pick_chain_streams).
    hipStream_t pool[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<hipStream_t> probed_for;
    bool seg_mode = false, seg_open = false;        // segmented capture in progress / a chain's capture is open
    hipStream_t seg_stream = nullptr;
    std::vector<SegOp> seg_prog;
    std::vector<GraphEntry> graphs;
    std::vector<std::vector<uint64_t>> seen_keys;
    bool use_graph = true;
    int64_t graph_launches = 0, eager_launches = 0;
    bool failed = false;            // a launch failed during the current forward
    // hcm_act_ex(HCM_ACT_REUSE_INSTRUCTION): skip BERT + the instruction stream of Visual_Ling_Attn and reuse the tensors the
    // previous step left in the workspace (same batch size required); set per call
    bool reuse_instruction = false;
    // HCM_ACT_HOST_FRAMES: rgb / depth of the current hcm_act_ex call are host pointers; staged per chain into these device buffers
    bool host_frames = false;
    void* stage_rgb = nullptr; float* stage_depth = nullptr;
    int last_hi_batch = -1, last_hi_L = -1;   // shape of the cached instruction stream; -1 = none (invalidated by every other entry point)
    int cur_L = 0;                  // instruction length of the current call (<= cfg.instr_len)
    // fp16 range calibration (hcm_finalize's synthetic batch, hcm_calibrate's caller batch): while `calib` is set the forward code
    // reduces max |x| / non-finite counts of every GEMM output of the fp16 sub-networks into calib_buf[2 * slot] (0 BERT, 1 depth trunks)
    bool calib = false;
    unsigned* calib_buf = nullptr;       // device, kCalibWords words: [2 * slot] = max |x| bits, [2 * slot + 1] = non-finite count (slots 0 BERT, 1 depth, 2 RGB,
                                         // 3 cross-modal block); [kStepBadWord] = the run-time overflow guard of the recurrent cells;
                                         // [16 + 2 * pos ..] = the same pair per GroupNorm-trunk conv position (un-normalised conv outputs)
    static constexpr int kStepBadWord = 12;
    int fp16_fallback = 0;               // bit 0: BERT was re-built on bf16 tiles, bit 1: the depth trunks
    float calib_max[4] = {0.f, 0.f, 0.f, 0.f};     // last calibration's max |x| per sub-network: BERT, depth trunks, RGB trunks, cross-modal block
    unsigned calib_bad[4] = {0u, 0u, 0u, 0u};
    bool host_weights = true;            // the f32 host copies of the state_dicts are still held (needed to re-build a sub-network)
    // fp16 range folding instead of a bf16 fall-back where the network is exactly scale-invariant:
    //   depth_fold[pos]  power of two folded into the weights of GroupNorm-trunk conv `pos` (GroupNorm removes it; eps scaled to match)
    //   rgb_fold         power of two carried by EVERY activation of the BatchNorm-folded RGB trunks (ReLU / max-pool / average pools are
    //                    positively homogeneous): stem weights and all folded biases are multiplied by it, the trunk-feature columns of
    //                    the consuming projections (rgb_kv, rgb_linear, the low-level model's fc) by its inverse
    static constexpr int kDepthPos = 66;  // conv1, 16 blocks x (c1, c2, c3, down-sample), compression
    static constexpr int kCalibWords = 16 + 2 * kDepthPos;
    float depth_fold[kDepthPos];
    float rgb_fold = 1.f;
    int range_fold = 0;                  // hcm_query(HCM_RANGE_FOLD): bit 1 depth, bit 2 RGB
    const int* cur_lens = nullptr;  // optional per-environment instruction lengths of the current call (device, [B]); null = all L
};
