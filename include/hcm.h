/*
 * libhcm -- MI355X-native per-step policy forward of robo-vln's Hierarchical Cross-Modal agent.
 *
 * Plain C ABI (no torch types): pointers, sizes, an opaque handle.  Every entry point returns an
 * int status (0 = ok, negative = hcm_status) and records a message readable by hcm_last_error().
 * All I/O buffers of the forward calls are caller-owned DEVICE pointers; weights passed to
 * hcm_load_tensor are HOST pointers.  All work of a forward call is enqueued on the passed
 * hipStream_t (as void*); no host synchronisation happens inside a forward call.  A handle is not
 * thread-safe: one handle per device per thread.
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * /root/reference/robo_vln_baselines/).  The reference is pure Python with no FFI; the ctypes
 * binding a maintainer would add is shown in INTEGRATION.md.
 */
#ifndef HCM_H
#define HCM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hcm_ctx* hcm_handle;

enum hcm_status {
    HCM_OK = 0,
    HCM_ERR_ARG = -1,          /* null / out-of-range argument                         (Python: ValueError)  */
    HCM_ERR_STATE = -2,        /* call order violated (e.g. forward before finalize)   (RuntimeError)        */
    HCM_ERR_KEY = -3,          /* unknown or missing state_dict key                    (KeyError)            */
    HCM_ERR_SHAPE = -4,        /* tensor shape does not match the configured model     (ValueError)          */
    HCM_ERR_HIP = -5,          /* HIP runtime error                                    (RuntimeError)        */
    HCM_ERR_UNSUPPORTED = -6,  /* configuration the reference itself cannot run        (ValueError)          */
    HCM_ERR_NOMEM = -7
};

enum hcm_dtype { HCM_F32 = 0, HCM_BF16 = 1, HCM_I32 = 2, HCM_I64 = 3, HCM_U8 = 4, HCM_F16 = 5 };
enum hcm_model { HCM_HIGH = 0, HCM_LOW = 1, HCM_CMA = 2 /* CMANet flat baseline: hcm_cma_create handles only */ };
enum hcm_encoder { HCM_ENC_RESNET = 0, HCM_ENC_SIMPLECNN = 1 };
enum hcm_rnn { HCM_LSTM = 0, HCM_GRU = 1 };

/* hcm_query() selectors: properties the reference's callers read from the models
 * (state_encoder.num_recurrent_layers hierarchical_trainer.py:1053,:1059; MODEL.STATE_ENCODER.hidden_size). */
enum hcm_query_what {
    HCM_NUM_RECURRENT_LAYERS = 0,  /* 2 for LSTM (cat[h,c]), 1 for GRU: models/decoder/state_encoder.py:41-45 */
    HCM_HIDDEN_SIZE = 1,
    HCM_NUM_ACTIONS = 2,           /* high-level sub-task logits (4) */
    HCM_RECORD_WIDTH = 3,          /* 7 = 4 logits + (v, w) + stop logit */
    HCM_WORKSPACE_BYTES = 4,
    HCM_WEIGHT_BYTES = 5,
    HCM_MAX_BATCH = 6,
    HCM_GRAPH_LAUNCHES = 7,        /* hcm_act calls served by a captured hipGraph replay */
    HCM_EAGER_LAUNCHES = 8,
    HCM_FP16_FALLBACK = 9,         /* bit 0: BERT, bit 1: the depth trunks, bit 2: the RGB trunks, bit 3: the cross-modal block were re-built on bf16 tiles after a range calibration */
    HCM_CALIB_MAX_BERT = 10,       /* max |x| (rounded down) over the GEMM outputs of BERT / the depth trunks in the last calibration forward */
    HCM_CALIB_MAX_DEPTH = 11,
    HCM_CALIB_NONFINITE = 12,      /* non-finite values seen in the last calibration forward */
    HCM_CALIB_MAX_RGB = 13,        /* as HCM_CALIB_MAX_BERT, over the conv outputs of the RGB trunks */
    HCM_CALIB_MAX_VLA = 14,        /* ... over the cross-modal block (its GEMM outputs and the fused layer's LDS-only intermediates) */
    HCM_STEP_NONFINITE = 15,       /* overflow guard: (sample, recurrent step) pairs since hcm_finalize whose gate pre-activations were not all
                                      finite -- an fp16 overflow or a NaN anywhere upstream of the state encoder lands there, and the squashing
                                      cell would otherwise turn it into finite garbage.  0 on a healthy engine.  Synchronises the device. */
    HCM_GATHER_JOINED = 17,        /* 1 when the handle's LAST hcm_act_gather / hcm_gather_poison call enqueued its ncclAllGather, 0 when it returned in
                                      front of the collective (argument error, no communicator): a caller whose peers are about to enter the step's
                                      all-gather must then join it itself (hcm_gather_poison) -- robo-vln_amd/policy.py act(gather=True) */
    HCM_RANGE_FOLD = 16            /* bit 1: a power-of-two scale was folded into convs of the GroupNorm depth trunks, bit 2: into the RGB trunks
                                      (the exact alternative to a bf16 fall-back where the network is scale-invariant; hcm_calibrate below) */
};

/* Model hyper-parameters: the values the reference reads from MODEL.* (config/default.py:131,:156-164,
 * :180-199).  Zero-initialise, set struct_size = sizeof(hcm_config), fill. */
typedef struct hcm_config {
    int32_t struct_size;
    int32_t precision;        /* HCM_F16: fp16 storage + fp16 MFMA tiles behind the range calibration below (the measured 16-bit mode);
                                 HCM_BF16: bf16 storage + bf16 MFMA tiles in BERT (f32 residual stream) and the cross-modal block -- the sub-networks that
                                 are not scale-invariant (no range limit there); both trunk kinds stay on range-folded fp16 tiles (GroupNorm's
                                 subtraction amplifies bf16's rounding 8x: 1.9e-2 on the record from the depth trunk alone; the RGB trunks' 50
                                 bf16-rounded layers were 6e-3 of a 1e-2 tolerance -- and the exact power-of-two folds make fp16 range-safe in
                                 both by construction; round 6);
                                 HCM_F32: fp32 MFMA.  fp32 accumulation and fp32 recurrent cells / heads in every mode */
    int32_t max_batch;        /* workspace is sized for this many environments per call */
    int32_t rgb_h, rgb_w;     /* frames are NHWC; any H x W >= 32 with HCM_ENC_RESNET (adaptive pools, resnet_encoders.py:211-236), >= 36 with HCM_ENC_SIMPLECNN (simple_cnns.py:63-73) */
    int32_t depth_h, depth_w; /* HCM_ENC_RESNET: square, a multiple of 64, <= 1024 (habitat sizes the encoder from the frame height); HCM_ENC_SIMPLECNN: any H x W >= 36 */
    int32_t instr_len;        /* MAXIMUM instruction length: sizes the workspace (and the positional tables); every forward call
                                 passes its own L <= instr_len (the reference model accepts any (B or 1, L) per call and its
                                 eval loop feeds unpadded ids, common/utils.py:18-20); <= bert_max_pos (512) */
    int32_t rgb_encoder;      /* hcm_encoder; SIMPLECNN is valid for the low-level model only */
    int32_t depth_encoder;
    int32_t rgb_out, depth_out, depth_baseplanes;
    int32_t vla_layers, d_model, vla_heads, d_ff, vis_in, ins_in;
    int32_t hidden, rnn_type, num_actions, num_sub_tasks, lo_actions;
    int32_t bert_layers, bert_hidden, bert_heads, bert_inter, bert_vocab, bert_max_pos;
    int32_t build_high, build_low;      /* which of the two models this handle holds */
    int32_t use_prev_action;            /* must be 0: broken branch in the reference (seq2seq_highlevel_cma.py:203-207) */
    int32_t ablate_instruction;         /* must be 0: broken branch (:183-184) */
    int32_t progress_monitor;           /* must be 0 in forward (:221-225 references an undefined name) */
    int32_t ablate_depth;               /* working flags of both models: the encoder output is multiplied by 0 */
    int32_t ablate_rgb;                 /* (seq2seq_highlevel_cma.py:185-188, seq2seq_lowlevel.py:132-135, config/default.py:92-93) */
    int32_t reserved[8];                /* [0..3]: storage-type override (hcm_dtype + 1, 0 = default) for the depth trunk /
                                           BERT / cross-modal block / RGB trunk; see DESIGN.md section 5.
                                           [4]: keep the f32 host copies of the weights after hcm_finalize so that hcm_calibrate can
                                           re-build a sub-network (release them with hcm_release_host_weights)
                                           [5]: 1 = do not share a trunk between the two models when their trunk weights are bit-identical
                                           (the default runs it once per step and feeds both heads: same values, half the work) */
} hcm_config;

/* Replaces model construction, hierarchical_trainer.py:315-328 (Seq2Seq_HighLevel_CMA.__init__
 * models/seq2seq_highlevel_cma.py:33-141, Seq2Seq_LowLevel.__init__ models/seq2seq_lowlevel.py:32-98).
 * Rejects the flags whose branches crash in the reference. */
int hcm_create(const hcm_config* cfg, hcm_handle* out);

/* Replaces `load_state_dict(ckpt["high_level_state_dict"])` / `["low_level_state_dict"]`
 * (hierarchical_trainer.py:343-345): call once per state_dict entry with the reference's own key.
 * `data` is a HOST pointer to a contiguous tensor of `dtype` (HCM_F32 or HCM_I64); it is copied.
 * Unknown keys fail with HCM_ERR_KEY, wrong shapes with HCM_ERR_SHAPE (strict=True semantics). */
int hcm_load_tensor(hcm_handle h, int model, const char* key, const void* data, int dtype,
                    const int64_t* shape, int ndim);

/* After the last hcm_load_tensor: checks that every required key arrived, folds eval-mode BatchNorm into
 * the convolutions, re-lays weights out for NHWC implicit GEMM, converts to the compute precision,
 * uploads, and allocates the per-handle workspace for max_batch.  (The reference does the equivalent
 * implicitly in `.to(device)` + `.eval()`, hierarchical_trainer.py:339-340,:1080-1081.) */
int hcm_finalize(hcm_handle h);

/* Replaces `logits, hidden' = high_level((observations, hidden, prev_actions, masks))`
 * (hierarchical_trainer.py:1096-1097 -> models/seq2seq_highlevel_cma.py:170-233).
 *   rgb    (B,H,W,3)  rgb_dtype HCM_F32 (values 0..255, the batch_obs contract common/utils.py:78-83) or HCM_U8
 *   depth  (B,H,W,1)  f32
 *   ids    (B,L)      ids_dtype HCM_I32 / HCM_I64 / HCM_F32 (the reference carries ids as f32 and casts .long());
 *                     L is per call, 1 <= L <= cfg.instr_len: BERT runs without an attention mask and the poolers average over
 *                     all L positions (:209-210), so a padded instruction gives a different result than the unpadded one --
 *                     pass the ids exactly as the reference caller does
 *   lengths (B,) int32 device pointer or NULL.  NULL = the reference call: every row has L tokens.  The reference evaluates ONE
 *                     environment per call with its unpadded instruction; a batched rollout whose environments carry
 *                     instructions of different lengths pads the rows to a common L and passes each row's token count here:
 *                     environment b then attends over / pools over its first lengths[b] positions only and gets, bit for bit,
 *                     the result of its own unpadded (1, lengths[b]) call.  Values are clamped to [1, L].
 *   h_in   (R,B,hidden) f32, R = hcm_query(HCM_NUM_RECURRENT_LAYERS)
 *   mask   (B,) f32   -- column 0 of the reference's masks (masks[:,0], :208); 0 at episode start
 *   logits (B,num_actions) f32 out;  h_out (R,B,hidden) f32 out (may alias h_in)
 * prev_actions is ignored on the working path of the reference and is not part of this ABI. */
int hcm_high_forward(hcm_handle h, const void* rgb, int rgb_dtype, const float* depth,
                     const void* ids, int ids_dtype, const int32_t* lengths, int B, int L,
                     const float* h_in, const float* mask,
                     float* logits, float* h_out, void* stream);

/* Replaces `vel, stop, hidden' = low_level((observations, hidden, prev_actions, masks, subtask))`
 * (hierarchical_trainer.py:1099-1100 -> models/seq2seq_lowlevel.py:116-162).
 *   subtask (B,) int64 in [0, num_sub_tasks];  vel (B,2) f32;  stop (B,1) f32 (logit). */
int hcm_low_forward(hcm_handle h, const void* rgb, int rgb_dtype, const float* depth, int B,
                    const float* h_in, const float* mask, const int64_t* subtask,
                    float* vel, float* stop, float* h_out, void* stream);

/* Training / validation path: the models are called on T*N frames at once
 * (hierarchical_trainer.py:505-506,:539 and :575-576,:613) and RNNStateEncoder.forward takes the seq_forward branch
 * (models/decoder/state_encoder.py:83-133): encoders on all T*N frames, then a T-step masked recurrent scan.
 *   rgb/depth/ids/subtask: T*N rows, time-major (row t*N + n);  masks (T*N,) f32;  h_in/h_out (R,N,hidden);
 *   logits (T*N,num_actions) / vel (T*N,2) / stop (T*N,1).  T*N must not exceed max_batch.  Inference only (no autograd). */
int hcm_high_forward_seq(hcm_handle h, const void* rgb, int rgb_dtype, const float* depth, const void* ids, int ids_dtype,
                         const int32_t* lengths /* (T*N,) or NULL */, int T, int N, int L, const float* h_in, const float* masks, float* logits, float* h_out, void* stream);
int hcm_low_forward_seq(hcm_handle h, const void* rgb, int rgb_dtype, const float* depth, int T, int N,
                        const float* h_in, const float* masks, const int64_t* subtask,
                        float* vel, float* stop, float* h_out, void* stream);

/* The caller-side step of the eval loop, hierarchical_trainer.py:1095-1101: high -> argmax(dim=1) -> low.
 *   record (B,7) f32 out: [4 sub-task logits, lin_vel, ang_vel, stop logit].
 * When called repeatedly with the same pointers on a non-default stream, the step (all forked encoder streams
 * included) is captured into a hipGraph on the second call and replayed afterwards (HCM_GRAPH=0 disables). */
int hcm_act(hcm_handle h, const void* rgb, int rgb_dtype, const float* depth,
            const void* ids, int ids_dtype, const int32_t* lengths, int B, int L,
            const float* hi_h_in, const float* lo_h_in, const float* mask,
            float* record, float* hi_h_out, float* lo_h_out, void* stream);

/* hcm_act with flags.  HCM_ACT_REUSE_INSTRUCTION: the instruction ids of every environment are the same as in the previous
 * hcm_act / hcm_act_ex call on this handle (same B and L): BERT and the instruction stream of Visual_Ling_Attn are not recomputed,
 * the tensors of the previous step are reused (the reference recomputes them every step although an instruction is fixed
 * for an episode, seq2seq_highlevel_cma.py:189-195).  NOT the measured configuration: bench.py and the parity tests run with
 * flags = 0; with the flag the step executes 13.8 GFLOP per environment less.  Returns HCM_ERR_STATE if there is no
 * previous hcm_act / hcm_act_ex step with this batch size and L, or if any other forward entry point ran on the handle in between
 * (they re-use the workspace region that holds the cached tensors). */
/* HCM_ACT_HOST_FRAMES: `rgb` and `depth` point to HOST memory (page-locked: hipHostMalloc / a pinned torch tensor; what a simulator process
 * hands over) instead of device memory.  The library copies each frame tensor host -> device at the head of the encoder chain that reads
 * it (the RGB frames on the RGB trunks' stream, the depth frames on the depth trunks' stream), so the copies run beside BERT and beside
 * each other's compute instead of in front of the whole step, and they are nodes of the captured hipGraph.  `ids`, the hidden states, the mask
 * and the outputs stay device pointers.  Results are bit-identical to copying the frames first and calling without the flag.  Measured at
 * B = 64 (29 MB of uint8 RGB + f32 depth per step, two boxes): 5.20-5.26 ms against 5.28-5.30 ms with the frames copied in front of the step
 * and 4.64 ms with resident frames -- on this runtime a graph's memcpy nodes overlap its kernels only marginally (eager launches with the
 * copy engine beside them reached 4.95-5.27 ms depending on the host). */
/* HCM_ACT_CHAIN_GRAPHS (round 4): replay the step as four LINEAR hipGraphs (BERT, depth trunks, RGB trunks, tail), one per chain on the chain's own
 * stream, stitched by events, instead of ONE graph captured across the forked streams.  hipGraphLaunch enqueues a forked graph node by node at ~2.1 us per
 * node (0.5-0.8 ms of host time for the step's ~240 nodes, in capture order: at B = 1 the last chain starts 0.3 ms late), a single-stream graph at ~0.1 us per
 * node.  With the flag the host cost of a step is ~0.18 ms and the synchronous latency of a B = 1 step drops by 7-25 % (box-dependent); the GPU runs the
 * linear graphs' nodes slightly slower than the forked graph's, so PIPELINED throughput is 2-4 % lower -- a latency option for single-environment loops
 * (hierarchical_trainer.py:1088-1107 calls the policy once per simulator step), not the measured configuration.  Results are bit-identical.  The side streams
 * are picked by a one-time timing probe so that the chains' streams do not share a hardware queue.  Together with HCM_ACT_HOST_FRAMES the replay enqueues the
 * two frame copies itself, outside the graphs, at the head of their chains' streams: they then run beside BERT (B = 64, PCIe-inclusive: 4.87 -> 4.40 ms per step;
 * a gain from ~4 MB per frame tensor up -- smaller pinned copies are carried out by the calling thread behind the stream's earlier work). */
enum hcm_act_flags { HCM_ACT_REUSE_INSTRUCTION = 1, HCM_ACT_HOST_FRAMES = 2, HCM_ACT_CHAIN_GRAPHS = 4 };
int hcm_act_ex(hcm_handle h, const void* rgb, int rgb_dtype, const float* depth, const void* ids, int ids_dtype,
               const int32_t* lengths, int B, int L,
               const float* hi_h_in, const float* lo_h_in, const float* mask, float* record, float* hi_h_out, float* lo_h_out,
               int flags, void* stream);

/* Companion of HCM_ACT_REUSE_INSTRUCTION for batched rollouts in which a few environments start a new episode: recomputes
 * the cached instruction stream of the n listed environments (HOST array of indices into the batch) from ids (B,L) (device),
 * leaving the other environments' cached tensors untouched.  Needs a previous hcm_act / hcm_act_ex step at this batch size. */
int hcm_refresh_instruction(hcm_handle h, const void* ids, int ids_dtype, const int32_t* lengths, int B, int L,
                            const int32_t* env_indices, int n, void* stream);

/* ---- multi-GPU: environment-sharded replicas, ONE collective per step (no reference counterpart: the reference evaluates one
 * environment in one process, hierarchical_trainer.py:1088-1107).  One process per GPU, every rank a full weight replica stepping its own
 * B_local environments; the (B_local, 7) action records are all-gathered over RCCL / xGMI into the (world * B_local, 7) record of the whole batch,
 * rank-major = environment order.  The collective is enqueued by the library on the step's stream right behind the (hipGraph-replayed) step:
 * no Python call per step, no host synchronisation.  RCCL is resolved with dlopen when first asked for.
 *   hcm_comm_unique_id   rank 0: 128 bytes (ncclUniqueId) to hand to every rank through any side channel (torch.distributed broadcast, a file, MPI)
 *   hcm_comm_init        every rank, collectively: creates the handle's communicator (blocks until all `world` ranks have called)
 *   hcm_act_gather       hcm_act_ex + ncclAllGather(record -> gathered) on `stream`; B must be the same on every rank.  A rank whose own step
 *                        fails (bad argument, workspace, launch error) STILL takes part in the collective, with an all-NaN record, and returns
 *                        its error afterwards -- the private communicator has no watchdog, so a rank that simply returned would leave the
 *                        others blocked in ncclAllGather; they see NaN rows for that rank's environments instead
 *                        (an argument error that every rank shares -- B outside [1, max_batch], null buffers -- returns HCM_ERR_ARG
 *                        WITHOUT entering the collective: the element count must agree across ranks)
 *   hcm_gather_poison    this rank cannot even start its step (its caller failed in front of the library: a broken observation, an exception
 *                        in the host code) -- join THIS step's all-gather with an all-NaN (B, 7) block so that the peers, which are already
 *                        inside it, see NaN rows for this rank's environments and leave at the same step
 *   hcm_comm_abort       ncclCommAbort of the handle's communicator (a rank that is going to stop stepping: an exception outside the library,
 *                        shutdown after a peer's NaN record); the handle can create a new one with hcm_comm_init afterwards */
int hcm_comm_unique_id(void* out128);
int hcm_comm_init(hcm_handle h, const void* unique_id128, int rank, int world);
int hcm_comm_abort(hcm_handle h);
int hcm_gather_poison(hcm_handle h, int B, float* record, float* gathered, void* stream);
int hcm_act_gather(hcm_handle h, const void* rgb, int rgb_dtype, const float* depth, const void* ids, int ids_dtype, const int32_t* lengths,
                   int B, int L, const float* hi_h_in, const float* lo_h_in, const float* mask, float* record, float* hi_h_out, float* lo_h_out,
                   int flags, float* gathered, void* stream);

/* fp16 range safety (no reference counterpart: the reference is fp32).  Sub-networks that store fp16 (all four in HCM_F16 mode, the depth
 * trunks in HCM_BF16 mode) have a range that ends at 65504.  hcm_finalize runs one forward on a synthetic batch with range hooks on every GEMM
 * output of those sub-networks; hcm_calibrate does the same on the caller's own observations (device pointers as for hcm_act; zero recurrent
 * state; synchronises the stream).  When max |x| exceeds 2^14 or a non-finite value appears (needs the host copies of the weights:
 * hcm_config.reserved[4], else HCM_ERR_STATE):
 *   - GroupNorm depth trunks: the un-normalised output of a conv is the only tensor that can grow without bound, and GroupNorm is invariant to
 *     its scale: a power of two is folded into that conv's weights and the GroupNorm's eps is scaled to match -- exact, the trunk stays on
 *     fp16 (hcm_query(HCM_RANGE_FOLD) bit 1);
 *   - BatchNorm-folded RGB trunks: conv + bias + ReLU, pools and the residual additions are positively homogeneous, so ONE power of two is
 *     carried by every activation of the trunk (stem weights and all folded biases scaled by it) and divided out in the weights of the
 *     projections that consume the trunk features -- exact up to fp16's sub-normal floor, the trunk stays on fp16 (HCM_RANGE_FOLD bit 2);
 *   - BERT and the cross-modal block (GELU / softmax are not homogeneous), or a trunk whose overflow the fold cannot reach: re-built on bf16
 *     tiles, reported by hcm_query(HCM_FP16_FALLBACK).
 * The forward is repeated on the re-built engine, so that what sat downstream of the overflow is judged on clean inputs and the reported
 * ranges are those of the engine as it runs.  At run time NaN / inf are never washed out (ReLU, max-pool and the variance clamps propagate them
 * as torch's do) and the recurrent cells count what reaches them: hcm_query(HCM_STEP_NONFINITE). */
int hcm_calibrate(hcm_handle h, const void* rgb, int rgb_dtype, const float* depth, const void* ids, int ids_dtype, int B, int L, void* stream);
int hcm_release_host_weights(hcm_handle h);

int hcm_query(hcm_handle h, int what, int64_t* out);

/* The run-time overflow guard without a synchronisation (hcm_query(HCM_STEP_NONFINITE) waits for the device): enqueues a copy of the
 * guard word to a pinned host word behind `stream` and returns in *out the value of the last copy that has COMPLETED -- i.e. the
 * count as of an earlier poll.  Cheap enough to call every few steps from a rollout loop; a non-zero value means that some
 * environment's recurrent cell saw non-finite gate pre-activations (a broken frame, or a sub-network outside its fp16 range:
 * hcm_calibrate on real observations) and that its actions since then are not to be trusted. */
int hcm_guard_poll(hcm_handle h, void* stream, int64_t* out);

/* Message of the last failing call on this handle (or on creation when h is NULL). */
const char* hcm_last_error(hcm_handle h);

void hcm_destroy(hcm_handle h);

/* ---- CMANet flat baseline ----
 * `CMANet` (models/cma.py:19-333; constructed at robo_vln_trainer.py:326-331 with num_actions = 2) behind the same
 * handle type: hcm_cma_create -> hcm_load_tensor(h, HCM_CMA, key, ...) for every entry of the module's state_dict
 * (strict) -> hcm_finalize -> hcm_cma_forward ...; hcm_query / hcm_last_error / hcm_destroy work as for the HCM handles. */
typedef struct hcm_cma_config {
    int32_t struct_size;
    int32_t precision;            /* HCM_F16 / HCM_BF16 (16-bit trunks as in hcm_config.precision; fp32 text / recurrent / attention side) or HCM_F32 */
    int32_t max_batch;
    int32_t rgb_h, rgb_w, depth_h, depth_w;
    int32_t instr_len;            /* MAXIMUM padded token count per instruction (<= 256); every forward passes its own L <= instr_len */
    int32_t vocab_size, embedding_size, instr_hidden, bidirectional;   /* MODEL.INSTRUCTION_ENCODER.* (default.py:97-115) */
    int32_t rgb_out, depth_out, depth_baseplanes;
    int32_t hidden, rnn_type;     /* MODEL.STATE_ENCODER.* for both state encoders */
    int32_t num_actions;          /* 2 */
    int32_t use_prev_action;      /* must be 0 (MODEL.CMA.use_prev_action default) */
    int32_t rcm_state_encoder;    /* must be 0 (MODEL.CMA.rcm_state_encoder default) */
    int32_t progress_monitor;     /* must be 0: the auxiliary loss is a training-only branch (cma.py:320-329) */
    /* round 6 (carved out of the reserved words: zero = the defaults, struct size unchanged) */
    int32_t instr_rnn;            /* MODEL.INSTRUCTION_ENCODER.rnn_type: HCM_LSTM (0, default.py:111) or HCM_GRU (instruction_encoder.py:42) */
    int32_t ablate_instruction;   /* cma.py:236-241: `embedding * 0` behind the encoder -- the encoder is then not run, everything downstream sees the zeros */
    int32_t ablate_depth;         /*   (text attention over an all-zero instruction: every position masked, uniform weights, text = 0 exactly) */
    int32_t ablate_rgb;
    int32_t reserved[4];
} hcm_cma_config;

int hcm_cma_create(const hcm_cma_config* cfg, hcm_handle* out);

/* Replaces `output, stop_out, rnn_hidden_states = actor_critic((observations, rnn_hidden_states, prev_actions, masks))`
 * (robo_vln_trainer.py:1096 -> models/cma.py:211-333).
 *   rgb (B,H,W,3) HCM_F32 / HCM_U8; depth (B,H,W,1) f32; ids (B,L) HCM_I32 / HCM_I64 / HCM_F32, 0 = padding
 *   h_in (R,B,hidden) f32 with R = hcm_query(HCM_NUM_RECURRENT_LAYERS) = both state encoders (4 for LSTM, 2 for GRU)
 *   mask (B,) f32 (column 0 of the reference's masks, cma.py:219)
 *   out (B,num_actions), stop (B,1), h_out (R,B,hidden): f32 outputs.  The reference writes the new hidden state into
 *   the tensor it was given and returns it; here h_out may alias h_in to get the same effect. */
int hcm_cma_forward(hcm_handle h, const void* rgb, int rgb_dtype, const float* depth, const void* ids, int ids_dtype, int B, int L,
                    const float* h_in, const float* mask, float* out, float* stop, float* h_out, void* stream);

/* ---- test / profiling hooks (not part of the drop-in surface) ---- */

/* Enable capture of named intermediate activations during the next forward calls. */
int hcm_debug_enable_taps(hcm_handle h, int enable);
/* Copy a captured intermediate (as f32) to host; synchronises the device.  *n_out = element count;
 * shape_out receives up to 4 dims (0-padded). */
int hcm_debug_get_tap(hcm_handle h, const char* name, float* host_out, int64_t capacity,
                      int64_t* n_out, int64_t* shape_out);
/* Development aid: cycle totals of the implicit-GEMM K-loop phases, collected only when the process runs with
 * HCM_IGEMM_PROF=1 (which selects instrumented builds of the 8-wave bf16 kernels).  out8: prologue, DMA issue,
 * fragment reads + MFMA issue, DMA wait, barrier, epilogue (cycles summed over waves), waves, K iterations. */
int hcm_debug_igemm_prof(uint64_t* out8, int reset);
/* Development aid: phase cycle totals of the profiled builds of the 256 x 256 GEMM kernels (`make DEV=1` library,
 * hcm_op_linear_impl variants 13-15): out1024 = [16 workgroups][8 waves][8 counters], see csrc/gemm256.hip. */
int hcm_debug_gemm256_prof(uint64_t* out1024, int reset);
/* Development aid (`make DEV=1` library created under HCM_MARKS=1; otherwise returns 0): wall-clock stamps (100 MHz ticks) that one-lane marker
 * kernels wrote at named points of the last step's chains -- how the three encoder chains really interleave inside a hipGraph-replayed step, which
 * a kernel trace cannot show (rocprofv3 serialises the streams).  Returns the number of marks; names = '\n'-joined, in slot order. */
int hcm_debug_marks(hcm_handle h, uint64_t* out256, char* names, int names_cap);

/* Stand-alone operator entry points used by the kernel-level parity tests (device pointers, f32 or bf16
 * per `dtype`; layouts NHWC / row-major); implemented at the end of robo-vln_amd/csrc/api.cpp. */
int hcm_op_conv2d(const void* x, const void* w_ohwi, const float* bias, const void* residual, void* y,
                  int dtype, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                  int act, void* stream);
/* conv2d (no bias) + GroupNorm(groups) (+ residual) (+ ReLU) in one launch; the output map must have Ho*Wo | 64 pixels and
 * Cout / groups must be a multiple of 8 dividing 128 (the GN-ResNet layers at 8x8 and 4x4). */
int hcm_op_conv2d_gn(const void* x, const void* w_ohwi, const float* gamma, const float* beta, const void* residual, void* y,
                     int dtype, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int groups, float eps,
                     int relu, void* stream);
/* Tail of a BatchNorm-folded ResNet bottleneck in ONE launch (16-bit dtypes, C1 = 64 or 128):
 *   y = relu(conv1x1(relu(conv3x3(x, w2, stride, pad 1) + b2), w3) + b3 + identity)
 * x (B,H,W,C1), w2 [C1][3][3][C1], w3 [4*C1][C1], identity and y (B,Ho,Wo,4*C1).  Bit-identical to hcm_op_conv2d applied
 * twice (reference: torchvision Bottleneck.forward conv2/bn2/relu/conv3/bn3/+identity/relu as used by
 * resnet_encoders.py:196-225 TorchVisionResNet50). */
int hcm_op_bottleneck_tail(const void* x, const void* w2, const float* b2, const void* w3, const float* b3, const void* identity,
                           void* y, int dtype, int B, int H, int W, int C1, int stride, void* stream);
/* hcm_op_bottleneck_tail plus the NEXT block's 1x1 reduction from the output tile, still one launch:
 *   o1 = relu(conv1x1(y, w1) + b1),  w1 [CN][4*C1], CN = 64 or 128 (128 only when C1 = 128 ... or C1 = 64), o1 (B,Ho,Wo,CN).
 * Bit-identical to hcm_op_conv2d applied three times. */
int hcm_op_bottleneck_tail_next(const void* x, const void* w2, const float* b2, const void* w3, const float* b3, const void* identity,
                                void* y, const void* w1, const float* b1, void* o1, int dtype, int B, int H, int W, int C1, int stride,
                                int CN, void* stream);
/* A ResNet stage's FIRST bottleneck (C1 = 64 mid channels, 64-channel block input xd): as hcm_op_bottleneck_tail_next with CN = 64,
 * but the identity is the block's own 1x1 down-sample conv, folded into the expansion GEMM: w3ds [256][128] = [W3 | Wds],
 * b3ds = b3 + bds;  y = relu(conv1x1(t, W3) + conv1x1_stride(xd, Wds) + b3ds),  t = relu(conv3x3_stride(x, w2) + b2). */
int hcm_op_bottleneck_tail_ds(const void* x, const void* w2, const float* b2, const void* w3ds, const float* b3ds, const void* xd,
                              void* y, const void* w1, const float* b1, void* o1, int dtype, int B, int H, int W, int stride,
                              void* stream);
/* conv2d (no bias) + GroupNorm(groups) (+ residual) (+ ReLU) for LARGE maps (Ho*Wo a multiple of 64 and >= 256, 16-bit dtypes): the
 * conv's epilogue emits the GroupNorm partial sums from its f32 tile image, one more launch normalises in place (the GN-ResNet
 * layers at 64x64 .. 16x16; resnet_encoders.py:37-101 / habitat resnet.py GroupNorm(ngroups)). */
int hcm_op_conv2d_gn_large(const void* x, const void* w_ohwi, const float* gamma, const float* beta, const void* residual, void* y,
                           int dtype, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int groups, float eps,
                           int relu, void* stream);
/* first-layer (Cin = 1 or 3) convolution gathering straight from the raw frame x (x_dtype HCM_F32 / HCM_U8 / dtype):
 * w is [Cout][Kp] with k = (kh*KW+kw)*C + ci (rowrun = 0) or k = kh*24 + kw*3 + ci (rowrun = 1, f32 RGB frames only). */
int hcm_op_stem_conv(const void* x, int x_dtype, const void* w, const float* bias, void* y, int dtype, int B, int H, int W, int C,
                     int Cout, int KH, int KW, int stride, int pad, int K, int Kp, int rowrun, float scale, int act, void* stream);
/* 7x7 stride-2 pad-3 RGB stem on a 16-bit trunk through the packed-frame path: x is the raw (B,H,W,3) frame (HCM_F32 or
 * HCM_U8, H and W even), w is [Cout][224] with k = kh*32 + kw*4 + ci, scratch holds hcm_op_stem_scratch_bytes(B,H,W). */
int hcm_op_stem_conv_packed(const void* x, int x_dtype, const void* w, const float* bias, void* y, int dtype, int B, int H, int W,
                            int Cout, float scale, int act, void* scratch, void* stream);
/* The same stem followed by ReLU and MaxPool2d(3, 2, 1) (torchvision resnet50 conv1/bn1/relu/maxpool), with the horizontal half of
 * the pool fused into the conv's epilogue and a vertical-only pool kernel after it: y is (B, H/4, W/4, Cout), bit-identical to
 * hcm_op_stem_conv_packed(act = ReLU) + hcm_op_maxpool3x3s2.  W/2 must be a power of two <= 128, Cout a multiple of 64; half_map holds
 * B * (H/2) * (W/4) * Cout elements of `dtype`. */
int hcm_op_stem_conv_packed_pool(const void* x, int x_dtype, const void* w, const float* bias, void* y, int dtype, int B, int H, int W,
                                 int Cout, float scale, void* scratch, void* half_map, void* stream);
/* Round 6: the same three modules as ONE launch behind the frame packing (csrc/stem.hip: weights in registers, the packed frame streamed through
 * an LDS ring, both pool halves in LDS).  W == 256, H % 4 == 0, Cout % 64 == 0, 16-bit dtypes; bit-identical to hcm_op_stem_conv_packed_pool. */
int hcm_op_stem_pool_fused(const void* x, int x_dtype, const void* w, const float* bias, void* y, int dtype, int B, int H, int W,
                           int Cout, float scale, void* scratch, void* stream);
/* ... plus layer1 block 0's 1x1 reduction of the pooled map in the same launch: o1 (B, H/4, W/4, Cout) = ReLU(w1 x pooled + b1) per 64-channel group
 * (w1 [Cout][64], group g = rows 64 g .. over the group's own 64 channels); bit-identical to hcm_op_conv2d (1x1, ReLU) applied to each 64-channel group. */
int hcm_op_stem_pool_fused_red(const void* x, int x_dtype, const void* w, const float* bias, void* y, int dtype, int B, int H, int W,
                               int Cout, float scale, void* scratch, const void* w1, const float* b1, void* o1, void* stream);
int64_t hcm_op_stem_scratch_bytes(int B, int H, int W);
/* SimpleDepthCNN's first layer, Conv2d(1, 32, 8, stride 4) (+ bias, activation) straight from a raw f32 depth frame (B,H,H,1) in one pass
 * (csrc/simplecnn.hip; H a multiple of 4, <= 1024); w is the OHWI weight [32][64] in `dtype` (16-bit), scratch holds B*H*H + 64 elements of
 * `dtype` (used by the convert + implicit-GEMM route only, HCM_NO_DEPTH_CONV0=1) (simple_cnns.py:76-84). */
int hcm_op_depth_conv8x8s4(const float* depth, const void* w, const float* bias, void* y, int dtype, int B, int H, int act, void* scratch,
                           void* stream);
/* SimpleDepthCNN's three convolutions (models/encoders/simple_cnns.py:76-100; depth branch :104-125) in ONE launch (csrc/simplecnn.hip): depth (B,H,H,1)
 * f32 -> y (B,h3,h3,32), the map the Linear reads; the 63 x 63 x 32 and 30 x 30 x 64 maps of a 256-pixel frame never leave the chip.  w0 (32,64) as
 * hcm_op_depth_conv8x8s4 takes it; w1_frag / w2_frag: the 4x4/2 and 3x3/1 weights as (64,512) / (32,576) rows with k = (kh*KW + kw)*Cin + ci (the OHWI
 * layout of hcm_op_conv2d), passed through hcm_op_pack_frag.  16-bit dtypes, H % 4 == 0, H <= 256.  Bit-identical to hcm_op_depth_conv8x8s4 +
 * hcm_op_conv2d (ReLU) + hcm_op_conv2d. */
int hcm_op_simplecnn3(const float* depth, const void* w0, const float* b0, const void* w1_frag, const float* b1, const void* w2_frag, const float* b2,
                      void* y, int dtype, int B, int H, void* stream);
int hcm_op_linear(const void* x, const void* w, const float* bias, const void* residual, void* y,
                  int dtype, int M, int N, int K, int act, int out_f32, void* stream);
/* One cross-modal layer after the projections for `streams` (1 or 2) visual streams in one launch (csrc/vla_fused.hip;
 * InterModuleAttnLayer.forward, models/transformer/transformer.py:209-221): [kv != NULL: softmax(q k^T / 8) v over Lk[s] <= 32 keys, kv[s] =
 * (B, Lk[s], 512) = fc_k | fc_v; else att[s] (B, L, 256) is the attention output] -> LayerNorm(I + . Wo^T + bo) -> LayerNorm(x1 + relu(x1 W1^T + b1)
 * W2^T + b2) -> out[s] (B, L, 256); pooled[s] (optional, L <= 80): mean over each environment's first lens[b] (or L) tokens -> pooled[s][b * ld_pool + c].
 * 16-bit dtypes; kv / att / out / pooled are HOST arrays of `streams` device pointers. */
int hcm_op_vla_layer(const void* q, const void* I, const void* const* kv, const int* Lk, const void* const* att, void* const* out, float* const* pooled,
                     int ld_pool, const void* wo, const float* bo, const void* w1, const float* b1, const void* w2, const float* b2, const float* g1,
                     const float* be1, const float* g2, const float* be2, const int32_t* lens, int dtype, int B, int L, int d_ff, int streams,
                     void* stream);
/* The same layer with its three weight matrices in MFMA-fragment order (hcm_op_pack_frag of the (256, 256), (d_ff, 256) and (256, d_ff) K-contiguous
 * weights): every wave reads its operand fragments straight from L2 into registers, no weight staging in LDS and no barrier per K tile (round 6;
 * the form the engine's own step uses).  Bit-identical to hcm_op_vla_layer. */
int hcm_op_vla_layer_frag(const void* q, const void* I, const void* const* kv, const int* Lk, const void* const* att, void* const* out,
                          float* const* pooled, int ld_pool, const void* wo_frag, const float* bo, const void* w1_frag, const float* b1,
                          const void* w2_frag, const float* b2, const float* g1, const float* be1, const float* g2, const float* be2,
                          const int32_t* lens, int dtype, int B, int L, int d_ff, int streams, void* stream);
/* hcm_op_linear with the kernel family chosen by the caller: impl 0 = the library's choice, 1 = the 128-wide implicit-GEMM kernels,
 * 2 = the 256 x 256-tile 8-phase kernel (csrc/gemm256.hip; HCM_ERR_ARG when the shape does not qualify), 3 = the few-row kernel (csrc/skinny.hip:
 * a wave per 16 x 16 output tile, operands straight into registers; any row count here, the library's own choice takes it up to 160 rows where its
 * cost model says so).  All three must agree bit for bit.
 * (impl values >= 16 select timing-experiment builds of the 256-wide kernel that exist in `make DEV=1` libraries only: HCM_ERR_HIP otherwise.) */
int hcm_op_linear_impl(const void* x, const void* w, const float* bias, const void* residual, void* y,
                       int dtype, int M, int N, int K, int act, int out_f32, int impl, void* stream);
int hcm_op_attention(const void* q, const void* k, const void* v, void* out, int dtype,
                     int B, int heads, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo, void* stream);
int hcm_op_layernorm(const void* x, const void* residual, const float* gamma, const float* beta,
                     void* y, int dtype, int rows, int D, float eps, void* stream);
/* BERT attention block tail in ONE launch (csrc/bert_block.hip; BertSelfAttention + BertSelfOutput of the BertModel call,
 * seq2seq_highlevel_cma.py:192-195): y = LayerNorm(softmax(Q K^T / 8) V Wo^T + bo + residual) over B samples of L <= 96 rows, 12 heads of 64;
 * qkv (B*L, 2304) = Q | K | V, lengths (B,) int32 keys per sample or NULL; wo_frag = the (768, 768) weight in MFMA-fragment order as
 * hcm_op_pack_frag writes it (N % 16 == 0, K % 32 == 0: the 16-byte chunk W[ct*16 + fr][ks*32 + fg*8 ..] at chunk index (ks*(N/16) + ct)*64 + fg*16 + fr:
 * a wave's operand fragment is 1 KB contiguous -- hcm_finalize keeps this second copy of every BERT layer's output projection).  Bit-identical to hcm_op_attention + hcm_op_linear
 * (+ residual) + hcm_op_layernorm.  residual32 / y32 non-NULL: the residual stream and the sum stay f32 (the "bf16" mode's BERT), y is the 16-bit
 * operand copy of the LayerNorm output and y32 the stream; y may alias residual, y32 may alias residual32. */
int hcm_op_pack_frag(const void* w, void* out, int dtype, int N, int K, void* stream);
int hcm_op_bert_attn_block(const void* qkv, const void* wo_frag, const float* bo, const void* residual, const float* residual32, const float* gamma,
                           const float* beta, void* y, float* y32, int dtype, int B, int L, const int32_t* lengths, float eps, void* stream);
/* LayerNorm followed by the addition of a row table: y[r] = LN(x[r] (+ residual[r])) + post[r % post_rows]  (the positional
 * encoding of Visual_Ling_Attn, transformer.py:265-269) */
int hcm_op_layernorm_post(const void* x, const void* residual, const float* gamma, const float* beta, const float* post, int post_rows,
                          void* y, int dtype, int rows, int D, float eps, void* stream);
int hcm_op_groupnorm(void* x_inplace, const void* residual, const float* gamma, const float* beta,
                     int dtype, int B, int HW, int C, int groups, float eps, int relu, void* stream);
/* conv (bias-free, statistics from its epilogue) -> MaxPool2d(3, 2, 1) over relu(GroupNorm(conv)) with the normalisation applied on load by the pool
 * (the depth stem of the GroupNorm trunk as the step runs it since round 5; replaces habitat's ResNet stem conv1 = Sequential(conv, GroupNorm, ReLU) +
 * maxpool as used at resnet_encoders.py:27-33).  y is [B][Hp][Wp][Cout]; 16-bit types, the conv's map a multiple of 64 pixels per sample. */
int hcm_op_conv2d_gn_pool(const void* x, const void* w_ohwi, const float* gamma, const float* beta, void* y, int dtype, int B, int H, int W, int Cin,
                          int Cout, int KH, int KW, int stride, int pad, int groups, float eps, void* stream);
/* y = relu?(GN(conv1x1(x, w)) + round(GN2(conv1x1_stride2(x2, w2)))) normalised in ONE pass over both un-normalised maps: the end of a stage-first
 * bottleneck of the GroupNorm trunk, `out = relu(bn3(conv3(.)) + downsample(x))`; x is [B][H][W][Cin], x2 [B][H*stride2][W*stride2][Cin2]. */
int hcm_op_conv2d_gn_res2(const void* x, const void* w_ohwi, const float* gamma, const float* beta, const void* x2, const void* w2_ohwi, const float* gamma2,
                          const float* beta2, void* y, int dtype, int B, int H, int W, int Cin, int Cin2, int stride2, int Cout, int groups, float eps,
                          int relu, void* stream);
int hcm_op_maxpool3x3s2(const void* x, void* y, int dtype, int B, int H, int W, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HCM_H */
