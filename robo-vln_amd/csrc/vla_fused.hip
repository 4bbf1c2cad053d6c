// One launch per cross-modal layer for BOTH Visual_Ling_Attn calls (rgb, depth): everything of InterModuleAttnLayer.forward
// (models/transformer/transformer.py:209-221) that follows the query / key / value projections --
//
//   att  = softmax(Q K^T / sqrt(64)) V                      ScaledDotProductAttention  (:81-109)   [in-kernel when Lk <= 32: layer 0,
//                                                                                                   whose keys are the 16 visual tokens]
//   x1   = LayerNorm(I + att Wo^T + bo)                     MultiHeadAttention         (:111-126)  (the residual is the query stream I)
//   out  = LayerNorm(x1 + relu(x1 W1^T + b1) W2^T + b2)     PositionWiseFeedForward    (:25-43)
//   pool = mean over the instruction's tokens of out        cross_pooler, seq2seq_highlevel_cma.py:209-210 (last layer only)
//
// for a block of up to 80 instruction tokens of one (environment, visual stream) per workgroup.  The activations of the block never
// leave the CU between these steps: attention output, x1 and the 256-wide slices of the 1024-wide FFN intermediate live in LDS as
// 16-bit MFMA operands (rows padded to 528 B: conflict-free fragment reads), the two LayerNorms reduce over the 8 waves through a
// small LDS table, the FFN runs in four 256-column slices (slice c of fc1 feeds k-range c of fc2, accumulated in registers).  Weights
// are the MFMA A operand: the layer's 1.15 MB travel L2 -> LDS as ONE stream of 36 K tiles (256 output rows x 64 k = 32 KB each: fc_o,
// then per slice fc1[c] and fc2[:, c]) by `buffer_load ... lds` into a 2 x 32 KB ring, a tile ahead of its use across the GEMM
// boundaries (XOR-swizzled 128-B rows as in igemm.hip); a lane's accumulator holds 4 consecutive channels of one token, as everywhere
// in this library.  (First version: weight fragments as plain global loads of 16 rows x 64 B per instruction -- 80 us per launch at
// B = 64, bound by the texture-address path; staged through LDS in full 128-B rows, with the
// in-kernel attention on MFMA, the same launch takes 42 us: ~19 of them are the weight stream -- every workgroup needs all 1.15 MB, and a ring
// that is one 32 KB tile ahead is a latency chain at ~62 GB/s per CU (the same CU draws 125 GB/s from L2 with more in flight) -- 3.6 the attention, ~13 the LayerNorm / epilogue / store phases, 6.5 launch + operand staging.)
//
// Against the seven launches it replaces (attention, fc_o, LayerNorm, fc1, fc2, LayerNorm, mean) the GEMMs use the same MFMA
// instruction over the same k order on the same rounded operands; the LayerNorm / mean reductions have a different (fixed) order, so
// results agree with the unfused path to fp32 round-off of those reductions (tests/test_fusion_toggles_gpu.py), not bit for bit.
#include <cstdlib>
#include "kernels.h"
#include "dev.h"

namespace hcm {

typedef float v_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 v_bf16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct VMma;
template <> struct VMma<bf16> {
    static __device__ __forceinline__ void run(v_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v_bf16x8, a), __builtin_bit_cast(v_bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct VMma<f16> {
    static __device__ __forceinline__ void run(v_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), acc, 0, 0, 0);
    }
};

#ifdef HCM_DEV_KNOBS
#define V_DBG(p) ((p).dbg)
#else
#define V_DBG(p) 0
#endif
constexpr int V_D = 256, V_RB = 80, V_MF = 5, V_LDA = 528;          // model width, rows per workgroup, 16-row fragments, LDS row bytes
constexpr int V_KVMAX = 32;                                         // keys of the in-kernel attention
constexpr int V_WT = 32768;                                         // one weight K tile: 256 rows x 128 B
constexpr size_t V_LDS = (size_t)2 * V_RB * V_LDA + (size_t)8 * V_RB * 2 * sizeof(float) + (size_t)2 * V_WT;      // K|V overlays the x1 buffer
static_assert((size_t)4 * 32 * 128 + (size_t)4 * 64 * 36 * 2 <= (size_t)V_RB * V_LDA, "K and V^T staging must fit the x1 buffer");
typedef int v_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void v_dma16(unsigned lds_addr, unsigned voff, v_v4i rsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ v_v4i v_make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    v_v4i r;
    r[0] = (int)(unsigned)a; r[1] = (int)((unsigned)(a >> 32) & 0xFFFFu); r[2] = (int)bytes; r[3] = 0x00020000;
    return r;
}

// workgroup barrier for the LDS activation buffers only (vla_post_wf_kernel): __syncthreads() would also drain the weight requests in flight
__device__ __forceinline__ void v_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// (mx: running max |v| of everything this thread rounds to the storage type -- the fp16 range check of the calibration forward; a NaN counts
//  as +inf.  The x1 rows and the feed-forward intermediate never leave LDS, so no hook outside the kernel can see them.)
template <typename T> __device__ __forceinline__ void v_st4(char* p, const float (&v)[4], float& mx) {
    T o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        Tr<T>::st(&o[e], v[e]);
        const float a = fabsf(v[e]);
        mx = (a != a) ? __builtin_inff() : fmaxf(mx, a);
    }
    *reinterpret_cast<uint2*>(p) = *reinterpret_cast<const uint2*>(o);
}
template <typename T> __device__ __forceinline__ void v_ld4(const char* p, float (&v)[4]) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    T o[4];
    *reinterpret_cast<uint2*>(o) = u;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = Tr<T>::ld(&o[e]);
}

// One weight K tile (in LDS, swizzled 128-B rows) x the matching 64 columns of the activation block: 2 k-steps of 32
template <typename T>
__device__ __forceinline__ void v_mma_tile(v_f32x4 (&acc)[2][V_MF], const char* sw, const char* sAct, int kt, int nb, int fr, int fg) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        uint4 wf[2], xf[V_MF];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = nb + i * 16 + fr;
            wf[i] = *reinterpret_cast<const uint4*>(sw + r * 128 + (((ks * 4 + fg) ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < V_MF; ++j) xf[j] = *reinterpret_cast<const uint4*>(sAct + (j * 16 + fr) * V_LDA + (kt * 64 + ks * 32 + fg * 8) * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < V_MF; ++j) VMma<T>::run(acc[i][j], wf[i], xf[j]);
    }
}

// LayerNorm over the 256 channels of every row of the block: a lane holds v[i][j][e] = channel wave*32 + i*16 + fg*4 + e of row j*16 + fr.
// Per row: sums over the lane's 8 values -> the 4 lane groups (xor 16, 32) -> the 8 waves through sRed; fixed order, no atomics.
template <typename T, bool NB = false>
__device__ __forceinline__ void v_layernorm(float (&v)[2][V_MF][4], const float4 (&gg)[2], const float4 (&bb)[2], float* sRed, int wave, int fr, int fg) {
#pragma unroll
    for (int j = 0; j < V_MF; ++j) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { a += v[i][j][e]; q += v[i][j][e] * v[i][j][e]; }
        a += __shfl_xor(a, 16, 64); q += __shfl_xor(q, 16, 64);
        a += __shfl_xor(a, 32, 64); q += __shfl_xor(q, 32, 64);
        if (fg == 0) { sRed[(wave * V_RB + j * 16 + fr) * 2] = a; sRed[(wave * V_RB + j * 16 + fr) * 2 + 1] = q; }
    }
    if constexpr (NB) v_lds_barrier(); else __syncthreads();
    float g4[2][4], b4[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        g4[i][0] = gg[i].x; g4[i][1] = gg[i].y; g4[i][2] = gg[i].z; g4[i][3] = gg[i].w;
        b4[i][0] = bb[i].x; b4[i][1] = bb[i].y; b4[i][2] = bb[i].z; b4[i][3] = bb[i].w;
    }
#pragma unroll
    for (int j = 0; j < V_MF; ++j) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { a += sRed[(w * V_RB + j * 16 + fr) * 2]; q += sRed[(w * V_RB + j * 16 + fr) * 2 + 1]; }
        const float mean = a * (1.0f / V_D);
        const float rstd = rsqrtf(relu_f(q * (1.0f / V_D) - mean * mean) + 1e-5f);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][j][e] = (v[i][j][e] - mean) * rstd * g4[i][e] + b4[i][e];
    }
    if constexpr (NB) v_lds_barrier(); else __syncthreads();                 // sRed may be written again
}

template <typename T> __device__ __forceinline__ uint32_t v_pack2(float a, float b) {
    T t[2];
    Tr<T>::st(&t[0], a);
    Tr<T>::st(&t[1], b);
    return (uint32_t)t[0].v | ((uint32_t)t[1].v << 16);
}

// softmax(Q K^T / 8) V for the block's 80 rows over Lk <= 32 keys, 4 heads of 64, on MFMA -- the scheme of attention_mfma_kernel
// (attention.hip): transposed scores S^T = K Q^T (A operand = K rows from LDS, B operand = Q rows from global), so a lane holds 4 keys of
// ONE query per 16-key tile and the softmax reduces in registers plus two cross-lane steps; the exponentiated scores are already in
// A-operand position for P.V (V staged transposed).  Units of (head, 16-query tile) are dealt round-robin to the 8 waves; the output goes
// to the LDS operand image sA instead of global memory.  sK: [4 heads][32 keys][64] T, 128-B rows with the chunk index XOR (row & 7);
// sVt: [4 heads][64][36] T.  Rows / keys beyond nrow / Lk: zeros in, nothing out of range read.
template <typename T>
__device__ __forceinline__ void v_attention_mfma(const T* __restrict__ q, const T* __restrict__ kv, int Lk, int nrow, char* sK, T* sVt, char* sA, int tid) {
    constexpr int VS = 36;
    // stage K (swizzled rows) and V^T for all four heads; keys >= Lk are zero
    for (int e = tid; e < 4 * 32 * 8; e += 512) {
        const int c = e & 7, row = (e >> 3) & 31, h = e >> 8;
        uint4 kk = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (row < Lk) {
            kk = *reinterpret_cast<const uint4*>(kv + (size_t)row * 512 + h * 64 + c * 8);
            vv = *reinterpret_cast<const uint4*>(kv + (size_t)row * 512 + 256 + h * 64 + c * 8);
        }
        *reinterpret_cast<uint4*>(sK + (h * 32 + row) * 128 + ((c ^ (row & 7)) << 4)) = kk;
        const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            T t;
            t.v = (uint16_t)(w[j >> 1] >> ((j & 1) * 16));
            sVt[(size_t)(h * 64 + c * 8 + j) * VS + row] = t;
        }
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    for (int unit = wave; unit < 4 * V_MF; unit += 8) {
        const int h = unit / V_MF, qt = unit - h * V_MF;
        int qrow = qt * 16 + fr;
        if (qrow >= nrow) qrow = nrow - 1;                       // clamped: garbage for rows that are zeroed below
        const T* qp = q + (size_t)qrow * V_D + h * 64;
        const uint4 q0 = *reinterpret_cast<const uint4*>(qp + fg * 8);
        const uint4 q1 = *reinterpret_cast<const uint4*>(qp + 32 + fg * 8);
        v_f32x4 sc[2];
        float mx = -3.0e38f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r = t * 16 + fr;
            const char* kr = sK + (h * 32 + r) * 128;
            const uint4 k0 = *reinterpret_cast<const uint4*>(kr + (((0 + fg) ^ (r & 7)) << 4));
            const uint4 k1 = *reinterpret_cast<const uint4*>(kr + (((4 + fg) ^ (r & 7)) << 4));
            sc[t] = (v_f32x4){0.f, 0.f, 0.f, 0.f};
            VMma<T>::run(sc[t], k0, q0);
            VMma<T>::run(sc[t], k1, q1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = t * 16 + fg * 4 + e;
                const float sv = key < Lk ? sc[t][e] * 0.125f : -3.0e38f;          // 1 / sqrt(64); padded keys masked
                sc[t][e] = sv;
                mx = fmaxf(mx, sv);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float pv = __expf(sc[t][e] - mx); sc[t][e] = pv; sum += pv; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        uint4 pa;
        pa.x = v_pack2<T>(sc[0][0], sc[0][1]); pa.y = v_pack2<T>(sc[0][2], sc[0][3]);
        pa.z = v_pack2<T>(sc[1][0], sc[1][1]); pa.w = v_pack2<T>(sc[1][2], sc[1][3]);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int dtile = 0; dtile < 4; ++dtile) {
            const T* vr = sVt + (size_t)(h * 64 + dtile * 16 + fr) * VS + fg * 4;
            const uint2 lo = *reinterpret_cast<const uint2*>(vr);
            const uint2 hi = *reinterpret_cast<const uint2*>(vr + 16);
            v_f32x4 o = (v_f32x4){0.f, 0.f, 0.f, 0.f};
            VMma<T>::run(o, pa, make_uint4(lo.x, lo.y, hi.x, hi.y));
            // o[e] belongs to query qt*16 + fg*4 + e, column dtile*16 + fr; its normaliser lives in lane (fg*4 + e)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float inv_q = __shfl(inv, fg * 4 + e, 64);
                const int row = qt * 16 + fg * 4 + e;
                T ov;
                Tr<T>::st(&ov, row < nrow ? o[e] * inv_q : 0.f);
                *reinterpret_cast<T*>(sA + row * V_LDA + (h * 64 + dtile * 16 + fr) * 2) = ov;
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(512) void vla_post_kernel(VlaPost p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                   // attention output, later the FFN intermediate slice, last the output image
    char* sX = smem + V_RB * V_LDA;                    // x1 = LayerNorm(I + att Wo^T)
    char* sKV = sX;                                    // K | V of the in-kernel attention: [Lk][512] T (dead before x1 is written)
    float* sRed = reinterpret_cast<float*>(sX + V_RB * V_LDA);
    char* sW = reinterpret_cast<char*>(sRed) + 8 * V_RB * 2 * sizeof(float);      // weight K-tile ring: 2 x 32 KB
    const int st = blockIdx.y;
    const int nblk = (p.L + V_RB - 1) / V_RB;
    const int b = blockIdx.x / nblk, r0 = (blockIdx.x - b * nblk) * V_RB;
    const int nrow = p.L - r0 < V_RB ? p.L - r0 : V_RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    float cmax = 0.f;                                  // max |x| of what this thread rounds to T (calibration forward: p.calib)
    const T* q = reinterpret_cast<const T*>(p.q) + ((size_t)b * p.L + r0) * V_D;
    const T* I = reinterpret_cast<const T*>(p.I) + ((size_t)b * p.L + r0) * V_D;
    T* out = reinterpret_cast<T*>(p.out[st]) + ((size_t)b * p.L + r0) * V_D;

    const int nb = wave * 32;                          // this wave's 32 output channels of every 256-wide GEMM
    // ---- the layer's weights as one stream of K tiles: tile t < 4: fc_o k-range t; then per 256-column slice c of the FFN four tiles
    // of fc1 rows [256c, 256c+256) and four tiles of fc2 k-range [256c + 64v, +64)
    const int nslice = p.d_ff / 256;
    const int n_tiles = 4 + 8 * nslice;
    const unsigned lds_w = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sW;
    const v_v4i r_o = v_make_rsrc(p.wo, 256u * 256u * 2u), r_1 = v_make_rsrc(p.w1, (unsigned)p.d_ff * 256u * 2u), r_2 = v_make_rsrc(p.w2, 256u * (unsigned)p.d_ff * 2u);
    const int rin = lane >> 3, csrc = (lane & 7) ^ rin;
    auto dma_tile = [&](int t) {
        if (t >= n_tiles) return;
        const unsigned dst = lds_w + (unsigned)(t & 1) * V_WT;
        int row0 = 0, k0 = t * 64, ld = 256, which = 0;
        if (t >= 4) {
            const int u = t - 4, c = u >> 3, v8 = u & 7;
            if (v8 < 4) { which = 1; row0 = c * 256; k0 = v8 * 64; ld = 256; }
            else { which = 2; row0 = 0; k0 = c * 256 + (v8 - 4) * 64; ld = p.d_ff; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = wave * 4 + i;                                         // 8-row piece of the 256-row tile
            const unsigned off = (unsigned)((row0 + q * 8 + rin) * ld + k0 + csrc * 8) * 2u;
            if (which == 0) v_dma16(dst + q * 1024, off, r_o); else if (which == 1) v_dma16(dst + q * 1024, off, r_1); else v_dma16(dst + q * 1024, off, r_2);
        }
    };
    int wt = 0;                                        // next tile to consume
    // consume tile `wt` against columns [64 kt, +64) of the activation block; the following tile is requested first and has landed
    // (every wave's pieces: vmcnt(0) + barrier) when the call returns
    auto w_step = [&](v_f32x4 (&acc)[2][V_MF], const char* sAct, int kt) {
        if (!(V_DBG(p) & 2)) dma_tile(wt + 1);
        if (!(V_DBG(p) & 4)) v_mma_tile<T>(acc, sW + (wt & 1) * V_WT, sAct, kt, nb, fr, fg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ++wt;
    };
    dma_tile(0);                                       // fc_o's first tile travels while the attention runs
    // Every small per-channel vector this lane will need (biases, LayerNorm gains) and its residual rows of I are requested NOW: left at
    // their points of use they were ~10 dependent L2 / HBM round trips on the workgroup's critical path (25 us of an otherwise empty kernel).
    float4 p_bo[2], p_b2[2], p_g1[2], p_be1[2], p_g2[2], p_be2[2], p_b1[2];
    uint2 p_res[2][V_MF];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = nb + i * 16 + fg * 4;
        p_bo[i] = *reinterpret_cast<const float4*>(p.bo + n); p_b2[i] = *reinterpret_cast<const float4*>(p.b2 + n);
        p_g1[i] = *reinterpret_cast<const float4*>(p.g1 + n); p_be1[i] = *reinterpret_cast<const float4*>(p.be1 + n);
        p_g2[i] = *reinterpret_cast<const float4*>(p.g2 + n); p_be2[i] = *reinterpret_cast<const float4*>(p.be2 + n);
        p_b1[i] = *reinterpret_cast<const float4*>(p.b1 + n);
#pragma unroll
        for (int j = 0; j < V_MF; ++j) {
            const int row = j * 16 + fr;
            p_res[i][j] = make_uint2(0u, 0u);
            if (row < nrow) p_res[i][j] = *reinterpret_cast<const uint2*>(I + (size_t)row * V_D + n);
        }
    }
    // ---- attention output of the block into sA (rows >= nrow: zeros)
    if (p.fuse_att) {
        const int Lk = p.Lk[st];
        const T* kv = reinterpret_cast<const T*>(p.kv[st]) + (size_t)b * Lk * 512;
        if (!(V_DBG(p) & 1)) v_attention_mfma<T>(q, kv, Lk, nrow, sKV, reinterpret_cast<T*>(sKV + 4 * 32 * 128), sA, tid);
    } else {
        const T* att = reinterpret_cast<const T*>(p.att[st]) + ((size_t)b * p.L + r0) * V_D;
        for (int e = tid; e < V_RB * 32; e += 512) {
            const int row = e >> 5, c = e & 31;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (row < nrow) v = *reinterpret_cast<const uint4*>(att + (size_t)row * V_D + c * 8);
            *reinterpret_cast<uint4*>(sA + row * V_LDA + c * 16) = v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (V_DBG(p) & 16) return;

    float v[2][V_MF][4];
    // ---- x1 = LayerNorm(I + att Wo^T + bo)
    {
        v_f32x4 acc[2][V_MF];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < V_MF; ++j) acc[i][j] = (v_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) w_step(acc, sA, kt);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 bb = p_bo[i];
#pragma unroll
            for (int j = 0; j < V_MF; ++j) {
                float r4[4];
                v_ld4<T>(reinterpret_cast<const char*>(&p_res[i][j]), r4);
                v[i][j][0] = acc[i][j][0] + bb.x + r4[0]; v[i][j][1] = acc[i][j][1] + bb.y + r4[1];
                v[i][j][2] = acc[i][j][2] + bb.z + r4[2]; v[i][j][3] = acc[i][j][3] + bb.w + r4[3];
            }
        }
    }
    v_layernorm<T>(v, p_g1, p_be1, sRed, wave, fr, fg);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < V_MF; ++j) v_st4<T>(sX + (j * 16 + fr) * V_LDA + (nb + i * 16 + fg * 4) * 2, v[i][j], cmax);
    __syncthreads();
    if (V_DBG(p) & 32) return;

    // ---- FFN in 256-column slices of the intermediate: H_c = relu(x1 W1[c]^T + b1[c]) (-> sA), acc2 += H_c W2[:, c]^T
    v_f32x4 acc2[2][V_MF];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < V_MF; ++j) acc2[i][j] = (v_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nslice; ++c) {
        v_f32x4 acc1[2][V_MF];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < V_MF; ++j) acc1[i][j] = (v_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) w_step(acc1, sX, kt);
        float4 b1c[2] = {p_b1[0], p_b1[1]};
        if (c + 1 < nslice) {                          // the next slice's fc1 bias: requested a slice ahead
#pragma unroll
            for (int i = 0; i < 2; ++i) p_b1[i] = *reinterpret_cast<const float4*>(p.b1 + (c + 1) * 256 + nb + i * 16 + fg * 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 bb = b1c[i];
#pragma unroll
            for (int j = 0; j < V_MF; ++j) {
                const float h[4] = {relu_f(acc1[i][j][0] + bb.x), relu_f(acc1[i][j][1] + bb.y), relu_f(acc1[i][j][2] + bb.z),
                                    relu_f(acc1[i][j][3] + bb.w)};
                v_st4<T>(sA + (j * 16 + fr) * V_LDA + (nb + i * 16 + fg * 4) * 2, h, cmax);
            }
        }
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) w_step(acc2, sA, kt);         // (its last barrier also frees sA for the next slice)
    }
    if (V_DBG(p) & 64) return;
    // ---- out = LayerNorm(x1 + ffn + b2)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = nb + i * 16 + fg * 4;
        const float4 bb = p_b2[i];
#pragma unroll
        for (int j = 0; j < V_MF; ++j) {
            float r4[4];
            v_ld4<T>(sX + (j * 16 + fr) * V_LDA + n * 2, r4);
            v[i][j][0] = acc2[i][j][0] + bb.x + r4[0]; v[i][j][1] = acc2[i][j][1] + bb.y + r4[1];
            v[i][j][2] = acc2[i][j][2] + bb.z + r4[2]; v[i][j][3] = acc2[i][j][3] + bb.w + r4[3];
        }
    }
    v_layernorm<T>(v, p_g2, p_be2, sRed, wave, fr, fg);
    // rounded output -> LDS image -> 16-byte row stores; pooled mean over the instruction's own tokens from the rounded values
    int len = p.L;
    if (p.lens) { len = p.lens[b]; len = len < 1 ? 1 : len > p.L ? p.L : len; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < V_MF; ++j) {
            char* dst = sA + (j * 16 + fr) * V_LDA + (nb + i * 16 + fg * 4) * 2;
            v_st4<T>(dst, v[i][j], cmax);
            if (p.pooled[st] && r0 + j * 16 + fr < len) {
                float r4[4];
                v_ld4<T>(dst, r4);
#pragma unroll
                for (int e = 0; e < 4; ++e) cs[e] += r4[e];
            }
        }
        if (p.pooled[st]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = cs[e];
                t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
                cs[e] = t;
            }
            if (fr == 0) {
                float* dst = p.pooled[st] + (size_t)b * p.ld_pool + nb + i * 16 + fg * 4;
                const float invl = 1.0f / (float)len;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[e] = cs[e] * invl;
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < nrow * 32; e += 512) {
        const int row = e >> 5, c = e & 31;
        *reinterpret_cast<uint4*>(out + (size_t)row * V_D + c * 8) = *reinterpret_cast<const uint4*>(sA + row * V_LDA + c * 16);
    }
    if (p.calib) {                                     // fp16 range check (absmax_kernel's slot format: max |x| bits, non-finite count)
        const float m = wave_max(cmax);
        if (lane == 0) {
            if (m <= 3.0e38f) atomicMax(p.calib, __float_as_uint(m));
            else atomicAdd(p.calib + 1, 1u);
        }
    }
}

// ---- Round 6: the same layer with its WEIGHTS STRAIGHT FROM L2 INTO REGISTERS (p.wfrag: wo / w1 / w2 in MFMA-fragment order, launch_pack_frag).
// A wave consumes only ITS 32 output channels of every weight tile, so nothing about the weights is shared between the waves of a workgroup: the
// LDS ring above was a 32 KB-per-step latency chain behind a workgroup barrier (36 steps of ~0.53 us for 0.15-0.2 us of MFMAs each, every step
// draining `vmcnt(0)`).  Here a lane reads the 16 bytes of its operand fragment directly (one contiguous 1 KB request per wave and fragment --
// what a request costs depends on the level it is served from, not on its shape, and these 1.15 MB are L2-resident for every workgroup but the
// first), half a 256-deep GEMM (8 fragments, 32 VGPRs) per request round into two register sets, the next half requested as soon as the
// current one's MFMAs are issued; the only barriers left are the ones the ACTIVATION buffers need (2 per FFN slice instead of 8, 12 instead of
// 40 per launch at d_ff = 1024).  The per-channel vectors of the second half of the layer (b2, LayerNorm 2) wait in LDS instead of 24 VGPRs.
// Same MFMA instruction over the same k order on the same operands: bit-identical to vla_post_kernel.
constexpr int V_PV = 3 * V_D * 4;                                   // b2 | g2 | be2 as f32
constexpr size_t V_LDS_WF = (size_t)2 * V_RB * V_LDA + (size_t)8 * V_RB * 2 * sizeof(float) + V_PV;

typedef uint4 v_half_t[2][2][2];                                    // [k tile of the half][k step][16-channel fragment]

__device__ __forceinline__ uint4 v_ld16(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
}
// fragments (k step ksg0 .. ksg0 + 3) x (two 16-channel tiles from ct0) of a fragment-order matrix with `nct` 16-channel tiles
__device__ __forceinline__ void v_load_half(v_half_t& R, __amdgpu_buffer_rsrc_t rs, int ksg0, int nct, int ct0, int lane) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned off = (unsigned)((((ksg0 + kt * 2 + ks) * nct + ct0) * 64 + lane) * 16);
            R[kt][ks][0] = v_ld16(rs, off);
            R[kt][ks][1] = v_ld16(rs, off + 1024u);
        }
}
template <typename T>
__device__ __forceinline__ void v_mma_half(v_f32x4 (&acc)[2][V_MF], const v_half_t& R, const char* sAct, int kt0, int fr, int fg) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 xf[V_MF];
#pragma unroll
            for (int j = 0; j < V_MF; ++j) xf[j] = *reinterpret_cast<const uint4*>(sAct + (j * 16 + fr) * V_LDA + ((kt0 + kt) * 64 + ks * 32 + fg * 8) * 2);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < V_MF; ++j) VMma<T>::run(acc[i][j], R[kt][ks][i], xf[j]);
        }
}

template <typename T>
__global__ __launch_bounds__(512) void vla_post_wf_kernel(VlaPost p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                   // attention output, later the FFN intermediate slice, last the output image
    char* sX = smem + V_RB * V_LDA;                    // x1 = LayerNorm(I + att Wo^T)
    char* sKV = sX;                                    // K | V of the in-kernel attention (dead before x1 is written)
    float* sRed = reinterpret_cast<float*>(sX + V_RB * V_LDA);
    float* sPV = sRed + 8 * V_RB * 2;                  // b2 | g2 | be2
    const int st = blockIdx.y;
    const int nblk = (p.L + V_RB - 1) / V_RB;
    const int b = blockIdx.x / nblk, r0 = (blockIdx.x - b * nblk) * V_RB;
    const int nrow = p.L - r0 < V_RB ? p.L - r0 : V_RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    float cmax = 0.f;
    const T* q = reinterpret_cast<const T*>(p.q) + ((size_t)b * p.L + r0) * V_D;
    const T* I = reinterpret_cast<const T*>(p.I) + ((size_t)b * p.L + r0) * V_D;
    T* out = reinterpret_cast<T*>(p.out[st]) + ((size_t)b * p.L + r0) * V_D;
    const int nb = wave * 32;
    const int nslice = p.d_ff / 256;
    const __amdgpu_buffer_rsrc_t r_o = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wo), 0, 256 * 256 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, p.d_ff * 256 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), 0, 256 * p.d_ff * 2, 0x00020000);
    const int nct1 = p.d_ff / 16;

    v_half_t R0, R1;
    v_load_half(R0, r_o, 0, 16, 2 * wave, lane);       // fc_o travels while the attention runs
    v_load_half(R1, r_o, 4, 16, 2 * wave, lane);
    float4 p_bo[2], p_g1[2], p_be1[2], p_b1[2];
    uint2 p_res[2][V_MF];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = nb + i * 16 + fg * 4;
        p_bo[i] = *reinterpret_cast<const float4*>(p.bo + n);
        p_g1[i] = *reinterpret_cast<const float4*>(p.g1 + n); p_be1[i] = *reinterpret_cast<const float4*>(p.be1 + n);
        p_b1[i] = *reinterpret_cast<const float4*>(p.b1 + n);
#pragma unroll
        for (int j = 0; j < V_MF; ++j) {
            const int row = j * 16 + fr;
            p_res[i][j] = make_uint2(0u, 0u);
            if (row < nrow) p_res[i][j] = *reinterpret_cast<const uint2*>(I + (size_t)row * V_D + n);
        }
    }
    if (tid < 192) {
        const float* src = tid < 64 ? p.b2 + tid * 4 : tid < 128 ? p.g2 + (tid - 64) * 4 : p.be2 + (tid - 128) * 4;
        *reinterpret_cast<float4*>(sPV + tid * 4) = *reinterpret_cast<const float4*>(src);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- attention output of the block into sA (rows >= nrow: zeros)
    if (p.fuse_att) {
        const int Lk = p.Lk[st];
        const T* kv = reinterpret_cast<const T*>(p.kv[st]) + (size_t)b * Lk * 512;
        v_attention_mfma<T>(q, kv, Lk, nrow, sKV, reinterpret_cast<T*>(sKV + 4 * 32 * 128), sA, tid);
    } else {
        const T* att = reinterpret_cast<const T*>(p.att[st]) + ((size_t)b * p.L + r0) * V_D;
        for (int e = tid; e < V_RB * 32; e += 512) {
            const int row = e >> 5, c = e & 31;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (row < nrow) v = *reinterpret_cast<const uint4*>(att + (size_t)row * V_D + c * 8);
            *reinterpret_cast<uint4*>(sA + row * V_LDA + c * 16) = v;
        }
    }
    v_lds_barrier();

    float v[2][V_MF][4];
    // ---- x1 = LayerNorm(I + att Wo^T + bo)
    {
        v_f32x4 acc[2][V_MF];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < V_MF; ++j) acc[i][j] = (v_f32x4){0.f, 0.f, 0.f, 0.f};
        v_mma_half<T>(acc, R0, sA, 0, fr, fg);
        __builtin_amdgcn_sched_barrier(0);
        v_load_half(R0, r_1, 0, nct1, 2 * wave, lane);                 // fc1 of slice 0
        __builtin_amdgcn_sched_barrier(0);
        v_mma_half<T>(acc, R1, sA, 2, fr, fg);
        __builtin_amdgcn_sched_barrier(0);
        v_load_half(R1, r_1, 4, nct1, 2 * wave, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 bb = p_bo[i];
#pragma unroll
            for (int j = 0; j < V_MF; ++j) {
                float r4[4];
                v_ld4<T>(reinterpret_cast<const char*>(&p_res[i][j]), r4);
                v[i][j][0] = acc[i][j][0] + bb.x + r4[0]; v[i][j][1] = acc[i][j][1] + bb.y + r4[1];
                v[i][j][2] = acc[i][j][2] + bb.z + r4[2]; v[i][j][3] = acc[i][j][3] + bb.w + r4[3];
            }
        }
    }
    v_layernorm<T, true>(v, p_g1, p_be1, sRed, wave, fr, fg);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < V_MF; ++j) v_st4<T>(sX + (j * 16 + fr) * V_LDA + (nb + i * 16 + fg * 4) * 2, v[i][j], cmax);
    v_lds_barrier();                                   // x1 complete; every wave is past its reads of sA (the LayerNorm's barriers)

    // ---- FFN in 256-column slices of the intermediate: H_c = relu(x1 W1[c]^T + b1[c]) (-> sA), acc2 += H_c W2[:, c]^T
    v_f32x4 acc2[2][V_MF];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < V_MF; ++j) acc2[i][j] = (v_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nslice; ++c) {
        v_f32x4 acc1[2][V_MF];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < V_MF; ++j) acc1[i][j] = (v_f32x4){0.f, 0.f, 0.f, 0.f};
        v_mma_half<T>(acc1, R0, sX, 0, fr, fg);
        __builtin_amdgcn_sched_barrier(0);
        v_load_half(R0, r_2, 8 * c, 16, 2 * wave, lane);               // fc2's k range of this slice
        __builtin_amdgcn_sched_barrier(0);
        v_mma_half<T>(acc1, R1, sX, 2, fr, fg);
        __builtin_amdgcn_sched_barrier(0);
        v_load_half(R1, r_2, 8 * c + 4, 16, 2 * wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        // (the last slice requests its own fc1 rows / bias again instead of branching: a conditional request makes hipcc's vmcnt counts at the
        //  merge conservative, which turns the wait for the OLDER register set into a wait for the one just requested)
        const int cn = c + 1 < nslice ? c + 1 : c;
        float4 b1c[2] = {p_b1[0], p_b1[1]};
#pragma unroll
        for (int i = 0; i < 2; ++i) p_b1[i] = *reinterpret_cast<const float4*>(p.b1 + cn * 256 + nb + i * 16 + fg * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 bb = b1c[i];
#pragma unroll
            for (int j = 0; j < V_MF; ++j) {
                const float h[4] = {relu_f(acc1[i][j][0] + bb.x), relu_f(acc1[i][j][1] + bb.y), relu_f(acc1[i][j][2] + bb.z),
                                    relu_f(acc1[i][j][3] + bb.w)};
                v_st4<T>(sA + (j * 16 + fr) * V_LDA + (nb + i * 16 + fg * 4) * 2, h, cmax);
            }
        }
        v_lds_barrier();                               // H_c complete
        v_mma_half<T>(acc2, R0, sA, 0, fr, fg);
        __builtin_amdgcn_sched_barrier(0);
        v_load_half(R0, r_1, 0, nct1, 16 * cn + 2 * wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        v_mma_half<T>(acc2, R1, sA, 2, fr, fg);
        __builtin_amdgcn_sched_barrier(0);
        v_load_half(R1, r_1, 4, nct1, 16 * cn + 2 * wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        v_lds_barrier();                               // sA free: the next slice's H, or the output image
    }
    // ---- out = LayerNorm(x1 + ffn + b2)
    float4 p_b2[2], p_g2[2], p_be2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = nb + i * 16 + fg * 4;
        p_b2[i] = *reinterpret_cast<const float4*>(sPV + n);
        p_g2[i] = *reinterpret_cast<const float4*>(sPV + V_D + n);
        p_be2[i] = *reinterpret_cast<const float4*>(sPV + 2 * V_D + n);
        const float4 bb = p_b2[i];
#pragma unroll
        for (int j = 0; j < V_MF; ++j) {
            float r4[4];
            v_ld4<T>(sX + (j * 16 + fr) * V_LDA + n * 2, r4);
            v[i][j][0] = acc2[i][j][0] + bb.x + r4[0]; v[i][j][1] = acc2[i][j][1] + bb.y + r4[1];
            v[i][j][2] = acc2[i][j][2] + bb.z + r4[2]; v[i][j][3] = acc2[i][j][3] + bb.w + r4[3];
        }
    }
    v_layernorm<T, true>(v, p_g2, p_be2, sRed, wave, fr, fg);
    int len = p.L;
    if (p.lens) { len = p.lens[b]; len = len < 1 ? 1 : len > p.L ? p.L : len; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < V_MF; ++j) {
            char* dst = sA + (j * 16 + fr) * V_LDA + (nb + i * 16 + fg * 4) * 2;
            v_st4<T>(dst, v[i][j], cmax);
            if (p.pooled[st] && r0 + j * 16 + fr < len) {
                float r4[4];
                v_ld4<T>(dst, r4);
#pragma unroll
                for (int e = 0; e < 4; ++e) cs[e] += r4[e];
            }
        }
        if (p.pooled[st]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = cs[e];
                t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
                cs[e] = t;
            }
            if (fr == 0) {
                float* dst = p.pooled[st] + (size_t)b * p.ld_pool + nb + i * 16 + fg * 4;
                const float invl = 1.0f / (float)len;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[e] = cs[e] * invl;
            }
        }
    }
    v_lds_barrier();
    for (int e = tid; e < nrow * 32; e += 512) {
        const int row = e >> 5, c = e & 31;
        *reinterpret_cast<uint4*>(out + (size_t)row * V_D + c * 8) = *reinterpret_cast<const uint4*>(sA + row * V_LDA + c * 16);
    }
    if (p.calib) {
        const float m = wave_max(cmax);
        if (lane == 0) {
            if (m <= 3.0e38f) atomicMax(p.calib, __float_as_uint(m));
            else atomicAdd(p.calib + 1, 1u);
        }
    }
}

bool vla_post_ok(int dt, int d_model, int heads, int d_ff) {
    return (dt == DT_BF16 || dt == DT_F16) && d_model == V_D && heads == 4 && d_ff >= 256 && d_ff % 256 == 0;
}

hipError_t launch_vla_post(const VlaPost& p, int dt, hipStream_t s) {
    if (dt != DT_BF16 && dt != DT_F16) return hipErrorInvalidValue;
    if (p.B < 1 || p.L < 1 || p.d_ff % 256) return hipErrorInvalidValue;
    for (int st = 0; st < p.streams; ++st)
        if (p.fuse_att && (p.Lk[st] < 1 || p.Lk[st] > V_KVMAX)) return hipErrorInvalidValue;
    // the pooled mean is complete inside one workgroup only when the block covers the whole instruction
    if ((p.pooled[0] || p.pooled[1]) && p.L > V_RB) return hipErrorInvalidValue;
    const void* fn = dt == DT_BF16 ? reinterpret_cast<const void*>(vla_post_kernel<bf16>) : reinterpret_cast<const void*>(vla_post_kernel<f16>);
    if (p.wfrag) fn = dt == DT_BF16 ? reinterpret_cast<const void*>(vla_post_wf_kernel<bf16>) : reinterpret_cast<const void*>(vla_post_wf_kernel<f16>);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vla_post_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vla_post_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vla_post_wf_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vla_post_wf_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_once.done();
    }
    VlaPost q = p;
#ifdef HCM_DEV_KNOBS
    static const int dbg = dev_env("HCM_VLA_DBG") ? atoi(dev_env("HCM_VLA_DBG")) : 0;      // `make DEV=1` builds only: timing experiments (results then wrong)
    q.dbg = dbg;
#else
    q.dbg = 0;
#endif
    void* args[] = {&q};
    const int nblk = (p.L + V_RB - 1) / V_RB;
    return hipLaunchKernel(fn, dim3(p.B * nblk, p.streams), dim3(512), args, p.wfrag ? V_LDS_WF : V_LDS, s);
}

}  // namespace hcm
