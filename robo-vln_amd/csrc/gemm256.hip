// Large-tile GEMM for the token-major linear layers whose output is wide (BERT's QKV and FFN1, the cross-modal FFN):
//
//   Y[m][n] = act( sum_k X[m][k] * W[n][k] + bias[n] )        X [M][ldx], W [N][ldw] (K contiguous), 16-bit storage, fp32 accumulate
//
// Why a second GEMM kernel.  igemm_dma_kernel (igemm.hip) works on 128 x 128 tiles: per 64-deep K step a CU moves 32 KB from L2 into
// LDS and reads 96 KB of fragments back out of LDS for 2 x 256 MFMA-cycles per SIMD -- at the matrix pipe's peak that would be
// 36 TB/s of L2 traffic and 75 % of the LDS read bandwidth, so the pipe idles (16-23 % MFMA busy, profiles/r2a_pmc_mfma_bench.md).
// A 256 x 256 tile (8 waves as 2 x 4, 128 x 64 outputs per wave, 128 accumulator registers) halves both ratios, and the K loop is
// built the way the CDNA4 guide's "8-phase" template is (cdna_hip_programming.md section 5, T1-T5):
//   * per 64-deep K tile a wave runs FOUR phases -- one 64 x 32 quadrant of its output each: {fragment reads + 2 LDS-DMA pieces}
//     -> s_barrier -> 16 MFMAs -> s_barrier;
//   * the two wave groups (rows 0-127 / 128-255 of the tile = the two waves of every SIMD) are staggered by one barrier, so while
//     one wave of a SIMD issues its 16 MFMAs the other one reads its next fragments and issues its DMA pieces;
//   * both operand tiles travel L2 -> LDS by `buffer_load_dwordx4 ... lds` into two 64 KB K-tile buffers, one half-tile (16 KB =
//     2 pieces per wave) per phase, a full K tile ahead; waits are counted (`vmcnt(4)` / `vmcnt(2)`, never 0 in the steady state), so
//     DMA pieces stay in flight across the barriers;
//   * LDS rows are 128 B with the 16-byte chunk index XOR-swizzled by (row & 7) on the SOURCE side of the DMA and on the fragment
//     reads (rule 21 / T2), exactly as in igemm_dma_kernel.
// The MFMA instruction, the k order inside a K tile and the order of the K tiles are those of igemm_dma_kernel, and so are the epilogue's
// f32 operations (acc + bias, erf-GELU / ReLU, one rounding): the results are BIT-IDENTICAL to that kernel's (tests/test_ops_gpu.py).
//
// Epilogue: the accumulators (4 consecutive channels of one token per lane) go through a 16-bit LDS image of the whole 256 x 256 output
// tile (rows padded to 528 B: conflict-free 8-byte writes), from which every thread stores 16-byte pieces of whole 512-byte rows.
//
// Reference ops replaced: nn.Linear of transformers' BertSelfAttention (query|key|value, concatenated) / BertIntermediate (+ GELU), as
// called at seq2seq_highlevel_cma.py:192-195, and PositionWiseFeedForward.fc1 (transformer.py:25-43).
#include <cstdlib>
#include <type_traits>
#include "kernels.h"
#include "dev.h"

namespace hcm {

typedef float g_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
typedef int g_v4i __attribute__((ext_vector_type(4)));
typedef unsigned g_u32x4 __attribute__((ext_vector_type(4)));

struct G256Dev {
    const char* x; const char* w; const float* bias; char* y;
    const char* res; int ldr;        // optional residual [M][ldr] (T), added in f32 before the activation and the one rounding
    int M, N, K, ldx, ldw, ldy, act;
    unsigned x_bytes, w_bytes, y_bytes;
    int tilesM, tilesN, gm, gn;      // XCD grid: the 8 XCDs own gm x gn rectangles of the tile grid (each has a private L2)
    // folded LayerNorm (IGemm::ln_*): x holds the pre-LayerNorm rows u, w = W diag(gamma)
    const float* ln_s; const float* ln_t; float* ln_stats_out; float ln_eps;
    const float* ln_part; int ln_part_P;          // IGemm::ln_part_in: partial row sums from the producer's epilogue (else: taken in the K loop)
};

template <typename T> struct GMma;
template <> struct GMma<bf16> {
    static __device__ __forceinline__ void run(g_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g_bf16x8, a), __builtin_bit_cast(g_bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct GMma<f16> {
    static __device__ __forceinline__ void run(g_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), acc, 0, 0, 0);
    }
};

__device__ __forceinline__ void g_dma16(unsigned lds_addr, unsigned voff, g_v4i rsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
template <int N> __device__ __forceinline__ void g_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
__device__ __forceinline__ g_v4i g_make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    g_v4i r;
    r[0] = (int)(unsigned)a;
    r[1] = (int)((unsigned)(a >> 32) & 0xFFFFu);
    r[2] = (int)bytes;
    r[3] = 0x00020000;
    return r;
}
#define G_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

constexpr int G_BM = 256, G_BN = 256;
constexpr unsigned G_BUF = 65536;          // one K tile: X 256 x 128 B, then W 256 x 128 B
constexpr unsigned G_WOFF = 32768;
constexpr int G_IMG_LD = 528;              // bytes per row of the epilogue's 16-bit tile image (512 + 16 pad)
constexpr size_t G_LDS = (size_t)G_BM * G_IMG_LD;      // 135168 >= 2 * G_BUF

// one quadrant (32 channels x 64 tokens) x one 64-deep K tile: 16 MFMAs, k order = igemm_dma_kernel's (ks outer)
template <typename T, int NQ, int MQ>
__device__ __forceinline__ void g_quad(g_f32x4 (&acc)[4][8], const uint4 (&wf)[2][2], const uint4 (&xf)[2][4]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) GMma<T>::run(acc[NQ * 2 + i][MQ * 4 + j], wf[ks][i], xf[ks][j]);
}

// Epilogue shared by the 8-wave kernels (gemm256_kernel, gemm256f_kernel): wave (wr, wc) holds tokens [wr*128, +128) x channels [wc*64, +64) of the tile,
// acc[i][j] = channel fragment i x token fragment j, a lane holds 4 consecutive channels of one token.
// LNF: the folded-LayerNorm form -- v = rstd[row] * (acc - mean[row] * ln_s[n]) + ln_t[n] instead of acc + bias[n] (mean / rstd: the lane's 8 token rows j)
template <typename T, bool LNF = false>
__device__ __forceinline__ void g_epilogue(const G256Dev& p, g_f32x4 (&acc)[4][8], char* smem, int m0, int n0, int tid, int wr, int wc, int fr, int fg,
                                           const float* ln_mean = nullptr, const float* ln_rstd = nullptr) {
    if (p.res) {
        // ---- residual epilogue: y = act(acc + bias + res), one rounding -- the order of f32 operations of igemm_epilogue.  The accumulators
        // go through an f32 LDS image of HALF the tile at a time (128 rows x 256 channels, rows padded to 1040 B); in the row pass every
        // thread adds bias + 8 residual channels read as one 16-byte piece, applies the activation, rounds and stores 16 bytes.
        constexpr int LDF = 260;                       // floats per image row
        float* sc = reinterpret_cast<float*>(smem);
        const int chunk = tid & 31, n = n0 + chunk * 8;
        float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
            bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
        }
        for (int half = 0; half < 2; ++half) {
            // the residual rows of this half: requested before the image is written, consumed after the barrier
            uint4 rr[8];
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                const int m = m0 + half * 128 + pass * 16 + (tid >> 5);
                rr[pass] = make_uint4(0u, 0u, 0u, 0u);
                if (m < p.M && n < p.N) rr[pass] = *reinterpret_cast<const uint4*>(p.res + ((size_t)m * p.ldr + n) * 2);
            }
            if (wr == half) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *reinterpret_cast<float4*>(sc + (j * 16 + fr) * LDF + wc * 64 + i * 16 + fg * 4) =
                            make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
            __syncthreads();
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                const int r = pass * 16 + (tid >> 5);
                const int m = m0 + half * 128 + r;
                if (m < p.M && n < p.N) {
                    const float4 a0 = *reinterpret_cast<const float4*>(sc + r * LDF + chunk * 8), a1 = *reinterpret_cast<const float4*>(sc + r * LDF + chunk * 8 + 4);
                    float v[8] = {a0.x + bias8[0], a0.y + bias8[1], a0.z + bias8[2], a0.w + bias8[3], a1.x + bias8[4], a1.y + bias8[5], a1.z + bias8[6], a1.w + bias8[7]};
                    float r8[8];
                    cvt_chunk<T>(rr[pass], r8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r8[e];
                    if (p.act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e]);
                    } else if (p.act == ACT_GELU) {
                        gelu_vec<T, 8>(v);
                    }
                    st_chunk(reinterpret_cast<T*>(p.y + ((size_t)m * p.ldy + n) * 2), v);
                }
            }
            __syncthreads();
        }
        return;
    }
    // ---- epilogue: acc + bias, activation, one rounding -> 16-bit tile image in LDS -> 16-byte stores of whole rows.
    // In NP = 2 parts of the wave's rows (fragment rows 0-3 / 4-7, i.e. tile rows {64 q .. 64 q + 63} of both row groups): the stores of
    // the first part are in flight while the second part's bias / activation / rounding arithmetic runs (FFN1's 128 GELUs per thread are
    // ~6 us of VALU work, the tile's 128 KB of stores ~6 us of drain).  NP = 4 measured no better (FFN1 34.3 vs 33.6 us, no-activation
    // shapes 0.4 us faster: inside the box-to-box noise).  The image rows of the parts are disjoint: one barrier per part.
    {
        constexpr int NP = 2, JP = 8 / NP, RP = JP * 16;          // parts, fragment rows per part, tile rows per part and row group
        float4 b4[4], s4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wc * 64 + i * 16 + fg * 4;
            if constexpr (LNF) {
                b4[i] = n < p.N ? *reinterpret_cast<const float4*>(p.ln_t + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                s4[i] = n < p.N ? *reinterpret_cast<const float4*>(p.ln_s + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                b4[i] = (p.bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                s4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const int chunk = tid & 31;                    // 16-byte piece of a 512-byte row
        const int n = n0 + chunk * 8;
#pragma unroll
        for (int part = 0; part < NP; ++part) {
#pragma unroll
            for (int jj = 0; jj < JP; ++jj) {
                const int j = part * JP + jj;
                // the fragment row's 16 values at once: eight independent GELU chains for the scheduler to interleave (a chain is ~14 dependent
                // packed-f32 operations; taken four values at a time, with the activation switch inside, the epilogue was latency-bound)
                float v[16];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (LNF) {
                        const float mu = ln_mean[j], rs = ln_rstd[j];
                        v[i * 4 + 0] = (acc[i][j][0] - mu * s4[i].x) * rs + b4[i].x; v[i * 4 + 1] = (acc[i][j][1] - mu * s4[i].y) * rs + b4[i].y;
                        v[i * 4 + 2] = (acc[i][j][2] - mu * s4[i].z) * rs + b4[i].z; v[i * 4 + 3] = (acc[i][j][3] - mu * s4[i].w) * rs + b4[i].w;
                    } else {
                        v[i * 4 + 0] = acc[i][j][0] + b4[i].x; v[i * 4 + 1] = acc[i][j][1] + b4[i].y;
                        v[i * 4 + 2] = acc[i][j][2] + b4[i].z; v[i * 4 + 3] = acc[i][j][3] + b4[i].w;
                    }
                }
                if (p.act == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = relu_f(v[e]);
                } else if (p.act == ACT_GELU) {
                    gelu_vec<T, 16>(v);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    T o4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) Tr<T>::st(&o4[e], v[i * 4 + e]);
                    const int r = wr * 128 + j * 16 + fr;
                    const int cb = (wc * 64 + i * 16 + fg * 4) * 2;
                    *reinterpret_cast<uint2*>(smem + r * G_IMG_LD + cb) = *reinterpret_cast<const uint2*>(o4);
                }
            }
            __syncthreads();
            if (n < p.N) {                             // N % 8 == 0 (launcher)
#pragma unroll
                for (int pass = 0; pass < 2 * RP / 16; ++pass) {
                    const int rr = pass * 16 + (tid >> 5);                      // 0 .. 2 RP - 1 over the part's rows
                    const int r = (rr / RP) * 128 + part * RP + (rr % RP);
                    const int m = m0 + r;
                    if (m < p.M)
                        *reinterpret_cast<uint4*>(p.y + ((size_t)m * p.ldy + n) * 2) = *reinterpret_cast<const uint4*>(smem + r * G_IMG_LD + chunk * 16);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Phase timing of the 8-wave K loops (`make DEV=1` builds: hcm_op_linear_impl variants 13-15; read back through hcm_debug_gemm256_prof).
// Per wave of workgroup 0..kG256ProfWgs-1, cycle totals over the K tiles of one launch (s_memtime ticks = shader cycles):
//   gemm256_kernel  (8-phase): [0] load segment incl. the wait at barrier 1 (phase start -> released + fragments arrived), [1] MFMA issue,
//                              [2] wait at barrier 2, [3] (PROF 2 only) the part of [0] that is the barrier-1 wait, [4] prologue, [5] epilogue
//   gemm256f_kernel (free)   : [0] first half (32 MFMAs + the other K half's fragment reads), [1] lgkmcnt + vmcnt waits, [2] the barrier,
//                              [3] second half (32 MFMAs + 8 DMA requests + the next tile's fragment reads), [4] prologue, [5] epilogue
//   [6] K tiles, [7] launches
constexpr int kG256ProfWgs = 16;
__device__ unsigned long long g_g256_prof[kG256ProfWgs][8][8];
__device__ __forceinline__ unsigned long long g_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// SCHED: how the 8 DMA pieces a wave requests per K tile are spread over the four phases (0: 2/2/2/2; 1: 0/3/1/4 -- the pieces go where
// the phase has few fragment reads: 12/4/8/0).  DBG (timing experiments only, results are then wrong): 1 no DMA in the loop, 2 no fragment
// reads after the first tile, 4 no MFMAs.
template <typename T, int SCHED = 0, int DBG = 0>
__global__ __launch_bounds__(512) void gemm256_kernel(G256Dev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PROF = (DBG & 16) ? 2 : (DBG & 8) ? 1 : 0;      // phase timing (see g_g256_prof)
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
    if constexpr (PROF) t_prev = g_now();
    auto lap = [&](int slot) {
        if constexpr (PROF) { const unsigned long long t = g_now(); pt[slot] += t - t_prev; t_prev = t; }
    };
    // ---- tile of this block: XCD (blockIdx % 8) owns a rectangle of the tile grid
    int tile_m, tile_n;
    {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        const int xm = xcd / p.gn, xn = xcd - xm * p.gn;
        const int m_lo = xm * p.tilesM / p.gm, m_hi = (xm + 1) * p.tilesM / p.gm;
        const int n_lo = xn * p.tilesN / p.gn, n_hi = (xn + 1) * p.tilesN / p.gn;
        const int nn = n_hi - n_lo;
        if (nn <= 0 || local >= (m_hi - m_lo) * nn) return;
        tile_m = m_lo + local / nn;
        tile_n = n_lo + local % nn;
    }
    const int m0 = tile_m * G_BM, n0 = tile_n * G_BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;          // wave (wr, wc): tokens [wr*128, +128) x channels [wc*64, +64)
    const int rin = lane >> 3;                        // row inside an 8-row DMA piece
    const int csrc = (lane & 7) ^ rin;                // source chunk this lane fetches (swizzle on the source side)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const g_v4i rx = g_make_rsrc(p.x, p.x_bytes);
    const g_v4i rw = g_make_rsrc(p.w, p.w_bytes);

    // DMA pieces of this wave: per half-tile h (X-h: token rows {h*64..+64} of both row groups; W-h: channel rows {h*32..+32} of all
    // four column groups) pieces q = 2*wave + i, i = 0, 1.  Out-of-range rows present an offset beyond num_records: hardware zero fill.
    unsigned xsrc[2][2], wsrc[2][2], xdst[2][2], wdst[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = 2 * wave + i;
            const int xrow = (q >> 3) * 128 + h * 64 + (q & 7) * 8;
            const int wrow = (q >> 2) * 64 + h * 32 + (q & 3) * 8;
            const int m = m0 + xrow + rin, n = n0 + wrow + rin;
            xsrc[h][i] = m < p.M ? (unsigned)(m * p.ldx + csrc * 8) * 2u : 0x80000000u;
            wsrc[h][i] = n < p.N ? (unsigned)(n * p.ldw + csrc * 8) * 2u : 0x80000000u;
            xdst[h][i] = lds_base + (unsigned)xrow * 128u;
            wdst[h][i] = lds_base + G_WOFF + (unsigned)wrow * 128u;
        }
    auto dma_x = [&](int h, unsigned buf, unsigned kbyte) {
#pragma unroll
        for (int i = 0; i < 2; ++i) g_dma16(xdst[h][i] + buf, xsrc[h][i] + kbyte, rx);
    };
    auto dma_w = [&](int h, unsigned buf, unsigned kbyte) {
#pragma unroll
        for (int i = 0; i < 2; ++i) g_dma16(wdst[h][i] + buf, wsrc[h][i] + kbyte, rw);
    };
    // piece q of a K tile in request order: X-h0[0,1], W-h0[0,1], W-h1[0,1], X-h1[0,1] (the order the next tile's phases need them)
    auto piece = [&](int q, unsigned buf, unsigned kbyte) {
        const int i = q & 1;
        if (q < 2) g_dma16(xdst[0][i] + buf, xsrc[0][i] + kbyte, rx);
        else if (q < 4) g_dma16(wdst[0][i] + buf, wsrc[0][i] + kbyte, rw);
        else if (q < 6) g_dma16(wdst[1][i] + buf, wsrc[1][i] + kbyte, rw);
        else g_dma16(xdst[1][i] + buf, xsrc[1][i] + kbyte, rx);
    };

    g_f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (g_f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / 64;
    // prologue: the whole first K tile
    dma_x(0, 0, 0); dma_w(0, 0, 0); dma_w(1, 0, 0); dma_x(1, 0, 0);
    g_wait_vmcnt<0>();
    G_BAR();
    if (wr == 1) G_BAR();                              // stagger the second row group by one barrier interval
    lap(4);

    const int fr = lane & 15, fg = lane >> 4;
    uint4 xf[2][4], wf0[2][2], wf1[2][2];
    auto rd_x = [&](int mq, unsigned buf) {
        const char* sx = smem + buf;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wr * 128 + mq * 64 + j * 16 + fr;
                xf[ks][j] = *reinterpret_cast<const uint4*>(sx + r * 128 + (((ks * 4 + fg) ^ (r & 7)) << 4));
            }
    };
    auto rd_w = [&](uint4 (&wf)[2][2], int nq, unsigned buf) {
        const char* sw = smem + buf + G_WOFF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wc * 64 + nq * 32 + i * 16 + fr;
                wf[ks][i] = *reinterpret_cast<const uint4*>(sw + r * 128 + (((ks * 4 + fg) ^ (r & 7)) << 4));
            }
    };

    // K tile t lives in buffer t & 1; tile t+1 is requested into the other buffer while t is consumed (that buffer's last reader,
    // the second row group's phase 3 of tile t-1, finished two barrier intervals before the first request)
    // SCHED 2: the two DMA pieces of a phase are issued from INSIDE its MFMA cluster (after the 4th and the 12th MFMA), where an LDS-DMA
    // instruction costs ~60 cycles of issue instead of 100-185 in a phase that is also reading fragments (MI355X_MICROARCH.md, cycle table)
    auto mma_phase_dma = [&](auto NQ, auto MQ, const uint4 (&wf)[2][2], auto LIVE, int q0, unsigned buf, unsigned kbyte) {
        constexpr int nq = decltype(NQ)::value, mq = decltype(MQ)::value;
        constexpr bool live = decltype(LIVE)::value;
        G_BAR();
        __builtin_amdgcn_s_setprio(1);
        int cnt = 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    GMma<T>::run(acc[nq * 2 + i][mq * 4 + j], wf[ks][i], xf[ks][j]);
                    ++cnt;
                    if (live && (cnt == 4 || cnt == 12)) {
                        __builtin_amdgcn_sched_barrier(0);
                        piece(q0 + (cnt == 12), buf, kbyte);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        __builtin_amdgcn_s_setprio(0);
        G_BAR();
    };
    auto mma_phase = [&](auto NQ, auto MQ, const uint4 (&wf)[2][2]) {
        if constexpr (PROF == 2) lap(0);               // (the stamp waits for the fragment reads: they complete BEFORE the barrier in this build)
        G_BAR();
        lap(PROF == 2 ? 3 : 0);
        if constexpr (!(DBG & 4)) {
            __builtin_amdgcn_s_setprio(1);
            g_quad<T, decltype(NQ)::value, decltype(MQ)::value>(acc, wf, xf);
            __builtin_amdgcn_s_setprio(0);
        }
        lap(1);
        G_BAR();
        lap(2);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto tile_body = [&](int t, auto LIVE) {
        constexpr bool live = decltype(LIVE)::value && !(DBG & 1);
        const unsigned cur = (t & 1) ? G_BUF : 0u, nxt = cur ^ G_BUF;
        const unsigned kb = (unsigned)(t + 1) * 128u;
        const bool rd = !(DBG & 2) || t == 0;
        if constexpr (SCHED == 0) {
            // ---- phase 1: channels 0-31 x tokens 0-63 of the wave's block
            if (rd) { rd_x(0, cur); rd_w(wf0, 0, cur); }
            if (live) { piece(0, nxt, kb); piece(1, nxt, kb); }
            mma_phase(I0{}, I0{}, wf0);
            // ---- phase 2: channels 32-63 x tokens 0-63
            if (rd) rd_w(wf1, 1, cur);
            if (live) { piece(2, nxt, kb); piece(3, nxt, kb); g_wait_vmcnt<4>(); } else { g_wait_vmcnt<0>(); }   // X-h1 of THIS tile has landed
            mma_phase(I1{}, I0{}, wf1);
            // ---- phase 3: channels 32-63 x tokens 64-127
            if (rd) rd_x(1, cur);
            if (live) { piece(4, nxt, kb); piece(5, nxt, kb); }
            mma_phase(I1{}, I1{}, wf1);
            // ---- phase 4: channels 0-31 x tokens 64-127 (both fragment sets are in registers)
            if (live) { piece(6, nxt, kb); piece(7, nxt, kb); g_wait_vmcnt<2>(); }     // X-h0, W-h0, W-h1 of the next tile have landed
            mma_phase(I0{}, I1{}, wf0);
        } else if constexpr (SCHED == 3) {
            // early requests, late waits: the whole next tile is requested in phases 1 and 2 (4 + 4 pieces) and every wait sits one phase
            // before the first read of what it waits for -- every piece has >= 3 phases (~0.8 us) between request and first use, where
            // SCHED 0 gives W-h1 one phase (~0.28 us: less than an L2 hit's latency under load).
            //   before phase 2 reads W-h1 of THIS tile (pieces 4, 5; requested in phase 2 of the previous tile): younger = 6, 7 + 0-3 of the next
            //   before phase 3 reads X-h1 of THIS tile (6, 7): younger = 0-7 of the next tile
            //   before the next phase 1 reads X-h0 / W-h0 of the next tile (0-3): younger = 4-7 of the next tile
            if (rd) { rd_x(0, cur); rd_w(wf0, 0, cur); }
            if (live) { piece(0, nxt, kb); piece(1, nxt, kb); piece(2, nxt, kb); piece(3, nxt, kb); g_wait_vmcnt<6>(); } else { g_wait_vmcnt<2>(); }
            mma_phase(I0{}, I0{}, wf0);
            if (rd) rd_w(wf1, 1, cur);
            if (live) { piece(4, nxt, kb); piece(5, nxt, kb); piece(6, nxt, kb); piece(7, nxt, kb); g_wait_vmcnt<8>(); } else { g_wait_vmcnt<0>(); }
            mma_phase(I1{}, I0{}, wf1);
            if (rd) rd_x(1, cur);
            mma_phase(I1{}, I1{}, wf1);
            if (live) g_wait_vmcnt<4>();
            mma_phase(I0{}, I1{}, wf0);
        } else if constexpr (SCHED == 2) {
            using LV = std::integral_constant<bool, live>;
            if (rd) { rd_x(0, cur); rd_w(wf0, 0, cur); }
            g_wait_vmcnt<2>();                                                            // W-h1 of this tile has landed (its X-h1 may be in flight)
            mma_phase_dma(I0{}, I0{}, wf0, LV{}, 0, nxt, kb);
            if (rd) rd_w(wf1, 1, cur);
            if (live) g_wait_vmcnt<2>(); else g_wait_vmcnt<0>();                          // X-h1 of this tile
            mma_phase_dma(I1{}, I0{}, wf1, LV{}, 2, nxt, kb);
            if (rd) rd_x(1, cur);
            mma_phase_dma(I1{}, I1{}, wf1, LV{}, 4, nxt, kb);
            if (live) g_wait_vmcnt<2>();                                                  // X-h0, W-h0 of the next tile
            mma_phase_dma(I0{}, I1{}, wf0, LV{}, 6, nxt, kb);
        } else {
            // pieces where the fragment reads are few: 0 / 3 / 1 / 4
            if (rd) { rd_x(0, cur); rd_w(wf0, 0, cur); }
            g_wait_vmcnt<2>();                                                            // W-h1 of this tile has landed (X-h1 may be in flight)
            mma_phase(I0{}, I0{}, wf0);
            if (rd) rd_w(wf1, 1, cur);
            if (live) { piece(0, nxt, kb); piece(1, nxt, kb); piece(2, nxt, kb); g_wait_vmcnt<3>(); } else { g_wait_vmcnt<0>(); }   // X-h1 of this tile
            mma_phase(I1{}, I0{}, wf1);
            if (rd) rd_x(1, cur);
            if (live) piece(3, nxt, kb);
            mma_phase(I1{}, I1{}, wf1);
            if (live) { piece(4, nxt, kb); piece(5, nxt, kb); piece(6, nxt, kb); piece(7, nxt, kb); g_wait_vmcnt<4>(); }   // X-h0, W-h0 of the next tile
            mma_phase(I0{}, I1{}, wf0);
        }
    };
    int t = 0;
    for (; t + 1 < nk; ++t) tile_body(t, std::true_type{});
    tile_body(t, std::false_type{});
    if (wr == 0) G_BAR();                              // re-align the two row groups: every fragment read is done

    if constexpr (PROF) t_prev = g_now();
    g_epilogue<T>(p, acc, smem, m0, n0, tid, wr, wc, fr, fg);
    if constexpr (PROF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lap(5);
        if (lane == 0 && blockIdx.x < kG256ProfWgs) {
            unsigned long long* slot = g_g256_prof[blockIdx.x][wave];
            for (int i = 0; i < 6; ++i) atomicAdd(&slot[i], pt[i]);
            atomicAdd(&slot[6], (unsigned long long)nk);
            atomicAdd(&slot[7], 1ull);
        }
    }
}


// "Free-running" form of the same 256 x 256 tile, same 8 waves (2 x 4, 128 tokens x 64 channels per wave), same DMA pieces, swizzle and
// two 64 KB K-tile buffers -- but ONE workgroup barrier per K tile instead of eight.  What round 4 measured about the 8-phase loop
// (profiles/r4_gemm256_ktile_trace.md): a barrier interval is the 16 MFMAs of one wave group (272 cycles of the SIMD's matrix pipe) plus
// ~60 cycles in which neither wave of the SIMD issues an MFMA -- the barrier's release, the lgkmcnt wait behind it and the first MFMA's
// issue -- eight times per K tile: 2680 cycles for 2176 of matrix work.  Here the barriers are not used as the pacing mechanism.  A wave
// software-pipelines itself: while the 32 MFMAs of one 32-deep K step issue (fragment set A), the 12 fragments of the next K step are read
// into set B, one read behind each of the first 12 MFMAs, and vice versa; the two waves of a SIMD drift against each other freely, so
// whenever one of them stalls (a DMA request costs its wave ~60 cycles of issue, a read burst its latency) the other one's MFMAs fill
// the pipe.  The one barrier sits in the middle of the K tile: by then every wave holds the tile's last fragments in registers, so the
// tile's buffer is free for tile t+2, and tile t+1 -- requested a whole tile earlier -- has landed (`vmcnt(0)`: the wave's 8 pieces, the
// barrier makes it everybody's).  Same MFMA instruction, same k order per accumulator (K step 0 then K step 1 of every tile): bit-identical
// to gemm256_kernel.
template <typename T, int PROF = 0, int OPT = 0, int LNF = 0>        // LNF: 0 plain, 1 folded LayerNorm with in-loop row statistics, 2 ... with the producer's partials
__global__ __launch_bounds__(512) void gemm256f_kernel(G256Dev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
    if constexpr (PROF) t_prev = g_now();
    auto lap = [&](int slot) {
        if constexpr (PROF) { const unsigned long long t = g_now(); pt[slot] += t - t_prev; t_prev = t; }
    };
    int tile_m, tile_n;
    {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        const int xm = xcd / p.gn, xn = xcd - xm * p.gn;
        const int m_lo = xm * p.tilesM / p.gm, m_hi = (xm + 1) * p.tilesM / p.gm;
        const int n_lo = xn * p.tilesN / p.gn, n_hi = (xn + 1) * p.tilesN / p.gn;
        const int nn = n_hi - n_lo;
        if (nn <= 0 || local >= (m_hi - m_lo) * nn) return;
        tile_m = m_lo + local / nn;
        tile_n = n_lo + local % nn;
    }
    const int m0 = tile_m * G_BM, n0 = tile_n * G_BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int rin = lane >> 3;
    const int csrc = (lane & 7) ^ rin;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const g_v4i rx = g_make_rsrc(p.x, p.x_bytes);
    const g_v4i rw = g_make_rsrc(p.w, p.w_bytes);

    // the wave's 8 DMA pieces per K tile: X rows {wave*32 + i*8 .. +8}, W rows likewise (i = 0..3) -- 8 rows x 128 B each
    unsigned xsrc[4], wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + rin;
        const int m = m0 + row, n = n0 + row;
        xsrc[i] = m < p.M ? (unsigned)(m * p.ldx + csrc * 8) * 2u : 0x80000000u;
        wsrc[i] = n < p.N ? (unsigned)(n * p.ldw + csrc * 8) * 2u : 0x80000000u;
    }
    const unsigned pdst = lds_base + (unsigned)(wave * 32) * 128u;
    auto piece = [&](int q, unsigned buf, unsigned kbyte) {              // q = 0..7: X pieces then W pieces of this wave
        if (q < 4) g_dma16(pdst + buf + (unsigned)q * 1024u, xsrc[q] + kbyte, rx);
        else g_dma16(pdst + buf + G_WOFF + (unsigned)(q - 4) * 1024u, wsrc[q - 4] + kbyte, rw);
    };

    g_f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (g_f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    // LNF: sum and sum of squares of the wave's 128 token rows over K, from the operand fragments the MFMAs consume anyway (fdot2 on packed halves: 8
    // VALU operations per fragment, 128 per K tile beside 64 MFMAs); the lane's share = its 8-element K chunks of rows j * 16 + fr
    float ln_su[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ln_sq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto ln_acc = [&](const uint4 (&xf)[8]) {
        if constexpr (LNF == 1 && std::is_same<T, f16>::value) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned w[4] = {xf[j].x, xf[j].y, xf[j].z, xf[j].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const h2 v = __builtin_bit_cast(h2, w[q]);
                    ln_su[j] = __builtin_amdgcn_fdot2(v, one, ln_su[j], false);
                    ln_sq[j] = __builtin_amdgcn_fdot2(v, v, ln_sq[j], false);
                }
            }
        }
    };
    // fragment f of a register set: f < 4 weight fragment i = f (A operand), else token fragment j = f - 4 (B operand)
    auto rd_frag = [&](uint4 (&wf)[4], uint4 (&xf)[8], int f, int ks, unsigned buf) {
        if (f < 4) {
            const int r = wc * 64 + f * 16 + fr;
            wf[f] = *reinterpret_cast<const uint4*>(smem + buf + G_WOFF + r * 128 + (((ks * 4 + fg) ^ (r & 7)) << 4));
        } else {
            const int r = wr * 128 + (f - 4) * 16 + fr;
            xf[f - 4] = *reinterpret_cast<const uint4*>(smem + buf + r * 128 + (((ks * 4 + fg) ^ (r & 7)) << 4));
        }
    };
    uint4 wfa[4], xfa[8], wfb[4], xfb[8];
    const int nk = p.K / 64;

    // prologue: K tiles 0 and 1 requested; tile 0 landed for everybody; K step 0 of tile 0 in register set A
#pragma unroll
    for (int q = 0; q < 8; ++q) piece(q, 0, 0);
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) piece(q, G_BUF, 128u);
        g_wait_vmcnt<8>();
    } else {
        g_wait_vmcnt<0>();
    }
    if constexpr (LNF == 2) {
        // row statistics of the tile's 256 token rows from the producer's partials (one thread per row, slices in order: deterministic), into the 4 KB
        // of LDS behind the two K-tile buffers -- while the first tiles' requests are in flight; every lane copies its 8 rows out before the epilogue
        // image claims those bytes
        if (tid < 256) {
            const int m = m0 + tid;
            float a = 0.f, q = 0.f;
            if (m < p.M) {
                const float2* pp = reinterpret_cast<const float2*>(p.ln_part) + (size_t)m * p.ln_part_P;
                for (int i = 0; i < p.ln_part_P; ++i) { const float2 v = pp[i]; a += v.x; q += v.y; }
            }
            const float inv_k = 1.0f / (float)p.K;
            const float mean = a * inv_k;
            const float var = q * inv_k - mean * mean;
            const float rstd = rsqrtf((var > 0.f ? var : 0.f) + p.ln_eps);
            *reinterpret_cast<float2*>(smem + 2 * G_BUF + tid * 8) = make_float2(mean, rstd);
            if (p.ln_stats_out && tile_n == 0 && m < p.M) *reinterpret_cast<float2*>(p.ln_stats_out + 2 * (size_t)m) = make_float2(mean, rstd);
        }
    }
    G_BAR();
#pragma unroll
    for (int f = 0; f < 12; ++f) rd_frag(wfa, xfa, f, 0, 0);
    lap(4);

    auto tile_body = [&](int t, auto MORE1, auto MORE2) {
        constexpr bool more1 = decltype(MORE1)::value, more2 = decltype(MORE2)::value;
        const unsigned cur = (t & 1) ? G_BUF : 0u, nxt = cur ^ G_BUF;
        // ---- first half: K step 0 (set A); behind each of the first 12 MFMAs one fragment of K step 1 -> set B
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                GMma<T>::run(acc[i][j], wfa[i], xfa[j]);
                const int m = i * 8 + j;
                if (m < 12) {
                    __builtin_amdgcn_sched_barrier(0);
                    rd_frag(wfb, xfb, m, 1, cur);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        ln_acc(xfa);
        lap(0);
        // this wave is past its last read of the tile's buffer; its pieces of tile t+1 (requested a tile ago) have landed
        // (sched_barrier: MFMAs are register-only, so hipcc would otherwise sink them below these waits)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        g_wait_vmcnt<0>();
        lap(1);
        G_BAR();
        lap(2);
        // ---- second half: K step 1 (set B); the 8 requests of tile t+2 go out first, one behind every second MFMA (they need the whole
        // next tile to land), then K step 0 of tile t+1 -> set A
        const unsigned kb2 = (unsigned)(t + 2) * 128u;
        if constexpr (OPT & 2) {
            if (more2) {
#pragma unroll
                for (int q = 0; q < 8; ++q) piece(q, cur, kb2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                GMma<T>::run(acc[i][j], wfb[i], xfb[j]);
                const int m = i * 8 + j;
                if (!(OPT & 2) && more2 && m < 16 && (m & 1) == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    piece(m >> 1, cur, kb2);
                    __builtin_amdgcn_sched_barrier(0);
                } else if (more1 && m >= 16 && m < 28) {
                    __builtin_amdgcn_sched_barrier(0);
                    rd_frag(wfa, xfa, m - 16, 0, nxt);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        ln_acc(xfb);
        lap(3);
    };
    if constexpr (OPT & 1) __builtin_amdgcn_s_setprio(1);      // over waves of OTHER kernels co-resident on the SIMD (a step runs three chains)
    {
        int t = 0;
        for (; t + 2 < nk; ++t) tile_body(t, std::true_type{}, std::true_type{});
        if (t + 1 < nk) { tile_body(t, std::true_type{}, std::false_type{}); ++t; }
        tile_body(t, std::false_type{}, std::false_type{});
    }
    if constexpr (OPT & 1) __builtin_amdgcn_s_setprio(0);
    __syncthreads();                                   // every wave is done with the operand buffers: the epilogue images reuse them
    if constexpr (PROF) t_prev = g_now();
    if constexpr (LNF == 2) {
        float mean[8], rstd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float2 st = *reinterpret_cast<const float2*>(smem + 2 * G_BUF + (wr * 128 + j * 16 + fr) * 8);
            mean[j] = st.x; rstd[j] = st.y;
        }
        __syncthreads();                               // every lane holds its rows' statistics: the image may overwrite the table
        g_epilogue<T, true>(p, acc, smem, m0, n0, tid, wr, wc, fr, fg, mean, rstd);
    } else if constexpr (LNF == 1) {
        float mean[8], rstd[8];
        const float inv_k = 1.0f / (float)p.K;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = ln_su[j], q = ln_sq[j];
            a += __shfl_xor(a, 16, 64); q += __shfl_xor(q, 16, 64);
            a += __shfl_xor(a, 32, 64); q += __shfl_xor(q, 32, 64);
            mean[j] = a * inv_k;
            const float var = q * inv_k - mean[j] * mean[j];
            rstd[j] = rsqrtf((var > 0.f ? var : 0.f) + p.ln_eps);
            const int m = m0 + wr * 128 + j * 16 + fr;
            if (p.ln_stats_out && wc == 0 && fg == 0 && tile_n == 0 && m < p.M) *reinterpret_cast<float2*>(p.ln_stats_out + 2 * (size_t)m) = make_float2(mean[j], rstd[j]);
        }
        g_epilogue<T, true>(p, acc, smem, m0, n0, tid, wr, wc, fr, fg, mean, rstd);
    } else {
        g_epilogue<T>(p, acc, smem, m0, n0, tid, wr, wc, fr, fg);
    }
    if constexpr (PROF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lap(5);
        if (lane == 0 && blockIdx.x < kG256ProfWgs) {
            unsigned long long* slot = g_g256_prof[blockIdx.x][wave];
            for (int i = 0; i < 6; ++i) atomicAdd(&slot[i], pt[i]);
            atomicAdd(&slot[6], (unsigned long long)nk);
            atomicAdd(&slot[7], 1ull);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Persistent form of gemm256f_kernel for tile grids of MORE than one tile per CU -- an EXPERIMENT of round 5 (the review's "give gemm256f_kernel
// something to overlap its store with"), compiled into `make DEV=1` builds only (HCM_GEMM256_PERSIST=1), bit-identical and SLOWER.
// The idea: gemm256f_kernel spends ~2.5 us of prologue (one HBM / L2 latency: the first K tile) per TILE with the matrix pipes of its CU idle, and the
// CU's next workgroup cannot start before this one has released its LDS.  Here a workgroup keeps its CU and walks its XCD's share of the tile grid:
// BEHIND the K loop of a tile -- both operand buffers are free then -- it requests K tiles 0 and 1 of its NEXT tile and only then runs the epilogue,
// so the next tile's first operand round trip and the store burst of this one overlap.  That needs the epilogue to stay out of LDS: the register
// epilogue (igemm_epilogue_regs' swap_pair form: 8 consecutive channels of one token per lane, one 16-byte store; the same f32 operations as
// g_epilogue).  Counted waits across the two kinds of vector-memory operation: a thread issues EXACTLY 16 epilogue stores (buffer stores; rows /
// channels past M / N present an offset beyond num_records and are dropped by the hardware instead of being branched around), so behind the epilogue
// the queue holds [8 + 8 requests of the next tile | 16 stores] and `vmcnt(16)` means "my requests have landed" (vmcnt retires in issue order).
// MEASURED (tools/gemm256_persist_bench.py, one box, fp16): FFN1 at 20480 rows (960 tiles, four per CU) 126.7 us against 122.3 for the one-tile
// workgroups, QKV 82.6 against 76.7, 10240 rows 61.1 / 61.3 and 50.1 / 46.0; configs[4] 8.95 against 8.86 ms per step (two interleaved runs each).
// Why: 122.3 us for four rounds is 30.6 us per round -- what ONE round costs as a launch of its own (31.4 us at 5120 rows).  The dispatcher already
// refills a CU the moment its workgroup exits, and inside a multi-round launch the rounds of different CUs drift apart, so a tile's prologue
// latency and store drain already overlap OTHER CUs' loops at the memory system; what the persistent form adds is the register epilogue's
// half-line stores and a request burst in front of them.  The per-tile cost is not idle-CU time that a longer-lived workgroup could reclaim.
#ifdef HCM_DEV_KNOBS
template <typename T, int OPT>
__global__ __launch_bounds__(512) void gemm256p_kernel(G256Dev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int xm = xcd / p.gn, xn = xcd - xm * p.gn;
    const int m_lo = xm * p.tilesM / p.gm, m_hi = (xm + 1) * p.tilesM / p.gm;
    const int n_lo = xn * p.tilesN / p.gn, n_hi = (xn + 1) * p.tilesN / p.gn;
    const int nn = n_hi - n_lo;
    const int ntl = nn > 0 ? (m_hi - m_lo) * nn : 0;
    if (slot >= ntl) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int rin = lane >> 3;
    const int csrc = (lane & 7) ^ rin;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const g_v4i rx = g_make_rsrc(p.x, p.x_bytes);
    const g_v4i rw = g_make_rsrc(p.w, p.w_bytes);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
    const unsigned pdst = lds_base + (unsigned)(wave * 32) * 128u;

    // the wave's 8 DMA pieces per K tile: X rows {wave*32 + i*8 .. +8}, W rows likewise (i = 0..3).  ONE offset register per operand (the kernel has
    // none to spare): piece i adds i * 8 rows; rows past M / N lie beyond the buffers' num_records by themselves (x_bytes <= M * ldx * 2), so the
    // hardware zero-fills them without the sentinel gemm256f_kernel selects per piece
    unsigned xsrc = 0, wsrc = 0;
    const unsigned xstep = (unsigned)(8 * p.ldx * 2), wstep = (unsigned)(8 * p.ldw * 2);
    int m0 = 0, n0 = 0;
    auto setup = [&](int local, unsigned& xs, unsigned& ws, int& mm, int& nb) {
        mm = (m_lo + local / nn) * G_BM;
        nb = (n_lo + local % nn) * G_BN;
        const int row = wave * 32 + rin;
        xs = (unsigned)((mm + row) * p.ldx + csrc * 8) * 2u;
        ws = (unsigned)((nb + row) * p.ldw + csrc * 8) * 2u;
    };
    auto piece_of = [&](unsigned xs, unsigned ws, int q, unsigned buf, unsigned kbyte) {
        if (q < 4) g_dma16(pdst + buf + (unsigned)q * 1024u, xs + (unsigned)q * xstep + kbyte, rx);
        else g_dma16(pdst + buf + G_WOFF + (unsigned)(q - 4) * 1024u, ws + (unsigned)(q - 4) * wstep + kbyte, rw);
    };
    const int nk = p.K / 64;                          // >= 2 (launcher)
    int local = slot;
    setup(local, xsrc, wsrc, m0, n0);
#pragma unroll
    for (int q = 0; q < 8; ++q) piece_of(xsrc, wsrc, q, 0, 0);
#pragma unroll
    for (int q = 0; q < 8; ++q) piece_of(xsrc, wsrc, q, G_BUF, 128u);
    bool first = true;
    for (;;) {
        g_f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (g_f32x4){0.f, 0.f, 0.f, 0.f};
        uint4 wfa[4], xfa[8], wfb[4], xfb[8];
        // (the lane id is made opaque once per output tile: the 24 fragment addresses below are invariant across the WHOLE kernel, and hoisted out of
        //  the tile loop they cost 43 spilled registers; hoisted out of the K loop only, they are what gemm256f_kernel carries)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int fr = lane_o & 15, fg = lane_o >> 4;
        auto rd_frag = [&](uint4 (&wf)[4], uint4 (&xf)[8], int f, int ks, unsigned buf) {
            if (f < 4) {
                const int r = wc * 64 + f * 16 + fr;
                wf[f] = *reinterpret_cast<const uint4*>(smem + buf + G_WOFF + r * 128 + (((ks * 4 + fg) ^ (r & 7)) << 4));
            } else {
                const int r = wr * 128 + (f - 4) * 16 + fr;
                xf[f - 4] = *reinterpret_cast<const uint4*>(smem + buf + r * 128 + (((ks * 4 + fg) ^ (r & 7)) << 4));
            }
        };
        // K tile 0 has landed: the first tile waits for everything but K tile 1's 8 requests; a later tile's 16 requests sit in FRONT of the previous
        // epilogue's 16 stores
        if (first) g_wait_vmcnt<8>(); else g_wait_vmcnt<16>();
        G_BAR();
#pragma unroll
        for (int f = 0; f < 12; ++f) rd_frag(wfa, xfa, f, 0, 0);

        auto tile_body = [&](int t, auto MORE1, auto MORE2) {
            constexpr bool more1 = decltype(MORE1)::value, more2 = decltype(MORE2)::value;
            const unsigned cur = (t & 1) ? G_BUF : 0u, nxt = cur ^ G_BUF;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    GMma<T>::run(acc[i][j], wfa[i], xfa[j]);
                    const int m = i * 8 + j;
                    if (m < 12) {
                        __builtin_amdgcn_sched_barrier(0);
                        rd_frag(wfb, xfb, m, 1, cur);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // K tile t + 1 has landed (this wave's pieces; the barrier makes it everybody's).  MID16: tile 0 of a later output tile -- the previous
            // epilogue's 16 stores may still be in flight behind those pieces
            if (!first && t == 0) g_wait_vmcnt<16>(); else g_wait_vmcnt<0>();      // (a scalar branch around the wait)
            G_BAR();
            const unsigned kb2 = (unsigned)(t + 2) * 128u;
            if constexpr (OPT & 2) {
                if (more2) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) piece_of(xsrc, wsrc, q, cur, kb2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    GMma<T>::run(acc[i][j], wfb[i], xfb[j]);
                    const int m = i * 8 + j;
                    if (!(OPT & 2) && more2 && m < 16 && (m & 1) == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        piece_of(xsrc, wsrc, m >> 1, cur, kb2);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (more1 && m >= 16 && m < 28) {
                        __builtin_amdgcn_sched_barrier(0);
                        rd_frag(wfa, xfa, m - 16, 0, nxt);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        };
        {
            int t = 0;
            for (; t + 2 < nk; ++t) tile_body(t, std::true_type{}, std::true_type{});
            if (t + 1 < nk) { tile_body(t, std::true_type{}, std::false_type{}); ++t; }
            tile_body(t, std::false_type{}, std::false_type{});
        }
        __syncthreads();                               // every wave is past its last fragment read: both operand buffers are free
        // ---- the next output tile's first two K tiles, requested in front of this tile's epilogue
        const int nlocal = local + nslots;
        const bool more = nlocal < ntl;
        int em0 = m0, en0 = n0;
        // (opaque to the optimiser: the epilogue's 16 store offsets depend only on these and the lane, and LICM would otherwise park them in
        //  registers across the K loop, which has none to spare: 43 spilled registers without this)
        asm volatile("" : "+s"(em0), "+s"(en0));
        if (more) {
            setup(nlocal, xsrc, wsrc, m0, n0);
#pragma unroll
            for (int q = 0; q < 8; ++q) piece_of(xsrc, wsrc, q, 0, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) piece_of(xsrc, wsrc, q, G_BUF, 128u);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- register epilogue: acc + bias, activation, one rounding; 16 buffer stores per thread, always issued
        {
            const int coff = (fg & 1) * 16 + (fg >> 1) * 8;
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                const int n = en0 + wc * 64 + ip * 32 + coff;
                const bool n_ok = n + 8 <= p.N;
                float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (p.bias && n_ok) {
                    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
                    bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v[8];
                    swap_pair(acc[2 * ip][j], acc[2 * ip + 1][j], v);
                    const int m = em0 + wr * 128 + j * 16 + fr;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bias8[e];
                    if (p.act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e]);
                    } else if (p.act == ACT_GELU) {
                        gelu_vec<T, 8>(v);
                    }
                    const uint4 o = pack_chunk<T>(v);
                    const unsigned off = (n_ok && m < p.M) ? (unsigned)(((size_t)m * p.ldy + n) * 2) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(g_u32x4, o), ry, (int)off, 0, 0);
                }
            }
        }
        if (!more) break;
        local = nlocal;
        first = false;
    }
}
#endif  // HCM_DEV_KNOBS (gemm256p_kernel)

hipError_t gemm256_prof_read(unsigned long long* host, bool reset) {
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_g256_prof), sizeof(unsigned long long) * kG256ProfWgs * 64);
    if (e != hipSuccess) return e;
    if (reset) {
        static unsigned long long zeros[kG256ProfWgs * 64] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_g256_prof), zeros, sizeof(zeros));
    }
    return e;
}

#ifdef HCM_DEV_KNOBS
// ------------------------------------------------------------------------------------------------------------------------------------
// Four-wave form of the same tile (one wave per SIMD, 128 x 128 outputs = 256 accumulator registers per wave, 2 x 2 waves) -- an EXPERIMENT,
// compiled into `make DEV=1` builds only (hcm_op_linear_impl variant 11 / HCM_GEMM256_W4=1), bit-identical to the 8-wave form and SLOWER: 1.55-1.87 us per K tile against 1.375.
// The idea: with 8 waves the LDS pipe moves 192 KB of fragment reads + 64 KB of DMA writes per K tile = 2048 cycles, exactly the tile's 2048
// MFMA cycles; a 128 x 128 wave tile reads (128 + 128) rows x 4 waves = 128 KB, 1536 LDS cycles with the DMA writes,
// which would leave the loop MFMA-bound.  With a single wave per SIMD nothing else hides a stall, so the loop is software-pipelined inside
// the wave: the fragments of k step s+1 are read (into the other register set) and the DMA pieces of K tile t+2 are requested from between
// the 64 MFMAs of k step s, one memory instruction after each MFMA; ONE workgroup barrier per K tile (in its middle: by then every wave holds
// tile t's last fragments in registers, so tile t's buffer is free for tile t+2, and tile t+1 -- requested a full tile earlier -- has landed).
// Why it loses: an LDS-DMA instruction costs the issuing wave ~60 cycles (MI355X_MICROARCH.md, cycle table), 16 pieces per wave per tile =
// ~960 cycles that the 8-wave form hides behind the SIMD's other wave and this one adds to its only wave's 2048 MFMA cycles.  A four-wave form
// would have to stage its operands through registers (buffer_load -> VGPR -> ds_write, 64 more registers on top of the 256 + 128 here).
template <typename T>
__global__ __launch_bounds__(256) void gemm256w_kernel(G256Dev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tile_m, tile_n;
    {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        const int xm = xcd / p.gn, xn = xcd - xm * p.gn;
        const int m_lo = xm * p.tilesM / p.gm, m_hi = (xm + 1) * p.tilesM / p.gm;
        const int n_lo = xn * p.tilesN / p.gn, n_hi = (xn + 1) * p.tilesN / p.gn;
        const int nn = n_hi - n_lo;
        if (nn <= 0 || local >= (m_hi - m_lo) * nn) return;
        tile_m = m_lo + local / nn;
        tile_n = n_lo + local % nn;
    }
    const int m0 = tile_m * G_BM, n0 = tile_n * G_BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;          // wave (wr, wc): tokens [wr*128, +128) x channels [wc*128, +128)
    const int rin = lane >> 3;
    const int csrc = (lane & 7) ^ rin;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const g_v4i rx = g_make_rsrc(p.x, p.x_bytes);
    const g_v4i rw = g_make_rsrc(p.w, p.w_bytes);

    // DMA pieces (8 rows x 128 B each): a K tile is 32 X pieces + 32 W pieces; wave w requests X pieces w*8 .. +8 and W pieces w*8 .. +8
    unsigned xsrc[8], wsrc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (wave * 8 + i) * 8 + rin;
        const int m = m0 + row, n = n0 + row;
        xsrc[i] = m < p.M ? (unsigned)(m * p.ldx + csrc * 8) * 2u : 0x80000000u;
        wsrc[i] = n < p.N ? (unsigned)(n * p.ldw + csrc * 8) * 2u : 0x80000000u;
    }
    const unsigned pdst = lds_base + (unsigned)(wave * 8) * 1024u;      // + i * 1024 (+ G_WOFF for W) + buffer
    auto piece = [&](int q, unsigned buf, unsigned kbyte) {              // q = 0..15: X pieces then W pieces of this wave
        if (q < 8) g_dma16(pdst + buf + (unsigned)q * 1024u, xsrc[q] + kbyte, rx);
        else g_dma16(pdst + buf + G_WOFF + (unsigned)(q - 8) * 1024u, wsrc[q - 8] + kbyte, rw);
    };

    g_f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (g_f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    // fragment f of a register set: f < 8 weight fragment i = f (A operand), else token fragment j = f - 8 (B operand)
    auto rd_frag = [&](uint4 (&wf)[8], uint4 (&xf)[8], int f, int ks, unsigned buf) {
        if (f < 8) {
            const int r = wc * 128 + f * 16 + fr;
            wf[f] = *reinterpret_cast<const uint4*>(smem + buf + G_WOFF + r * 128 + (((ks * 4 + fg) ^ (r & 7)) << 4));
        } else {
            const int r = wr * 128 + (f - 8) * 16 + fr;
            xf[f - 8] = *reinterpret_cast<const uint4*>(smem + buf + r * 128 + (((ks * 4 + fg) ^ (r & 7)) << 4));
        }
    };
    uint4 wfa[8], xfa[8], wfb[8], xfb[8];
    const int nk = p.K / 64;

    // prologue: K tiles 0 and 1 requested; tile 0 landed for everybody; k step 0 of tile 0 in register set A
#pragma unroll
    for (int q = 0; q < 16; ++q) piece(q, 0, 0);
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) piece(q, G_BUF, 128u);
        g_wait_vmcnt<16>();
    } else {
        g_wait_vmcnt<0>();
    }
    G_BAR();
#pragma unroll
    for (int f = 0; f < 16; ++f) rd_frag(wfa, xfa, f, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // one K tile; MORE1 / MORE2: a tile t+1 / t+2 exists (compile-time, so that the steady-state loop carries no branches between its MFMAs)
    auto tile_body = [&](int t, auto MORE1, auto MORE2) {
        constexpr bool more1 = decltype(MORE1)::value, more2 = decltype(MORE2)::value;
        const unsigned cur = (t & 1) ? G_BUF : 0u, nxt = cur ^ G_BUF;
        // ---- first half: MFMAs of k step 0 (set A); between them the fragments of k step 1 of this tile -> set B
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                GMma<T>::run(acc[i][j], wfa[i], xfa[j]);
                const int m = i * 8 + j;
                if (m < 16) {
                    __builtin_amdgcn_sched_barrier(0);
                    rd_frag(wfb, xfb, m, 1, cur);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
        // every wave is past its last read of this tile's buffer once its set-B fragments have arrived; tile t+1 (requested a tile ago) landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        g_wait_vmcnt<0>();
        G_BAR();
        // ---- second half: MFMAs of k step 1 (set B); between them k step 0 of tile t+1 -> set A and the requests of tile t+2 into this tile's buffer
        const unsigned kb2 = (unsigned)(t + 2) * 128u;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                GMma<T>::run(acc[i][j], wfb[i], xfb[j]);
                const int m = i * 8 + j;
                if (more1 && m < 16) {
                    __builtin_amdgcn_sched_barrier(0);
                    rd_frag(wfa, xfa, m, 0, nxt);
                    __builtin_amdgcn_sched_barrier(0);
                } else if (more2 && m >= 16 && m < 32) {
                    __builtin_amdgcn_sched_barrier(0);
                    piece(m - 16, cur, kb2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    {
        int t = 0;
        for (; t + 2 < nk; ++t) tile_body(t, std::true_type{}, std::true_type{});
        if (t + 1 < nk) { tile_body(t, std::true_type{}, std::false_type{}); ++t; }
        tile_body(t, std::false_type{}, std::false_type{});
    }
    __syncthreads();                                   // every wave is done with the operand buffers: the epilogue images reuse them

    if (p.res) {
        // residual epilogue (as in gemm256_kernel): f32 image of half the tile (the rows of one wave row), then bias + residual + activation
        constexpr int LDF = 260;
        float* sc = reinterpret_cast<float*>(smem);
        const int chunk = tid & 31, n = n0 + chunk * 8;
        float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
            bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
        }
        for (int half = 0; half < 2; ++half) {
            if (wr == half) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        *reinterpret_cast<float4*>(sc + (j * 16 + fr) * LDF + wc * 128 + i * 16 + fg * 4) =
                            make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
            __syncthreads();
#pragma unroll 4
            for (int pass = 0; pass < 16; ++pass) {
                const int r = pass * 8 + (tid >> 5);
                const int m = m0 + half * 128 + r;
                if (m < p.M && n < p.N) {
                    const uint4 rr = *reinterpret_cast<const uint4*>(p.res + ((size_t)m * p.ldr + n) * 2);
                    const float4 a0 = *reinterpret_cast<const float4*>(sc + r * LDF + chunk * 8), a1 = *reinterpret_cast<const float4*>(sc + r * LDF + chunk * 8 + 4);
                    float v[8] = {a0.x + bias8[0], a0.y + bias8[1], a0.z + bias8[2], a0.w + bias8[3], a1.x + bias8[4], a1.y + bias8[5], a1.z + bias8[6], a1.w + bias8[7]};
                    float r8[8];
                    cvt_chunk<T>(rr, r8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r8[e];
                    if (p.act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e]);
                    } else if (p.act == ACT_GELU) {
                        gelu_vec<T, 8>(v);
                    }
                    st_chunk(reinterpret_cast<T*>(p.y + ((size_t)m * p.ldy + n) * 2), v);
                }
            }
            __syncthreads();
        }
        return;
    }
    // 16-bit tile image in two halves of the wave's token fragments (0-3 / 4-7: tile rows {0-63, 128-191} / {64-127, 192-255})
    {
        float4 b4[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = n0 + wc * 128 + i * 16 + fg * 4;
            b4[i] = (p.bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int chunk = tid & 31;
        const int n = n0 + chunk * 8;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = half * 4 + jj;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v[4] = {acc[i][j][0] + b4[i].x, acc[i][j][1] + b4[i].y, acc[i][j][2] + b4[i].z, acc[i][j][3] + b4[i].w};
                    if (p.act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = relu_f(v[e]);
                    } else if (p.act == ACT_GELU) {
                        gelu_vec<T, 4>(v);
                    }
                    T o4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) Tr<T>::st(&o4[e], v[e]);
                    const int r = wr * 128 + j * 16 + fr;
                    const int cb = (wc * 128 + i * 16 + fg * 4) * 2;
                    *reinterpret_cast<uint2*>(smem + r * G_IMG_LD + cb) = *reinterpret_cast<const uint2*>(o4);
                }
            }
            __syncthreads();
            if (n < p.N) {
#pragma unroll 4
                for (int pass = 0; pass < 16; ++pass) {
                    const int rr = pass * 8 + (tid >> 5);
                    const int r = (rr >> 6) * 128 + half * 64 + (rr & 63);
                    const int m = m0 + r;
                    if (m < p.M)
                        *reinterpret_cast<uint4*>(p.y + ((size_t)m * p.ldy + n) * 2) = *reinterpret_cast<const uint4*>(smem + r * G_IMG_LD + chunk * 16);
                }
            }
        }
    }
}

#endif  // HCM_DEV_KNOBS

// layout / range constraints only (a forced launch, IGemm::force_gemm256: the folded-LayerNorm GEMMs must run on this kernel at every batch size so that a
// row's result does not depend on the batch it is computed in)
static bool gemm256_valid(const IGemm& g, int dt) {
    if (dt != DT_BF16 && dt != DT_F16) return false;
    if (g.KH != 1 || g.KW != 1 || g.stride != 1 || g.pad != 0 || g.H != 1 || g.W != 1) return false;
    if (g.out_f32 || g.res_f32 || g.groups > 1 || g.gn_gamma || g.cs_part || g.hpool || g.x_src_dt >= 0 || g.gi_stats) return false;
    const int Kp = g.Kp ? g.Kp : g.K, ldx = g.xC ? g.xC : g.Cin, ldy = g.ldy ? g.ldy : g.N, ldr = g.ldr ? g.ldr : g.N;
    if (g.K % 64 || Kp % 8 || ldx % 8 || ldy % 8 || g.N % 8 || (g.res && ldr % 8) || g.K < 128) return false;
    if ((size_t)g.M * ldx * 2 >= 0x7FFFFFF0ull || (size_t)g.N * Kp * 2 >= 0x7FFFFFF0ull) return false;
    return g.M >= 1 && g.N >= 8;
}
bool gemm256_applicable(const IGemm& g, int dt) {
    if (dt != DT_BF16 && dt != DT_F16) return false;
    if (g.KH != 1 || g.KW != 1 || g.stride != 1 || g.pad != 0 || g.H != 1 || g.W != 1) return false;       // plain row-major GEMM
    if (g.out_f32 || g.res_f32 || g.groups > 1 || g.gn_gamma || g.cs_part || g.hpool || g.x_src_dt >= 0 || g.gi_stats) return false;
    const int Kp = g.Kp ? g.Kp : g.K, ldx = g.xC ? g.xC : g.Cin, ldy = g.ldy ? g.ldy : g.N, ldr = g.ldr ? g.ldr : g.N;
    if (g.K % 64 || Kp % 8 || ldx % 8 || ldy % 8 || g.N % 8 || (g.res && ldr % 8)) return false;
    if ((size_t)g.M * ldx * 2 >= 0x7FFFFFF0ull || (size_t)g.N * Kp * 2 >= 0x7FFFFFF0ull) return false;   // offsets + the out-of-range sentinel
    // worth it when the tile grid fills a good part of the chip with whole tiles: wide outputs over many rows
    const long tiles = (long)((g.M + G_BM - 1) / G_BM) * ((g.N + G_BN - 1) / G_BN);
    // (HCM_GEMM256_MIN_TILES / HCM_GEMM256_MIN_N: A/B knobs for the in-step choice of the narrow-output layers)
    static const int min_tiles = dev_env("HCM_GEMM256_MIN_TILES") ? atoi(dev_env("HCM_GEMM256_MIN_TILES")) : 96;
    static const int min_n = dev_env("HCM_GEMM256_MIN_N") ? atoi(dev_env("HCM_GEMM256_MIN_N")) : 512;
    return g.N >= min_n && g.M >= 2048 && tiles >= min_tiles && g.K >= 256;
}

hipError_t launch_gemm256(const IGemm& g, int dt, hipStream_t s) {
    if (g.force_gemm256 ? !gemm256_valid(g, dt) : !gemm256_applicable(g, dt)) return hipErrorInvalidValue;
    G256Dev d;
    d.x = (const char*)g.x; d.w = (const char*)g.w; d.bias = g.bias; d.y = (char*)g.y;
    d.res = (const char*)g.res; d.ldr = g.ldr ? g.ldr : g.N;
    d.M = g.M; d.N = g.N; d.K = g.K; d.ldx = g.xC ? g.xC : g.Cin; d.ldw = g.Kp ? g.Kp : g.K; d.ldy = g.ldy ? g.ldy : g.N; d.act = g.act;
    d.ln_s = g.ln_s; d.ln_t = g.ln_t; d.ln_stats_out = g.ln_stats_out; d.ln_eps = g.ln_eps; d.ln_part = g.ln_part_in; d.ln_part_P = g.ln_part_P;
    if (g.ln_part_in && (!g.ln_s || g.ln_part_P < 1)) return hipErrorInvalidValue;
    if (g.ln_s && (dt != DT_F16 || !g.ln_t || g.res)) return hipErrorInvalidValue;      // folded LayerNorm: fp16, no residual
    d.x_bytes = (unsigned)(((size_t)(g.M - 1) * d.ldx + g.K) * 2);
    d.w_bytes = (unsigned)((size_t)g.N * d.ldw * 2);
    d.tilesM = (g.M + G_BM - 1) / G_BM;
    d.tilesN = (g.N + G_BN - 1) / G_BN;
    // XCD grid: the factorisation of 8 with the fewest workgroups on the busiest XCD, then the smallest per-XCD operand footprint
    int best_cnt = 1 << 30, best_fp = 1 << 30;
    d.gm = 8; d.gn = 1;
    for (int gm = 1; gm <= 8; gm *= 2) {
        const int gn = 8 / gm;
        int cnt = 0, fp = 0;
        for (int x = 0; x < 8; ++x) {
            const int xm = x / gn, xn = x % gn;
            const int nm = (xm + 1) * d.tilesM / gm - xm * d.tilesM / gm, nn = (xn + 1) * d.tilesN / gn - xn * d.tilesN / gn;
            if (nm * nn > cnt) cnt = nm * nn;
            if (nm + nn > fp) fp = nm + nn;
        }
        if (cnt < best_cnt || (cnt == best_cnt && fp < best_fp)) { best_cnt = cnt; best_fp = fp; d.gm = gm; d.gn = gn; }
    }
    // round 4: the one-barrier-per-K-tile form (gemm256f_kernel, bit-identical to the 8-phase gemm256_kernel); its 8 DMA requests go out right behind
    // the barrier for short K (the launch is half prologue + epilogue there, profiles/r4_gemm256_ktile_trace.md), interleaved with the MFMAs beyond.
    // HCM_GEMM256_8PHASE=1 (development build): the 8-phase form of rounds 2-3, for the A/B.
    static const bool eight_phase = dev_env("HCM_GEMM256_8PHASE") != nullptr;
    const void* fn;
    if (eight_phase) fn = dt == DT_BF16 ? reinterpret_cast<const void*>(gemm256_kernel<bf16>) : reinterpret_cast<const void*>(gemm256_kernel<f16>);
    else if (g.K <= 1024) fn = dt == DT_BF16 ? reinterpret_cast<const void*>(gemm256f_kernel<bf16, 0, 2>) : reinterpret_cast<const void*>(gemm256f_kernel<f16, 0, 2>);
    else fn = dt == DT_BF16 ? reinterpret_cast<const void*>(gemm256f_kernel<bf16, 0, 0>) : reinterpret_cast<const void*>(gemm256f_kernel<f16, 0, 0>);
    if (g.ln_s && g.ln_part_in) fn = g.K <= 1024 ? reinterpret_cast<const void*>(gemm256f_kernel<f16, 0, 2, 2>) : reinterpret_cast<const void*>(gemm256f_kernel<f16, 0, 0, 2>);
    else if (g.ln_s) fn = g.K <= 1024 ? reinterpret_cast<const void*>(gemm256f_kernel<f16, 0, 2, 1>) : reinterpret_cast<const void*>(gemm256f_kernel<f16, 0, 0, 1>);
    int threads = 512;
#ifdef HCM_DEV_KNOBS
    // `make DEV=1` builds only (libhcm_dev.so): g.impl >> 4 selects the experiment builds DESIGN.md section 7 quotes (f16): 1 = SCHED 1;
    // 2/3/4 = SCHED 0 with parts removed (DBG 1/2/4: timing only, results are wrong); 5-8 DBG combinations; 9 = SCHED 3 (early requests,
    // bit-identical); 10 = SCHED 3 without MFMAs; 11 = the four-wave form (bit-identical).  HCM_GEMM256_W4=1: the four-wave form everywhere.
    const int var = g.impl >> 4;
    if (dt == DT_F16 && var) {
        switch (var) {
            case 1: fn = reinterpret_cast<const void*>(gemm256_kernel<f16, 1, 0>); break;
            case 2: fn = reinterpret_cast<const void*>(gemm256_kernel<f16, 0, 1>); break;
            case 3: fn = reinterpret_cast<const void*>(gemm256_kernel<f16, 0, 2>); break;
            case 4: fn = reinterpret_cast<const void*>(gemm256_kernel<f16, 0, 4>); break;
            case 5: fn = reinterpret_cast<const void*>(gemm256_kernel<f16, 0, 5>); break;
            case 6: fn = reinterpret_cast<const void*>(gemm256_kernel<f16, 0, 6>); break;
            case 7: fn = reinterpret_cast<const void*>(gemm256_kernel<f16, 0, 7>); break;
            case 8: fn = reinterpret_cast<const void*>(gemm256_kernel<f16, 0, 3>); break;
            case 9: fn = reinterpret_cast<const void*>(gemm256_kernel<f16, 3, 0>); break;
            case 10: fn = reinterpret_cast<const void*>(gemm256_kernel<f16, 3, 4>); break;
            case 11: break;                                                           // the four-wave form, selected below
            case 12: fn = reinterpret_cast<const void*>(gemm256f_kernel<f16, 0>); break;   // free-running: one barrier per K tile (bit-identical)
            case 13: fn = reinterpret_cast<const void*>(gemm256_kernel<f16, 0, 8>); break;  // 8-phase with phase stamps
            case 14: fn = reinterpret_cast<const void*>(gemm256_kernel<f16, 0, 16>); break; // ... plus a stamp in front of barrier 1
            case 15: fn = reinterpret_cast<const void*>(gemm256f_kernel<f16, 1>); break;   // free-running with phase stamps
            case 16: fn = reinterpret_cast<const void*>(gemm256_kernel<f16, 0, 0>); break;  // the 8-phase form whatever the default is
            case 17: fn = reinterpret_cast<const void*>(gemm256f_kernel<f16, 0, 1>); break; // free-running + static s_setprio(1) over the loop
            case 18: fn = reinterpret_cast<const void*>(gemm256f_kernel<f16, 0, 2>); break; // free-running, the 8 DMA requests right behind the barrier
            case 19: fn = reinterpret_cast<const void*>(gemm256f_kernel<f16, 0, 3>); break; // both
            case 20: fn = reinterpret_cast<const void*>(gemm256f_kernel<f16, 1, 2>); break; // DMA-first form with phase stamps
            default: return hipErrorInvalidValue;
        }
    }
    static const bool free_default = dev_env("HCM_GEMM256_FREE") != nullptr;           // in-step A/B of the free-running form
    if (free_default && !var) {
        const int fo = atoi(dev_env("HCM_GEMM256_FREE"));         // 1 plain, 2 + static priority, 3 DMA requests first, 4 both
        if (dt == DT_BF16) fn = reinterpret_cast<const void*>(gemm256f_kernel<bf16, 0>);
        else fn = fo == 2 ? reinterpret_cast<const void*>(gemm256f_kernel<f16, 0, 1>) : fo == 3 ? reinterpret_cast<const void*>(gemm256f_kernel<f16, 0, 2>)
                : fo == 4 ? reinterpret_cast<const void*>(gemm256f_kernel<f16, 0, 3>) : reinterpret_cast<const void*>(gemm256f_kernel<f16, 0>);
    }
    static const bool w4_default = dev_env("HCM_GEMM256_W4") != nullptr;
    if (var == 11 || (w4_default && !var)) {
        fn = dt == DT_BF16 ? reinterpret_cast<const void*>(gemm256w_kernel<bf16>) : reinterpret_cast<const void*>(gemm256w_kernel<f16>);
        threads = 256;
    }
#else
    if (g.impl >> 4) return hipErrorInvalidValue;          // experiment variants exist in `make DEV=1` builds only
#endif
#ifdef HCM_DEV_KNOBS
    // round-5 experiment (development build, HCM_GEMM256_PERSIST=1): more than one tile per CU on the busiest XCD -> the persistent form
    // (gemm256p_kernel; bit-identical, measured SLOWER: see its header).  Plain layers only.
    static const bool persist = dev_env("HCM_GEMM256_PERSIST") != nullptr && atoi(dev_env("HCM_GEMM256_PERSIST")) != 0;
    const size_t y_bytes = ((size_t)(g.M - 1) * d.ldy + g.N) * 2;
    if (persist && !eight_phase && !g.res && !g.ln_s && (g.impl >> 4) == 0 && threads == 512 && best_cnt > 32 && g.K >= 128 && y_bytes < 0x7FFFFFF0ull) {
        d.y_bytes = (unsigned)y_bytes;
        const void* pf = g.K <= 1024 ? (dt == DT_BF16 ? reinterpret_cast<const void*>(gemm256p_kernel<bf16, 2>) : reinterpret_cast<const void*>(gemm256p_kernel<f16, 2>))
                                     : (dt == DT_BF16 ? reinterpret_cast<const void*>(gemm256p_kernel<bf16, 0>) : reinterpret_cast<const void*>(gemm256p_kernel<f16, 0>));
        hipError_t e = hipFuncSetAttribute(pf, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        void* pargs[] = {&d};
        return hipLaunchKernel(pf, dim3(8 * 32), dim3(512), pargs, 2 * G_BUF, s);
    }
#endif
    {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    void* args[] = {&d};
    return hipLaunchKernel(fn, dim3(8 * best_cnt), dim3(threads), args, G_LDS, s);
}

}  // namespace hcm
