// Implicit-GEMM convolution / linear layer on MFMA (gfx950).
//
//   D[n][m] = sum_k W[n][k] * A[m][k]     W = weights [N][Kp], A = im2col(x) gathered on the fly (NHWC)
//
// The weights are the MFMA "A" operand and the activations the "B" operand, so that in the 16x16
// accumulator tile (col = lane&15, row = (lane>>4)*4 + reg) a lane holds FOUR CONSECUTIVE output
// channels of ONE output pixel: the epilogue (bias + residual + activation) then stores 8 B (bf16) or
// 16 B (f32) per lane into the NHWC output with no transpose.
//
// Tile: BM output pixels x BN output channels x 128 bytes of K per step (64 bf16/f16, 32 f32); 4 waves (2x2) or 8 waves
// (2x4, 4x2 for the 256x128 tile).  Two kernels share the tile math and the epilogue:
//   igemm_dma_kernel  (every conv / linear): both operand tiles travel L2/HBM -> LDS with bounds-checked
//                     `buffer_load_dwordx4 ... lds` issued from inline asm (zero fill for the im2col padding comes from the
//                     hardware range check), 2-buffer or 3-deep ring, counted vmcnt waits, optionally with the DMA issue
//                     interleaved into the MFMA stream (ILV) and a rotated loop; see the comment above the kernel;
//   igemm_kernel      register-staged (global -> VGPR -> LDS); its narrow-channel form gathers first layers (Cin = 1 or 3)
//                     element-wise from the raw frame (fp32 path, depth stems, SimpleCNN).
// LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row & 7), which makes the staging writes and the
// ds_read_b128 fragment reads bank-conflict free (cdna_hip_programming.md T2; 0 conflicts measured).
//
// K ordering inside one MFMA step is permuted consistently for both operands (lane group g consumes the
// g-th 16-byte chunk of the 64-byte half row), which a dot product is invariant to; this lets the f32
// path feed four v_mfma_f32_16x16x4_f32 from one ds_read_b128.
//
// Reference ops replaced: cuDNN conv fwd + BatchNorm(eval) + ReLU of torchvision resnet50, the GN-ResNet
// convs, and every nn.Linear / Conv1d(k=1) on the path.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#include <algorithm>
#include <mutex>
#include <string>
#include <unordered_map>
#include "kernels.h"
#include "dev.h"

namespace hcm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct IGemmDev {
    const char* x; const char* w; const float* bias; const char* res; char* y;
    int B, H, W, Cin, xC, Ho, Wo, KH, KW, stride, pad;
    int stride_w;                  // horizontal stride (== stride except for the packed-frame stem, see launch_pack_frame)
    int M, N, K, Kp, ldy, ldr, act, out_f32;
    int image_epi;                 // force the LDS-image epilogue where the register epilogue would apply (HCM_IGEMM_IMAGE=1: A/B, toggle test)
    int res_f32;                   // the residual is an f32 tensor [M][ldr] (16-bit kernels: the f32 residual stream of the bf16 BERT)
    int cin_shift, kw_rcp, tilesM, tilesN, map;
    unsigned x_bytes, w_bytes;     // extents for the bounds-checked buffer loads of the DMA variant
    float x_scale;                 // narrow-channel first-layer gather: value = src * x_scale
    int rowrun;                    // RGB f32 stem: K laid out as KH runs of 24 (see igemm_kernel)
    // grouped launch (blockIdx.y = group): element offsets added to x / w / bias / y+res per group
    int groups; long long g_x, g_w, g_b, g_y;
    // fused GroupNorm epilogue (small maps: a 64-row tile holds whole samples): y = GN(conv) * gamma + beta (+ res) (ReLU)
    const float* gn_gamma; const float* gn_beta; int gn_cg, gn_hw; float gn_eps;
    int hpool;         // horizontal half of MaxPool2d(3, 2, 1) in the epilogue (igemm_epilogue_hpool)
    // GroupNorm statistics of the conv output from the f32 tile image: per (sample, 64-row block, group) sum / sum of squares into
    // cs_part[((b * cs_hw / 64 + block) * cs_G + group) * 2] (the stand-alone statistics launch and its read of the map disappear)
    float* cs_part; int cs_cg, cs_hw, cs_G;
    // GroupNorm on load (IGemm::gi_*, igemm_gnin_kernel)
    const float* gi_stats; const float* gi_gamma; const float* gi_beta; int gi_ps, gi_cg, gi_G, gi_hw, gi_relu; float gi_eps;
    const char* gi_res; char* gi_out;
    // residual = LayerNorm(res) on the fly (IGemm::rln_*)
    const float* rln_stats; const float* rln_gamma; const float* rln_beta;
    float* ln_part; int ln_part_P;       // IGemm::ln_part_out: (sum, sum of squares) of every stored row over this wave column's 32 channels
};
// the residual chunk rr (8 channels n .. n + 7 of row m, read from the pre-LayerNorm tensor) -> its LayerNorm
__device__ __forceinline__ void rln_apply(const IGemmDev& p, int m, int n, float (&rr)[8]) {
    const float2 st = *reinterpret_cast<const float2*>(p.rln_stats + 2 * (size_t)m);
    const float4 g0 = *reinterpret_cast<const float4*>(p.rln_gamma + n), g1 = *reinterpret_cast<const float4*>(p.rln_gamma + n + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(p.rln_beta + n), b1 = *reinterpret_cast<const float4*>(p.rln_beta + n + 4);
    const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) rr[e] = (rr[e] - st.x) * st.y * ga[e] + be[e];
}

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<f16> {
    static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

// XCD-aware tile order (block b runs on XCD b % 8, each XCD has a private 4 MB L2).
//   map 0: the XCD owns whole pixel-tiles (tile_m = 8j + xcd) and walks all channel tiles of one before the next: the
//          gathered activation rows are re-read from that XCD's L2, the (small) weight matrix is resident everywhere.
//   map 1: for big weight matrices (> ~2 MB: the BERT GEMMs) the XCD owns a contiguous slice of the CHANNEL tiles
//          instead, so its slice of the weights stays L2-resident while the activations stream through once per XCD.
__device__ __forceinline__ bool tile_of_block(const IGemmDev& p, int bid, int& tile_m, int& tile_n) {
    const int xcd = bid & 7;
    const int local = bid >> 3;
    if (p.map == 0) {
        tile_n = local % p.tilesN;
        tile_m = (local / p.tilesN) * 8 + xcd;
        return tile_m < p.tilesM;
    }
    const int n_lo = (xcd * p.tilesN) >> 3, n_hi = ((xcd + 1) * p.tilesN) >> 3;   // this XCD's channel tiles
    const int nn = n_hi - n_lo;
    if (nn <= 0) return false;
    tile_m = local / nn;
    tile_n = n_lo + local % nn;
    return tile_m < p.tilesM;
}

// Shared epilogue of both kernel variants (see the comment at its top).
template <typename T, int BM, int BN, int NW = 4, int WMc = 2, int NPRE = 1>
__device__ __forceinline__ void igemm_epilogue(const IGemmDev& p, f32x4 (&acc)[BN / (NW / WMc) / 16][BM / WMc / 16], char* smem, int m0, int n0,
                                               int tid, int wm, int wn, int fr, int fg, const uint4 (&rpre)[NPRE], bool have_pre) {
    constexpr int WNc = NW / WMc;          // waves along the channel axis (WMc along the pixel axis)
    constexpr int TM = BM / WMc / 16;
    constexpr int TN = BN / WNc / 16;
    // ---- epilogue ----
    // Phase 1: every lane parks its accumulators (4 consecutive channels of one pixel) in an f32 LDS image of the
    // output tile (the A/B tiles are dead: the K loop ended with a barrier).  Phase 2: each thread takes 8
    // consecutive channels of a row, applies bias + residual + activation in f32, rounds once, and stores 16 B --
    // a row of the tile leaves as one contiguous BN*sizeof(T)-byte run (the accumulator layout alone would store
    // 32-byte fragments).  Row stride BN+4 floats keeps the ds_write_b128 of phase 1 conflict free.
    constexpr int LDC = BN + 4;
    float* sc = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int r = wm * (BM / WMc) + j * 16 + fr;
            const int cc = wn * (BN / WNc) + i * 16 + fg * 4;
            *reinterpret_cast<float4*>(sc + r * LDC + cc) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    __syncthreads();
    // Fused GroupNorm (habitat GN-ResNet layers whose map has <= 64 pixels): the tile holds BM / hw whole samples and BN / cg
    // whole groups; one wave per (sample, group) reduces its hw x cg block of the f32 image (fixed order), then the
    // normalisation rides in phase 2.  The un-normalised conv output is never written.
    float* gst = sc + BM * LDC;            // [sample][group][mean, rstd]
    const int gn_ng = p.gn_cg ? BN / p.gn_cg : 0;
    if (p.gn_cg) {
        const int hw = p.gn_hw, cg = p.gn_cg;
        const int pairs = (BM / hw) * gn_ng;
        const int lane_ = tid & 63;
        for (int pr = tid >> 6; pr < pairs; pr += NW) {
            const int sidx = pr / gn_ng, g = pr - sidx * gn_ng;
            const int ne = hw * cg;
            float a = 0.f, q = 0.f;
            for (int e = lane_; e < ne; e += 64) {
                const int r = sidx * hw + e / cg, cix = g * cg + e % cg;
                const float v = sc[r * LDC + cix];
                a += v; q += v * v;
            }
            a = wave_sum(a); q = wave_sum(q);
            if (lane_ == 0) {
                const float inv = 1.0f / (float)ne;
                const float mean = a * inv;
                gst[pr * 2] = mean;
                gst[pr * 2 + 1] = rsqrtf(relu_f(q * inv - mean * mean) + p.gn_eps);
            }
        }
        __syncthreads();
    }
    if (p.cs_part) {
        // column sums of the image in PARTS row slices (one thread per (slice, column), fixed order), then per 64-row block and group
        constexpr int PARTS = 64 * NW / BN, RP = BM / PARTS, PPB = 64 / RP;      // slices; rows per slice; slices per 64-row block
        static_assert(RP >= 8 && RP <= 64 && 64 % RP == 0, "column-sum slices");
        float* red = sc + BM * LDC + 256;
        // The summation order is the same for every tile shape (so a sample's statistics do not depend on the batch it runs in): rows in
        // aligned runs of 8, left to right, then a balanced binary tree over the runs of a 64-row block, then the group's columns in order.
        static_assert(RP % 8 == 0, "column-sum slices are whole 8-row runs");
        {
            const int col = tid % BN, part = tid / BN;
            float a8[RP / 8], q8[RP / 8];
#pragma unroll
            for (int u = 0; u < RP / 8; ++u) {
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = part * RP + u * 8 + rr;
                    const float v = (m0 + r < p.M) ? sc[r * LDC + col] : 0.f;
                    a += v; q += v * v;
                }
                a8[u] = a; q8[u] = q;
            }
#pragma unroll
            for (int st = 1; st < RP / 8; st *= 2)
#pragma unroll
                for (int u = 0; u < RP / 8; u += 2 * st) { a8[u] += a8[u + st]; q8[u] += q8[u + st]; }
            red[(part * BN + col) * 2] = a8[0]; red[(part * BN + col) * 2 + 1] = q8[0];
        }
        __syncthreads();
        const int cg = p.cs_cg, ng = BN / cg;
        if (tid < (BM / 64) * ng) {
            const int blk = tid / ng, gl = tid - blk * ng;
            const int col0 = gl * cg;
            const int m = m0 + blk * 64;
            if (n0 + col0 < p.N && m < p.M) {
                float a = 0.f, q = 0.f;
                for (int j = 0; j < cg; ++j) {
                    float at[PPB], qt[PPB];
#pragma unroll
                    for (int pp = 0; pp < PPB; ++pp) {
                        const float* e = red + (((blk * PPB + pp) * BN) + col0 + j) * 2;
                        at[pp] = e[0]; qt[pp] = e[1];
                    }
#pragma unroll
                    for (int st = 1; st < PPB; st *= 2)
#pragma unroll
                        for (int u = 0; u < PPB; u += 2 * st) { at[u] += at[u + st]; qt[u] += qt[u + st]; }
                    a += at[0]; q += qt[0];
                }
                const int smp = m / p.cs_hw, pblk = (m - smp * p.cs_hw) >> 6;
                const int grp = ((int)blockIdx.y * p.N + n0 + col0) / cg;
                float* o = p.cs_part + (((size_t)smp * (p.cs_hw >> 6) + pblk) * p.cs_G + grp) * 2;
                o[0] = a; o[1] = q;
            }
        }
    }
    constexpr int TPR = BN / 8;            // threads per tile row
    constexpr int RPP = 64 * NW / TPR;     // rows per pass
    const int c8 = (tid % TPR) * 8;
    const int n = n0 + c8;
    if (n >= p.N) return;
    const bool hi_ok = (n + 4) < p.N;      // N % 4 == 0: the second group of four is all-valid or all-invalid
    float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n);
        bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
        if (hi_ok) {
            const float4 b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
            bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
        }
    }
    const bool wide16 = sizeof(T) == 2 && !p.out_f32 && hi_ok && (p.ldy % 8 == 0);
    const bool wide16r = sizeof(T) == 2 && hi_ok && (p.ldr % 8 == 0);
    // The bias and the prefetched identity chunks must be IN registers before the pass loop.  Left to be waited for inside the
    // conditionally executed passes, every pass gets an `s_waitcnt vmcnt(0)` (the skipped-pass path has nothing younger in
    // flight, so the merged wait is 0) -- which also drains the previous pass's stores: four serialized HBM round trips per tile.
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(bias8[e]));
    if (have_pre) {
#pragma unroll
        for (int pass = 0; pass < NPRE; ++pass) asm volatile("" ::"v"(rpre[pass].x));
    }
#pragma unroll
    for (int pass = 0; pass < BM / RPP; ++pass) {
        const int r = pass * RPP + tid / TPR;
        const int m = m0 + r;
        if (m >= p.M) continue;
        float v[8];
        {
            const float4 a0 = *reinterpret_cast<const float4*>(sc + r * LDC + c8);
            const float4 a1 = *reinterpret_cast<const float4*>(sc + r * LDC + c8 + 4);
            v[0] = a0.x + bias8[0]; v[1] = a0.y + bias8[1]; v[2] = a0.z + bias8[2]; v[3] = a0.w + bias8[3];
            v[4] = a1.x + bias8[4]; v[5] = a1.y + bias8[5]; v[6] = a1.z + bias8[6]; v[7] = a1.w + bias8[7];
        }
        if (p.gn_cg) {                     // cg is a multiple of 8: the thread's 8 channels share one group
            const float* ms = gst + ((r / p.gn_hw) * gn_ng + c8 / p.gn_cg) * 2;
            const float mean = ms[0], rstd = ms[1];
            const float4 g0 = *reinterpret_cast<const float4*>(p.gn_gamma + n), g1 = *reinterpret_cast<const float4*>(p.gn_gamma + n + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(p.gn_beta + n), b1 = *reinterpret_cast<const float4*>(p.gn_beta + n + 4);
            const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean) * rstd * ga[e] + be[e];
        }
        if (p.res && p.res_f32 && p.rln_stats) {
            // round 5 (the "bf16" mode's f32 residual stream): res holds the PRE-LayerNorm f32 sum, the stream value is rebuilt here from the row
            // statistics the LayerNorm launch left -- that launch no longer writes the f32 stream (N % 8 == 0: checked by the launcher)
            const float* rp = reinterpret_cast<const float*>(p.res) + (size_t)m * p.ldr + n;
            const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
            float rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            rln_apply(p, m, n, rr);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rr[e];
        } else if (p.res && p.res_f32) {
            const float* rp = reinterpret_cast<const float*>(p.res) + (size_t)m * p.ldr + n;
            const float4 r0 = *reinterpret_cast<const float4*>(rp);
            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
            if (hi_ok) {
                const float4 r1 = *reinterpret_cast<const float4*>(rp + 4);
                v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
            }
        } else if (p.res) {
            const T* rp = reinterpret_cast<const T*>(p.res) + (size_t)m * p.ldr + n;
            if constexpr (sizeof(T) == 2) {
                if (have_pre) {                 // residual chunk prefetched at kernel start (see igemm_dma_kernel)
                    float rr[8];
                    cvt_chunk<T>(rpre[pass < NPRE ? pass : 0], rr);
                    if (p.rln_stats) rln_apply(p, m, n, rr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rr[e];
                } else if (wide16r) {
                    float rr[8];
                    ld_chunk(rp, rr);
                    if (p.rln_stats) rln_apply(p, m, n, rr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rr[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += Tr<T>::ld(rp + e);
                    if (hi_ok) {
#pragma unroll
                        for (int e = 4; e < 8; ++e) v[e] += Tr<T>::ld(rp + e);
                    }
                }
            } else {
                const float4 r0 = *reinterpret_cast<const float4*>(rp);
                v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                if (hi_ok) {
                    const float4 r1 = *reinterpret_cast<const float4*>(rp + 4);
                    v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                }
            }
        }
        if (p.act == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e]);
        } else if (p.act == ACT_GELU) {
            gelu_vec<T, 8>(v);
        }
        if (p.out_f32 || sizeof(T) == 4) {
            float* yp = reinterpret_cast<float*>(p.y) + (size_t)m * p.ldy + n;
            *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
            if (hi_ok) *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            if constexpr (sizeof(T) == 2) {
                T* yp = reinterpret_cast<T*>(p.y) + (size_t)m * p.ldy + n;
                if (wide16) {
                    st_chunk(yp, v);
                } else {
                    T o4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) Tr<T>::st(&o4[e], v[e]);
                    *reinterpret_cast<uint2*>(yp) = *reinterpret_cast<const uint2*>(o4);
                    if (hi_ok) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) Tr<T>::st(&o4[e], v[4 + e]);
                        *reinterpret_cast<uint2*>(yp + 4) = *reinterpret_cast<const uint2*>(o4);
                    }
                }
            }
        }
    }
}

// Epilogue with the HORIZONTAL half of MaxPool2d(3, 2, 1) fused (7x7 stem -> max-pool): a tile holds whole rows of the conv's
// output map (BM % Wo == 0), so the three horizontal neighbours of a pooled pixel sit in the f32 image; the conv map is written
// at half width, max(relu(x + b)) == relu(max(x) + b) exactly (monotonic adds and rounding), and the stand-alone pool that follows
// only has the vertical half left (vpool3s2_kernel).  y is [M / Wo][Wo / 2][ldy].
template <typename T, int BM, int BN, int NW, int WMc>
__device__ __forceinline__ void igemm_epilogue_hpool(const IGemmDev& p, f32x4 (&acc)[BN / (NW / WMc) / 16][BM / WMc / 16], char* smem, int m0, int n0,
                                                     int tid, int wm, int wn, int fr, int fg) {
    constexpr int WNc = NW / WMc;
    constexpr int TM = BM / WMc / 16;
    constexpr int TN = BN / WNc / 16;
    constexpr int LDC = BN + 4;
    float* sc = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int r = wm * (BM / WMc) + j * 16 + fr;
            const int cc = wn * (BN / WNc) + i * 16 + fg * 4;
            *reinterpret_cast<float4*>(sc + r * LDC + cc) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    __syncthreads();
    constexpr int TPR = BN / 8;
    constexpr int QPP = 64 * NW / TPR;                 // pooled pixels per pass
    const int c8 = (tid % TPR) * 8;
    const int n = n0 + c8;
    if (n + 8 > p.N) return;
    float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
    }
    const int Wq = p.Wo >> 1;
#pragma unroll
    for (int pass = 0; pass < (BM / 2) / QPP; ++pass) {
        const int q = pass * QPP + tid / TPR;          // pooled pixel of the tile
        const int row = q / Wq, pc = q - row * Wq;
        const int r1 = row * p.Wo + 2 * pc;            // centre tap; left neighbour clamped onto it at the map's edge
        if (m0 + r1 >= p.M) continue;
        const int r0 = pc ? r1 - 1 : r1, r2 = r1 + 1;
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 a = *reinterpret_cast<const float4*>(sc + r0 * LDC + c8 + 4 * h);
            const float4 b = *reinterpret_cast<const float4*>(sc + r1 * LDC + c8 + 4 * h);
            const float4 c = *reinterpret_cast<const float4*>(sc + r2 * LDC + c8 + 4 * h);
            v[4 * h + 0] = max_nan(max_nan(a.x, b.x), c.x); v[4 * h + 1] = max_nan(max_nan(a.y, b.y), c.y);
            v[4 * h + 2] = max_nan(max_nan(a.z, b.z), c.z); v[4 * h + 3] = max_nan(max_nan(a.w, b.w), c.w);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] += bias8[e];
            if (p.act == ACT_RELU) v[e] = relu_f(v[e]);
        }
        const size_t pix = (size_t)((m0 / p.Wo) + row) * Wq + pc;
        if constexpr (sizeof(T) == 2) st_chunk(reinterpret_cast<T*>(p.y) + pix * p.ldy + n, v);
    }
}

// The same epilogue with the f32 image processed in SPLIT row slabs (fused bottleneck kernel: one slab of LDS instead of the
// whole tile).  A separate function: a run-time slab loop in the shared epilogue costs every kernel ~60 VGPRs.
template <typename T, int BM, int BN, int NW, int WMc, int NPRE, int SPLIT>
__device__ __forceinline__ void igemm_epilogue_split(const IGemmDev& p, f32x4 (&acc)[BN / (NW / WMc) / 16][BM / WMc / 16], char* smem, int m0, int n0,
                                               int tid, int wm, int wn, int fr, int fg, const uint4 (&rpre)[NPRE], bool have_pre,
                                                     char* ytile = nullptr) {
    constexpr int WNc = NW / WMc;          // waves along the channel axis (WMc along the pixel axis)
    constexpr int TM = BM / WMc / 16;
    constexpr int TN = BN / WNc / 16;
    // ---- epilogue ----
    // Phase 1: every lane parks its accumulators (4 consecutive channels of one pixel) in an f32 LDS image of the
    // output tile (the A/B tiles are dead: the K loop ended with a barrier).  Phase 2: each thread takes 8
    // consecutive channels of a row, applies bias + residual + activation in f32, rounds once, and stores 16 B --
    // a row of the tile leaves as one contiguous BN*sizeof(T)-byte run (the accumulator layout alone would store
    // 32-byte fragments).  Row stride BN+4 floats keeps the ds_write_b128 of phase 1 conflict free.
    // SPLIT = 2 / 4: the image holds BM/2 or BM/4 rows at a time and the two phases run once
    // per slab, so the launch needs one slab of LDS instead of the whole f32 tile.
    constexpr int LDC = BN + 4;
    float* sc = reinterpret_cast<float*>(smem);
    constexpr int split = SPLIT;
    constexpr int slab_rows = BM / split;
    constexpr int TPR = BN / 8;            // threads per tile row
    constexpr int RPP = 64 * NW / TPR;     // rows per pass
    const int c8 = (tid % TPR) * 8;
    const int n = n0 + c8;
    const bool n_ok = n < p.N;
    const bool hi_ok = (n + 4) < p.N;      // N % 4 == 0: the second group of four is all-valid or all-invalid
    float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias && n_ok) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n);
        bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
        if (hi_ok) {
            const float4 b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
            bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
        }
    }
    const bool wide16 = sizeof(T) == 2 && !p.out_f32 && hi_ok && (p.ldy % 8 == 0);
    const bool wide16r = sizeof(T) == 2 && hi_ok && (p.ldr % 8 == 0);
    // (see igemm_epilogue: keeps per-pass s_waitcnt vmcnt(0) -- and with it the drain of the previous pass's stores -- out of the loop)
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(bias8[e]));
    if (have_pre) {
#pragma unroll
        for (int pass = 0; pass < NPRE; ++pass) asm volatile("" ::"v"(rpre[pass].x));
    }
#pragma unroll
  for (int slab = 0; slab < split; ++slab) {
    const int row0 = slab * slab_rows;
    if (slab) __syncthreads();             // the previous slab's rows have been read
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int rb = wm * (BM / WMc) + j * 16;       // wave-uniform: a 16-row fragment lies inside one slab
        if (rb < row0 || rb >= row0 + slab_rows) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int r = rb + fr - row0;
            const int cc = wn * (BN / WNc) + i * 16 + fg * 4;
            *reinterpret_cast<float4*>(sc + r * LDC + cc) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
    __syncthreads();
    // Fused GroupNorm (habitat GN-ResNet layers whose map has <= 64 pixels): the tile holds BM / hw whole samples and BN / cg
    // whole groups; one wave per (sample, group) reduces its hw x cg block of the f32 image (fixed order), then the
    // normalisation rides in phase 2.  The un-normalised conv output is never written.
    float* gst = sc + BM * LDC;            // [sample][group][mean, rstd]
    const int gn_ng = p.gn_cg ? BN / p.gn_cg : 0;
    if (p.gn_cg) {
        const int hw = p.gn_hw, cg = p.gn_cg;
        const int pairs = (BM / hw) * gn_ng;
        const int lane_ = tid & 63;
        for (int pr = tid >> 6; pr < pairs; pr += NW) {
            const int sidx = pr / gn_ng, g = pr - sidx * gn_ng;
            const int ne = hw * cg;
            float a = 0.f, q = 0.f;
            for (int e = lane_; e < ne; e += 64) {
                const int r = sidx * hw + e / cg, cix = g * cg + e % cg;
                const float v = sc[r * LDC + cix];
                a += v; q += v * v;
            }
            a = wave_sum(a); q = wave_sum(q);
            if (lane_ == 0) {
                const float inv = 1.0f / (float)ne;
                const float mean = a * inv;
                gst[pr * 2] = mean;
                gst[pr * 2 + 1] = rsqrtf(relu_f(q * inv - mean * mean) + p.gn_eps);
            }
        }
        __syncthreads();
    }
    if (!n_ok) continue;
#pragma unroll
    for (int pass = 0; pass < BM / RPP; ++pass) {
        if (pass * RPP < row0 || pass * RPP >= row0 + slab_rows) continue;
        const int r = pass * RPP + tid / TPR;
        const int m = m0 + r;
        if (m >= p.M) continue;
        float v[8];
        {
            const float4 a0 = *reinterpret_cast<const float4*>(sc + (r - row0) * LDC + c8);
            const float4 a1 = *reinterpret_cast<const float4*>(sc + (r - row0) * LDC + c8 + 4);
            v[0] = a0.x + bias8[0]; v[1] = a0.y + bias8[1]; v[2] = a0.z + bias8[2]; v[3] = a0.w + bias8[3];
            v[4] = a1.x + bias8[4]; v[5] = a1.y + bias8[5]; v[6] = a1.z + bias8[6]; v[7] = a1.w + bias8[7];
        }
        if (p.gn_cg) {                     // cg is a multiple of 8: the thread's 8 channels share one group
            const float* ms = gst + ((r / p.gn_hw) * gn_ng + c8 / p.gn_cg) * 2;
            const float mean = ms[0], rstd = ms[1];
            const float4 g0 = *reinterpret_cast<const float4*>(p.gn_gamma + n), g1 = *reinterpret_cast<const float4*>(p.gn_gamma + n + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(p.gn_beta + n), b1 = *reinterpret_cast<const float4*>(p.gn_beta + n + 4);
            const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean) * rstd * ga[e] + be[e];
        }
        if (p.res && p.res_f32) {
            const float* rp = reinterpret_cast<const float*>(p.res) + (size_t)m * p.ldr + n;
            const float4 r0 = *reinterpret_cast<const float4*>(rp);
            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
            if (hi_ok) {
                const float4 r1 = *reinterpret_cast<const float4*>(rp + 4);
                v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
            }
        } else if (p.res) {
            const T* rp = reinterpret_cast<const T*>(p.res) + (size_t)m * p.ldr + n;
            if constexpr (sizeof(T) == 2) {
                if (have_pre) {                 // residual chunk prefetched at kernel start (see igemm_dma_kernel)
                    float rr[8];
                    cvt_chunk<T>(rpre[pass < NPRE ? pass : 0], rr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rr[e];
                } else if (wide16r) {
                    float rr[8];
                    ld_chunk(rp, rr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rr[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += Tr<T>::ld(rp + e);
                    if (hi_ok) {
#pragma unroll
                        for (int e = 4; e < 8; ++e) v[e] += Tr<T>::ld(rp + e);
                    }
                }
            } else {
                const float4 r0 = *reinterpret_cast<const float4*>(rp);
                v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                if (hi_ok) {
                    const float4 r1 = *reinterpret_cast<const float4*>(rp + 4);
                    v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                }
            }
        }
        if (p.act == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e]);
        } else if (p.act == ACT_GELU) {
            gelu_vec<T, 8>(v);
        }
        if (p.out_f32 || sizeof(T) == 4) {
            float* yp = reinterpret_cast<float*>(p.y) + (size_t)m * p.ldy + n;
            *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
            if (hi_ok) *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            if constexpr (sizeof(T) == 2) {
                T* yp = reinterpret_cast<T*>(p.y) + (size_t)m * p.ldy + n;
                if (wide16) {
                    st_chunk(yp, v);
                    if (ytile) {        // the same 16 bytes, parked as an MFMA operand row chunk (swizzled like a DMA'd tile; BN <= 64)
                        T o8[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) Tr<T>::st(&o8[e], v[e]);
                        *reinterpret_cast<uint4*>(ytile + r * 128 + (((c8 >> 3) ^ (r & 7)) << 4)) = *reinterpret_cast<const uint4*>(o8);
                    }
                } else {
                    T o4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) Tr<T>::st(&o4[e], v[e]);
                    *reinterpret_cast<uint2*>(yp) = *reinterpret_cast<const uint2*>(o4);
                    if (hi_ok) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) Tr<T>::st(&o4[e], v[4 + e]);
                        *reinterpret_cast<uint2*>(yp + 4) = *reinterpret_cast<const uint2*>(o4);
                    }
                }
            }
        }
    }
  }
}

// S = void: activations are T with Cin a multiple of the 16-byte chunk (the general path).
// S = float / uint8_t / T: narrow-channel FIRST layers (Cin = 1 or 3: the 7x7/2 stems, SimpleCNN's 8x8/4): the im2col row
// is gathered element-wise straight from the raw frame (`permute`, `/255`, dtype conversion fused; no im2col matrix).
template <typename S> __device__ __forceinline__ float ld_src_elem(const S* p) { return (float)*p; }
template <> __device__ __forceinline__ float ld_src_elem<bf16>(const bf16* p) { return bf2f(p->v); }
template <> __device__ __forceinline__ float ld_src_elem<f16>(const f16* p) { return Tr<f16>::ld(p); }

template <typename T, int BM, int BN, typename S = void, int PF = 1>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IGemmDev p) {
    constexpr int CH = Tr<T>::CH;          // elements per 16-byte chunk
    constexpr int BK = 8 * CH;             // elements per 128-byte tile row
    constexpr int TM = BM / 32;            // 16-wide pixel tiles per wave
    constexpr int TN = BN / 32;            // 16-wide channel tiles per wave
    constexpr int A_IT = BM / 32;          // 16-byte chunks staged per thread
    constexpr int B_IT = BN / 32;
    constexpr int TILE_BYTES = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.groups > 1) {
        const long long g = blockIdx.y;
        p.x += g * p.g_x * (long long)sizeof(T);
        p.w += g * p.g_w * (long long)sizeof(T);
        if (p.bias) p.bias += g * p.g_b;
        if (p.res) p.res += g * p.g_y * (long long)sizeof(T);
        p.y += g * p.g_y * (long long)(p.out_f32 ? 4 : sizeof(T));
    }

    // XCD-aware tile order: block b runs on XCD b%8; give each XCD whole pixel-tiles (all channel tiles of
    // one pixel tile back to back) so the gathered activation rows are re-read from that XCD's L2.
    int tile_m, tile_n;
    if (!tile_of_block(p, blockIdx.x, tile_m, tile_n)) return;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int c = tid & 7;                 // chunk column this thread stages
    const int r0 = tid >> 3;               // first row this thread stages

    // ---- per-thread gather coordinates (fixed for the whole K loop) ----
    int a_pix[A_IT], a_iy0[A_IT], a_ix0[A_IT];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + 32 * i;
        if (m < p.M) {
            const int b = m / HoWo;
            const int rem = m - b * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            a_pix[i] = b * p.H * p.W;
            a_iy0[i] = oy * p.stride - p.pad;
            a_ix0[i] = ox * p.stride_w - p.pad;
        } else {
            a_pix[i] = -1; a_iy0[i] = 0; a_ix0[i] = 0;
        }
    }
    const T* xg = reinterpret_cast<const T*>(p.x);
    const T* wg = reinterpret_cast<const T*>(p.w);
    const bool spatial = (p.KH * p.KW) > 1;

    // PF register sets: with PF = 2 the loads of tile t+2 are issued while tile t+1 is still in flight (two tiles of
    // loads outstanding per workgroup: in-flight bytes, not LDS capacity, bound the latency-limited shapes)
    uint4 rsa[PF][A_IT], rsb[PF][B_IT];
    auto load_tiles = [&](int kt, auto SET) {
        uint4 (&ra)[A_IT] = rsa[decltype(SET)::value];
        uint4 (&rb)[B_IT] = rsb[decltype(SET)::value];
        const int k = kt * BK + c * CH;
        if constexpr (std::is_void<S>::value) {
            const bool kvalid = k < p.K;
            int kh = 0, kw = 0, ci = k;
            if (spatial) {
                const int khw = k >> p.cin_shift;
                ci = k & (p.Cin - 1);
                kh = (khw * p.kw_rcp) >> 16;
                kw = khw - kh * p.KW;
            }
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
                const bool ok = kvalid && a_pix[i] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ok) v = *reinterpret_cast<const uint4*>(xg + (size_t)(a_pix[i] + iy * p.W + ix) * p.xC + ci);
                ra[i] = v;
            }
        } else {
            const S* xs = reinterpret_cast<const S*>(p.x);
            if constexpr (std::is_same<S, float>::value) {
                if (p.rowrun) {
                    // RGB f32 stem fast path.  K is laid out as KH runs of 24 (= KW*3 = 21 taps + 3 zero-weight pads): for a
                    // fixed kernel row the 21 taps of a pixel are 21 CONSECUTIVE floats of the NHWC frame, so a chunk of CH
                    // k-values is CH consecutive floats: CH/4 bounds-checked 16-byte buffer loads (4-byte aligned) instead
                    // of CH scalar gathers; taps left/right of the image are masked per element afterwards.
                    const int kh = k / 24, j0 = k - kh * 24;
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
#pragma unroll
                    for (int i = 0; i < A_IT; ++i) {
                        const int iy = a_iy0[i] + kh;
                        const bool rowok = kh < p.KH && a_pix[i] >= 0 && (unsigned)iy < (unsigned)p.H;
                        const int off = (a_pix[i] + iy * p.W + a_ix0[i]) * 3 + j0;           // floats; may be negative (-> OOB -> 0)
                        float v[CH];
                        // the run may start before the first / end after the last float of the tensor (first and last image
                        // rows only): do not rely on partial range checking of a 16-byte access there, gather those element-wise
                        const bool inside = off >= 0 && (unsigned)(off + CH) * 4u <= p.x_bytes;
                        if (inside || !rowok) {
#pragma unroll
                            for (int q = 0; q < CH / 4; ++q) {
                                const f32x4 ld = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    rs, rowok ? (off + 4 * q) * 4 : -1, 0, 0));
                                v[4 * q] = ld[0]; v[4 * q + 1] = ld[1]; v[4 * q + 2] = ld[2]; v[4 * q + 3] = ld[3];
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < CH; ++e) {
                                const int o1 = off + e;
                                v[e] = (o1 >= 0 && (unsigned)(o1 + 1) * 4u <= p.x_bytes) ? xs[o1] : 0.f;
                            }
                        }
                        T packed[CH];
#pragma unroll
                        for (int e = 0; e < CH; ++e) {
                            const int ix = a_ix0[i] + (j0 + e) / 3;
                            const bool ok = rowok && (unsigned)ix < (unsigned)p.W;
                            Tr<T>::st(&packed[e], ok ? v[e] * p.x_scale : 0.f);
                        }
                        ra[i] = *reinterpret_cast<const uint4*>(packed);
                    }
                    goto weights;
                }
            }
            int dkh[CH], dkw[CH], dci[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int kk = k + j;
                const int khw = p.Cin == 1 ? kk : (p.Cin == 3 ? (kk * 21846) >> 16 : kk / p.Cin);
                dci[j] = kk - khw * p.Cin;
                dkh[j] = kk < p.K ? (khw * p.kw_rcp) >> 16 : -100000;          // k tail: always out of the image
                dkw[j] = khw - ((khw * p.kw_rcp) >> 16) * p.KW;
            }
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                float v[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int iy = a_iy0[i] + dkh[j], ix = a_ix0[i] + dkw[j];
                    const bool ok = a_pix[i] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                    v[j] = ok ? ld_src_elem<S>(xs + (size_t)(a_pix[i] + iy * p.W + ix) * p.xC + dci[j]) * p.x_scale : 0.f;
                }
                T packed[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) Tr<T>::st(&packed[j], v[j]);
                ra[i] = *reinterpret_cast<const uint4*>(packed);
            }
        }
    weights:
        const bool kvalid_w = k < p.Kp;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = n0 + r0 + 32 * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kvalid_w && n < p.N) v = *reinterpret_cast<const uint4*>(wg + (size_t)n * p.Kp + k);
            rb[i] = v;
        }
    };
    auto store_tiles = [&](int buf, auto SET) {
        uint4 (&ra)[A_IT] = rsa[decltype(SET)::value];
        uint4 (&rb)[B_IT] = rsb[decltype(SET)::value];
        char* sa = smem + buf * TILE_BYTES;
        char* sb = sa + BM * 128;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int r = r0 + 32 * i;
            *reinterpret_cast<uint4*>(sa + r * 128 + ((c ^ (r & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int r = r0 + 32 * i;
            *reinterpret_cast<uint4*>(sb + r * 128 + ((c ^ (r & 7)) << 4)) = rb[i];
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    const int fr = lane & 15;      // fragment row (pixel for A-tile, channel for W-tile)
    const int fg = lane >> 4;      // 16-byte chunk within the 64-byte half row
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, PF - 1>;

    auto compute = [&](int cur) {
        const char* sa = smem + cur * TILE_BYTES;
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 xa[TM], wb[TN];
            const int chunk = ks * 4 + fg;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int r = wm * (BM / 2) + j * 16 + fr;
                xa[j] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int r = wn * (BN / 2) + i * 16 + fr;
                wb[i] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) Mma<T>::run(acc[i][j], wb[i], xa[j]);
        }
    };

    load_tiles(0, S0{});
    store_tiles(0, S0{});
    if constexpr (PF == 1) {
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) load_tiles(kt + 1, S0{});
            compute(cur);
            if (kt + 1 < nk) store_tiles(cur ^ 1, S0{});
            __syncthreads();
        }
    } else {
        if (nk > 1) load_tiles(1, S1{});
        __syncthreads();
        // iteration kt: request tile kt+2 into the set tile kt just vacated, compute tile kt, park tile kt+1 in LDS
        auto iter = [&](int kt, auto LOADSET, auto STORESET) {
            if (kt + 2 < nk) load_tiles(kt + 2, LOADSET);
            compute(kt & 1);
            if (kt + 1 < nk) store_tiles((kt + 1) & 1, STORESET);
            __syncthreads();
        };
        for (int kt = 0; kt < nk; kt += 2) {
            iter(kt, S0{}, S1{});
            if (kt + 1 < nk) iter(kt + 1, S1{}, S0{});
        }
    }

    const uint4 no_pre[1] = {make_uint4(0u, 0u, 0u, 0u)};
    igemm_epilogue<T, BM, BN>(p, acc, smem, m0, n0, tid, wm, wn, fr, fg, no_pre, false);
}


// ---------------------------------------------------------------------------------------------------------------------------
// GroupNorm ON LOAD (round 4): the register-staged implicit GEMM whose operand staging also NORMALISES.  The GroupNorm trunk's large maps
// (>= 256 pixels per sample) used to take conv (+ statistics in its epilogue) -> gn_apply_kernel (a read + write pass and a launch of their
// own: 27 launches, 0.9 GB, 0.33 ms per step) -> next conv.  Here the consumer conv reads the UN-normalised map and applies
// relu?(x * scale - shift (+ residual)) per (sample, channel) between its global loads and its LDS stores; scale / shift come from the
// producer's per-(sample, 64-pixel block, group) sums, combined in the prologue exactly as gn_apply_kernel combines them (same order, same
// expressions: the staged values are bit-identical to what that kernel would have stored).  A tile is BM pixels of ONE sample (launcher:
// Ho * Wo % BM == 0), so the table is one sample's Cin channels, in LDS behind the two tile buffers.  With gi_out (1x1, stride 1, one channel
// tile) the normalised values are written back as well -- the block output the next residual add reads.  Out-of-image taps of a 3x3 stay zero
// (the zero padding applies to the normalised map).  Reference op: habitat's GroupNorm ResNet bottleneck, resnet_encoders.py:27-62.
template <typename T, int BM, int BN>
__global__ __launch_bounds__(256, 2) void igemm_gnin_kernel(IGemmDev p) {
    constexpr int CH = Tr<T>::CH;
    constexpr int BK = 8 * CH;
    constexpr int TM = BM / 32, TN = BN / 32, A_IT = BM / 32, B_IT = BN / 32;
    constexpr int TILE_BYTES = (BM + BN) * 128;
    static_assert(sizeof(T) == 2, "16-bit storage types");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int grp = blockIdx.y;
    if (p.groups > 1) {
        const long long g = grp;
        p.x += g * p.g_x * (long long)sizeof(T);
        p.w += g * p.g_w * (long long)sizeof(T);
        if (p.bias) p.bias += g * p.g_b;
        if (p.gn_cg) { p.gn_gamma += g * p.g_b; p.gn_beta += g * p.g_b; }
        if (p.res) p.res += g * p.g_y * (long long)sizeof(T);
        p.y += g * p.g_y * (long long)(p.out_f32 ? 4 : sizeof(T));
        if (p.gi_res) p.gi_res += g * p.g_x * (long long)sizeof(T);
        if (p.gi_out) p.gi_out += g * p.g_x * (long long)sizeof(T);
    }
    int tile_m, tile_n;
    if (!tile_of_block(p, blockIdx.x, tile_m, tile_n)) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int c = tid & 7, r0 = tid >> 3;
    const int HoWo = p.Ho * p.Wo;
    const int smp = m0 / HoWo;                                   // the tile's sample (input and output)

    float* s_scale = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);
    float* s_shift = s_scale + p.Cin;
    float* s_mean = s_shift + p.Cin;
    float* s_rstd = s_mean + 64;
    float* s_part = s_rstd + 64;                                  // [ng][PS][2]
    int a_pix[A_IT], a_iy0[A_IT], a_ix0[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + 32 * i;
        if (m < p.M) {
            const int b = m / HoWo;
            const int rem = m - b * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            a_pix[i] = b * p.H * p.W;
            a_iy0[i] = oy * p.stride - p.pad;
            a_ix0[i] = ox * p.stride_w - p.pad;
        } else {
            a_pix[i] = -1; a_iy0[i] = 0; a_ix0[i] = 0;
        }
    }
    const T* xg = reinterpret_cast<const T*>(p.x);
    const T* rg = reinterpret_cast<const T*>(p.gi_res);
    T* og = reinterpret_cast<T*>(p.gi_out);
    const T* wg = reinterpret_cast<const T*>(p.w);
    const bool spatial = (p.KH * p.KW) > 1;
    const bool wback = og != nullptr && tile_n == 0;

    uint4 rsa[A_IT], rsb[B_IT], rsr[A_IT];
    size_t roff[A_IT];
    bool rok[A_IT];
    int rci = 0;
    auto load_tiles = [&](int kt) {
        const int k = kt * BK + c * CH;
        const bool kvalid = k < p.K;
        int kh = 0, kw = 0, ci = k;
        if (spatial) {
            const int khw = k >> p.cin_shift;
            ci = k & (p.Cin - 1);
            kh = (khw * p.kw_rcp) >> 16;
            kw = khw - kh * p.KW;
        }
        rci = ci;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
            const bool ok = kvalid && a_pix[i] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const size_t off = (size_t)(a_pix[i] + iy * p.W + ix) * p.xC + ci;
            rok[i] = ok; roff[i] = off;
            uint4 v = make_uint4(0, 0, 0, 0), r = make_uint4(0, 0, 0, 0);
            if (ok) {
                v = *reinterpret_cast<const uint4*>(xg + off);
                if (rg) r = *reinterpret_cast<const uint4*>(rg + off);
            }
            rsa[i] = v; rsr[i] = r;
        }
        const bool kvalid_w = k < p.Kp;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = n0 + r0 + 32 * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kvalid_w && n < p.N) v = *reinterpret_cast<const uint4*>(wg + (size_t)n * p.Kp + k);
            rsb[i] = v;
        }
    };
    // normalise the staged operand chunks (after the loads have been issued, before they go to LDS)
    auto transform = [&]() {
        float sc[CH], sh[CH];
        const bool cvalid = rci + CH <= p.Cin;
#pragma unroll
        for (int j = 0; j < CH; ++j) { sc[j] = cvalid ? s_scale[rci + j] : 0.f; sh[j] = cvalid ? s_shift[rci + j] : 0.f; }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            if (!rok[i]) continue;                                  // out of the image / past M / past K: stays zero
            float v[CH], r[CH];
            cvt_chunk<T>(rsa[i], v);
            if (rg) cvt_chunk<T>(rsr[i], r);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                float o = v[j] * sc[j] - sh[j];
                if (rg) o += r[j];
                if (p.gi_relu) o = relu_f(o);
                v[j] = o;
            }
            rsa[i] = pack_chunk<T>(v);
            if (wback) *reinterpret_cast<uint4*>(og + roff[i]) = rsa[i];
        }
    };
    auto store_tiles = [&](int buf) {
        char* sa = smem + buf * TILE_BYTES;
        char* sb = sa + BM * 128;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int r = r0 + 32 * i;
            *reinterpret_cast<uint4*>(sa + r * 128 + ((c ^ (r & 7)) << 4)) = rsa[i];
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int r = r0 + 32 * i;
            *reinterpret_cast<uint4*>(sb + r * 128 + ((c ^ (r & 7)) << 4)) = rsb[i];
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = (p.K + BK - 1) / BK;
    const int fr = lane & 15, fg = lane >> 4;
    auto compute = [&](int cur) {
        const char* sa = smem + cur * TILE_BYTES;
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 xa[TM], wb[TN];
            const int chunk = ks * 4 + fg;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int r = wm * (BM / 2) + j * 16 + fr;
                xa[j] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int r = wn * (BN / 2) + i * 16 + fr;
                wb[i] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) Mma<T>::run(acc[i][j], wb[i], xa[j]);
        }
    };
    load_tiles(0);
    // ---- this sample's scale / shift for the conv group's Cin input channels: gn_apply_kernel's arithmetic, verbatim -- the partial sums of a group are
    // added in block order by ONE thread -- but fetched by PS x groups threads at once (one round trip instead of PS dependent ones per workgroup),
    // and behind the first tile's global loads, which do not depend on it
    {
        const int ch0 = (p.groups > 1 ? grp * (int)p.g_x : 0);   // first input channel of this conv group inside the map
        const int ng = p.Cin / p.gi_cg, g0 = ch0 / p.gi_cg;
        for (int e = tid; e < ng * p.gi_ps; e += 256) {
            const int g = e / p.gi_ps, i = e - g * p.gi_ps;
            const float2 o = *reinterpret_cast<const float2*>(p.gi_stats + (((size_t)smp * p.gi_ps + i) * p.gi_G + g0 + g) * 2);
            s_part[e * 2] = o.x; s_part[e * 2 + 1] = o.y;
        }
        __syncthreads();
        for (int g = tid; g < ng; g += 256) {
            float a = 0.f, q = 0.f;
            for (int i = 0; i < p.gi_ps; ++i) { a += s_part[(g * p.gi_ps + i) * 2]; q += s_part[(g * p.gi_ps + i) * 2 + 1]; }
            const float inv_n = 1.0f / ((float)p.gi_hw * (float)p.gi_cg);
            const float mean = a * inv_n;
            const float var = relu_f(q * inv_n - mean * mean);
            s_mean[g] = mean;
            s_rstd[g] = rsqrtf(var + p.gi_eps);
        }
        __syncthreads();
        for (int ch = tid; ch < p.Cin; ch += 256) {
            const int g = ch / p.gi_cg;
            const float sc = s_rstd[g] * p.gi_gamma[ch0 + ch];
            s_scale[ch] = sc;
            s_shift[ch] = s_mean[g] * sc - p.gi_beta[ch0 + ch];
        }
        __syncthreads();
    }
    transform();
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        compute(cur);
        if (kt + 1 < nk) { transform(); store_tiles(cur ^ 1); }
        __syncthreads();
    }
    const uint4 no_pre[1] = {make_uint4(0u, 0u, 0u, 0u)};
    igemm_epilogue<T, BM, BN>(p, acc, smem, m0, n0, tid, wm, wn, fr, fg, no_pre, false);
}

template <typename T, int BM, int BN>
static hipError_t launch_gnin_cfg(IGemmDev d, hipStream_t s) {
    d.tilesM = (d.M + BM - 1) / BM;
    d.tilesN = (d.N + BN - 1) / BN;
    d.map = 0;
    const int grid = ((d.tilesM + 7) / 8) * 8 * d.tilesN;
    size_t lds = 2 * (size_t)(BM + BN) * 128 + (size_t)(2 * d.Cin + 128 + 2 * (d.Cin / d.gi_cg) * d.gi_ps) * 4;
    const size_t lds_c = (size_t)BM * (BN + 4) * 4 + 1024 + (d.cs_part ? 64 * 8 * 2 * 4 : 0);
    if (lds_c > lds) lds = lds_c;
    const void* fn = reinterpret_cast<const void*>(igemm_gnin_kernel<T, BM, BN>);
    static DeviceOnce once;
    if (once.need()) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        once.done();
    }
    void* args[] = {&d};
    return hipLaunchKernel(fn, dim3(grid, d.groups), dim3(256), args, lds, s);
}
template <typename T>
static hipError_t launch_gnin(const IGemmDev& d, hipStream_t s) {
    if (d.gn_cg) return launch_gnin_cfg<T, 64, 128>(d, s);        // fused GroupNorm epilogue: 64 x 128 tiles of whole samples and groups
    // 128-pixel tiles when the map allows and there are plenty of them (every workgroup pays the scale / shift prologue once)
    const bool big = (d.Ho * d.Wo) % 128 == 0 && (long)d.groups * (d.M / 128) >= 512;
    if (d.N <= 32) return big ? launch_gnin_cfg<T, 128, 32>(d, s) : launch_gnin_cfg<T, 64, 32>(d, s);
    if (d.N <= 64) return big ? launch_gnin_cfg<T, 128, 64>(d, s) : launch_gnin_cfg<T, 64, 64>(d, s);
    return big ? launch_gnin_cfg<T, 128, 128>(d, s) : launch_gnin_cfg<T, 64, 128>(d, s);
}
bool igemm_gnin_ok(const IGemm& g, int dt) {
    if (dt != DT_BF16 && dt != DT_F16) return false;
    const int hw_out = g.Ho * g.Wo;
    if (!g.gi_stats || !g.gi_gamma || !g.gi_beta || g.gi_cg < 1 || g.gi_ps < 1 || g.gi_hw < 1) return false;
    if (g.x_src_dt >= 0 || g.hpool || g.out_f32 || g.res_f32 || hw_out % 64 || g.M % hw_out || g.Cin % 8 || g.Cin % g.gi_cg || g.Cin > 1024 ||
        g.Cin / g.gi_cg > 64 || (g.KH * g.KW > 1 && (g.Cin & (g.Cin - 1))))
        return false;
    if (g.groups > 1 && (g.g_x % g.gi_cg)) return false;
    if (g.gi_out && (g.KH != 1 || g.KW != 1 || g.stride != 1 || g.pad != 0 || g.N > 128)) return false;
    if (g.gi_res && !g.gi_out && (g.KH * g.KW > 1)) return false;     // (a residual on a 3x3 consumer is never needed)
    return true;
}

// Register epilogue (round 3) for the plain case -- 16-bit output, bias / residual / activation only, a wave tile of an EVEN number of
// 16-channel accumulator tiles: swap_pair (v_permlane16_swap_b32, dev.h) regroups two adjacent tiles into 8 consecutive channels of one pixel per
// lane, so bias + residual + activation + the one rounding happen in registers and every lane stores its 16 bytes directly -- no f32 LDS
// image, no barrier.  The same f32 operations in the same order as igemm_epilogue: bit-identical.  A wave instruction stores 64
// contiguous bytes per pixel (16 pixels); the neighbouring channel wave completes the 128-byte lines within the same few hundred cycles.
// rpre (optional): the residual chunks in THIS function's lane mapping, rpre[ip * TM + j].
template <typename T, int BM, int BN, int NW, int WMc, int NPRE>
__device__ __forceinline__ void igemm_epilogue_regs(const IGemmDev& p, f32x4 (&acc)[BN / (NW / WMc) / 16][BM / WMc / 16], int m0, int n0,
                                                    int wm, int wn, int fr, int fg, const uint4 (&rpre)[NPRE], bool have_pre) {
    constexpr int WNc = NW / WMc;
    constexpr int TM = BM / WMc / 16;
    constexpr int TN = BN / WNc / 16;
    static_assert(TN % 2 == 0 && sizeof(T) == 2 && (TN / 2) * TM == NPRE, "register epilogue: tile pairs");
    const int coff = (fg & 1) * 16 + (fg >> 1) * 8;
#pragma unroll
    for (int ip = 0; ip < TN / 2; ++ip) {
        const int n = n0 + wn * (BN / WNc) + ip * 32 + coff;
        const bool n_ok = n + 8 <= p.N;                    // N % 8 == 0 (checked by the caller): a chunk is all-valid or all-invalid
        float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.bias && n_ok) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
            bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            float v[8];
            swap_pair(acc[2 * ip][j], acc[2 * ip + 1][j], v);
            const int m = m0 + wm * (BM / WMc) + j * 16 + fr;
            const bool ok = n_ok && m < p.M;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias8[e];
            if (p.res) {
                float rr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (have_pre) cvt_chunk<T>(rpre[ip * TM + j], rr);
                else if (ok) ld_chunk(reinterpret_cast<const T*>(p.res) + (size_t)m * p.ldr + n, rr);
                if (p.rln_stats && ok) rln_apply(p, m, n, rr);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rr[e];
            }
            if (p.act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e]);
            } else if (p.act == ACT_GELU) {
                gelu_vec<T, 8>(v);
            }
            const uint4 o = pack_chunk<T>(v);
            if (ok) *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.y) + (size_t)m * p.ldy + n) = o;
            if (p.ln_part) {
                // statistics of the STORED (rounded) values over the 32 channels this wave column holds of row m: the four K-group lanes of the row
                float r8[8];
                cvt_chunk<T>(o, r8);
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { a += r8[e]; q += r8[e] * r8[e]; }
                a += __shfl_xor(a, 16, 64); q += __shfl_xor(q, 16, 64);
                a += __shfl_xor(a, 32, 64); q += __shfl_xor(q, 32, 64);
                if (fg == 0 && m < p.M && n_ok) {
                    const int slot = (n0 + wn * (BN / WNc) + ip * 32) >> 5;
                    *reinterpret_cast<float2*>(p.ln_part + ((size_t)m * p.ln_part_P + slot) * 2) = make_float2(a, q);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Main variant: LDS-DMA staging, 3-deep ring, counted waits.
//   * both tiles go L2/HBM -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR round trip, no ds_write pass).  The DMA
//     destination is lane-linear (wave base + lane*16 B): one wave instruction fills 8 tile rows x 128 B, so the XOR
//     swizzle moves to the SOURCE address: lane (row r, slot c') fetches chunk c' ^ (r & 7) (rule 21 of the guide).
//   * buffer resources with range checking: a lane whose im2col tap is outside the image (or past M / N / K) presents
//     offset 0xFFFFFFFF and the hardware deposits zeros -- branch-free zero fill.
//   * the DMA is issued from inline asm: hipcc models an LDS-DMA builtin as an LDS store and would put
//     `s_waitcnt vmcnt(0)` in front of every ds_read of the loop, exposing the whole load latency.  Hidden from the
//     compiler, tile t+2 is requested before tile t is consumed and only `vmcnt(<loads of one tile>)` is waited for at
//     the end of an iteration (cdna_hip_programming.md T3/T4): two tiles of loads stay in flight across the barrier.
typedef int v4i_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, v4i_t rsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

__device__ __forceinline__ v4i_t make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    v4i_t r;
    r[0] = (int)(unsigned)a;
    r[1] = (int)((unsigned)(a >> 32) & 0xFFFFu);     // stride 0
    r[2] = (int)bytes;                               // num_records (bytes for raw buffers)
    r[3] = 0x00020000;
    return r;
}

// Phase timing of the K loop (debug builds of the loop selected with HCM_IGEMM_PROF=1; read back through
// hcm_debug_igemm_prof): per-wave cycle totals [0] prologue, [1] DMA issue, [2] fragment reads + MFMA issue,
// [3] wait for the next tile's DMA, [4] barrier, [5] epilogue, [6] waves, [7] K iterations.
constexpr int kProfSlots = 65536;
__device__ unsigned long long g_igemm_prof[kProfSlots][8];
__device__ __forceinline__ unsigned long long prof_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// NW = 4 (2x2 waves, wave tile BM/2 x BN/2) or 8 (2x4 waves, wave tile BM/2 x BN/4: twice the waves per SIMD on the same
// LDS footprint -- more thread-level parallelism to cover ds_read / DMA-issue latency)
// ILV = 2: ILV = 1 plus a rotated loop -- the second K half's MFMAs are issued AFTER the barrier and after the next tile's
// first fragment reads, so the matrix pipe has work while those reads are in flight (3-deep ring only).
// ILV = 1: the DMA instructions of the next tile are not issued in one burst at the top of an iteration (where all waves
// of the workgroup queue 16-32 KB on the CU's one texture-address path at once and then sit in the issue stall) but
// one at a time between groups of MFMAs; both K halves' fragments are read up front.
template <typename T, int BM, int BN, int NBUF, int NW = 4, int WMc = 2, bool PROF = false, int ILV = 0, bool HPOOL = false>
__global__ __launch_bounds__(64 * NW) void igemm_dma_kernel(IGemmDev p) {
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
    if constexpr (PROF) t_prev = prof_now();
    auto lap = [&](int slot) {
        if constexpr (PROF) { const unsigned long long t = prof_now(); pt[slot] += t - t_prev; t_prev = t; }
    };
    constexpr int CH = Tr<T>::CH;
    constexpr int BK = 8 * CH;
    constexpr int WNc = NW / WMc;
    constexpr int TM = BM / WMc / 16;
    constexpr int TN = BN / WNc / 16;
    constexpr int A_IT = BM / 8 / NW;      // wave-level DMA instructions per tile (8 rows each)
    constexpr int B_IT = BN / 8 / NW;
    static_assert(A_IT >= 1 && B_IT >= 1 && TN >= 1, "tile too small for this wave count");
    constexpr int LPT = A_IT + B_IT;       // DMA instructions per wave per K tile
    constexpr int TILE_BYTES = (BM + BN) * 128;
    static_assert(NBUF >= 2 && NBUF <= 6, "ring depth");
    // NBUF > 3 (the DEEP ring): for launches whose whole grid fits the chip one workgroup per CU -- the depth trunk's 8 x 8 and 4 x 4 maps,
    // every conv at small batch -- LDS is free, and the K loop is a latency chain: an iteration waits for the tile requested NBUF - 1
    // iterations earlier, and an L2 -> LDS request takes ~0.85 us to land, so the iteration costs L / (NBUF - 1) until it reaches its own
    // ~0.15 us of fragment reads + MFMAs.  Waits stay counted: (NBUF - 2) tiles in flight across the barrier in the steady state.
    auto wait_tiles = [&](int n) {          // at most n whole tiles (n * LPT requests) of this wave still in flight
        if constexpr (NBUF <= 3) { if (n >= 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>(); }
        else {
            switch (n) {
                case 0: wait_vmcnt<0>(); break;
                case 1: wait_vmcnt<LPT>(); break;
                case 2: wait_vmcnt<2 * LPT>(); break;
                case 3: wait_vmcnt<3 * LPT>(); break;
                default: wait_vmcnt<4 * LPT>(); break;
            }
        }
    };

    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.groups > 1) {
        const long long g = blockIdx.y;
        p.x += g * p.g_x * (long long)sizeof(T);
        p.w += g * p.g_w * (long long)sizeof(T);
        if (p.bias) p.bias += g * p.g_b;
        if (p.gn_cg) { p.gn_gamma += g * p.g_b; p.gn_beta += g * p.g_b; }
        if (p.res) p.res += g * p.g_y * (long long)sizeof(T);
        p.y += g * p.g_y * (long long)(p.out_f32 ? 4 : sizeof(T));
    }

    int tile_m, tile_n;
    if (!tile_of_block(p, blockIdx.x, tile_m, tile_n)) return;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNc, wn = wave % WNc;
    const int rin = lane >> 3;                 // row inside the 8-row DMA group
    const int c = (lane & 7) ^ rin;            // source chunk this lane fetches (swizzle on the source side)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    int a_pix[A_IT], a_iy0[A_IT], a_ix0[A_IT];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + (wave + NW * i) * 8 + rin;
        if (m < p.M) {
            const int b = m / HoWo;
            const int rem = m - b * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            a_pix[i] = b * p.H * p.W;
            a_iy0[i] = oy * p.stride - p.pad;
            a_ix0[i] = ox * p.stride_w - p.pad;
        } else {
            a_pix[i] = -1; a_iy0[i] = 0; a_ix0[i] = 0;
        }
    }
    const bool spatial = (p.KH * p.KW) > 1;
    const v4i_t rx = make_rsrc(p.x, p.x_bytes);
    const v4i_t rw = make_rsrc(p.w, p.w_bytes);

    auto stage = [&](int kt, int buf) {
        const unsigned sa = lds_base + buf * TILE_BYTES;
        const unsigned sb = sa + BM * 128;
        const int k = kt * BK + c * CH;
        int kh = 0, kw = 0, ci = k;
        if (spatial) {
            const int khw = k >> p.cin_shift;
            ci = k & (p.Cin - 1);
            kh = (khw * p.kw_rcp) >> 16;
            kw = khw - kh * p.KW;
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
            const bool ok = (k < p.K) & (a_pix[i] >= 0) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            const unsigned off = (unsigned)((a_pix[i] + iy * p.W + ix) * p.xC + ci) * (unsigned)sizeof(T);
            dma16(sa + (wave + NW * i) * 1024, ok ? off : 0xFFFFFFFFu, rx);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = n0 + (wave + NW * i) * 8 + rin;
            const bool ok = (k < p.Kp) & (n < p.N);
            const unsigned off = (unsigned)(n * p.Kp + k) * (unsigned)sizeof(T);
            dma16(sb + (wave + NW * i) * 1024, ok ? off : 0xFFFFFFFFu, rw);
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    // Residual prefetch (8-wave, 16-bit kernels): the epilogue adds res[m][n..n+8) per thread and pass.  Requested here,
    // before the first tile, the residual tile travels together with the operand tiles instead of being a second exposed
    // HBM round trip after the K loop -- the 1x1 expansion convs of the ResNet bottlenecks have K = 64..512, i.e. 1-8
    // K iterations, and are bounded by exactly these round trips.  (Older than every DMA request, so each counted vmcnt
    // wait of the loop also covers them.)
    constexpr int E_TPR = BN / 8, E_RPP = 64 * NW / E_TPR, E_NP = BM / E_RPP;
    uint4 rpre[E_NP];
    bool have_pre = false;
    // register epilogue (igemm_epilogue_regs) for the plain 16-bit case; the LDS-image epilogue keeps the fused GroupNorm / statistics / pool /
    // f32-output forms and every tile whose wave holds a single 16-channel accumulator tile (p.image_epi: HCM_IGEMM_IMAGE=1, A/B)
    constexpr bool kRegsOk = sizeof(T) == 2 && !HPOOL && (TN % 2 == 0) && ((TN / 2) * TM == E_NP);
    const bool regs_epi = kRegsOk && !p.out_f32 && !p.gn_cg && !p.cs_part && !p.res_f32 && (p.ldy % 8) == 0 && (p.N % 8) == 0 &&
                          (!p.res || (p.ldr % 8) == 0) && !p.image_epi;
    if constexpr (sizeof(T) == 2 && NW == 8 && E_NP <= 8) {
        if (p.res && !p.res_f32 && (p.ldr % 8) == 0 && (p.N % 8) == 0) {
            have_pre = true;
            if (regs_epi) {
                if constexpr (kRegsOk) {
                    const int coff = ((lane >> 4) & 1) * 16 + (lane >> 5) * 8;
#pragma unroll
                    for (int ip = 0; ip < TN / 2; ++ip)
#pragma unroll
                        for (int j = 0; j < TM; ++j) {
                            const int m = m0 + wm * (BM / WMc) + j * 16 + (lane & 15);
                            const int n = n0 + wn * (BN / WNc) + ip * 32 + coff;
                            rpre[ip * TM + j] = make_uint4(0u, 0u, 0u, 0u);
                            if (m < p.M && n + 8 <= p.N) rpre[ip * TM + j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.res) + (size_t)m * p.ldr + n);
                        }
                }
            } else {
                const int n = n0 + (tid % E_TPR) * 8;
#pragma unroll
                for (int pass = 0; pass < E_NP; ++pass) {
                    const int m = m0 + pass * E_RPP + tid / E_TPR;
                    rpre[pass] = make_uint4(0u, 0u, 0u, 0u);
                    if (m < p.M && n + 8 <= p.N) rpre[pass] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.res) + (size_t)m * p.ldr + n);
                }
            }
        }
    }
    // prologue: NBUF-1 tiles in flight; tile 0 must have landed (for every wave) before the first fragment read
    stage(0, 0);
    if constexpr (NBUF <= 3) {
        if (NBUF == 3 && nk > 1) { stage(1, 1); wait_vmcnt<LPT>(); } else { wait_vmcnt<0>(); }
    } else {
        int staged = 1;
#pragma unroll
        for (int t = 1; t < NBUF - 1; ++t)
            if (t < nk) { stage(t, t); ++staged; }
        wait_tiles(staged - 1);
    }
    __builtin_amdgcn_s_barrier();
    lap(0);

    const int fr = lane & 15;
    const int fg = lane >> 4;
    int cur = 0;
    if constexpr (ILV == 2) {
        static_assert(NBUF == 3, "the rotated loop needs the 3-deep ring");
        constexpr int HM = TM * TN;                     // MFMAs per K half
        constexpr int P1 = LPT / 2;                     // DMA pieces issued among the first half's MFMAs
        constexpr int G1 = P1 > 0 ? (HM / (P1 + 1) >= 1 ? HM / (P1 + 1) : 1) : HM + 1;
        constexpr int G2 = HM / (LPT - P1 + 1) >= 1 ? HM / (LPT - P1 + 1) : 1;
        auto piece = [&](int i, unsigned sa2, unsigned sb2, int k, int kh, int kw, int ci) {
            if (i < A_IT) {
                const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
                const bool ok = (k < p.K) & (a_pix[i] >= 0) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                const unsigned off = (unsigned)((a_pix[i] + iy * p.W + ix) * p.xC + ci) * (unsigned)sizeof(T);
                dma16(sa2 + (wave + NW * i) * 1024, ok ? off : 0xFFFFFFFFu, rx);
            } else {
                const int ib = i - A_IT;
                const int n = n0 + (wave + NW * ib) * 8 + rin;
                const bool ok = (k < p.Kp) & (n < p.N);
                const unsigned off = (unsigned)(n * p.Kp + k) * (unsigned)sizeof(T);
                dma16(sb2 + (wave + NW * ib) * 1024, ok ? off : 0xFFFFFFFFu, rw);
            }
        };
        uint4 xa[2][TM], wb[2][TN];
        auto rd = [&](int ks, int slot) {
            const char* sa = smem + slot * TILE_BYTES;
            const char* sb = sa + BM * 128;
            const int chunk = ks * 4 + fg;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int r = wm * (BM / WMc) + j * 16 + fr;
                xa[ks][j] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int r = wn * (BN / WNc) + i * 16 + fr;
                wb[ks][i] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
        };
        rd(0, 0);
        rd(1, 0);
        auto body = [&](int kt, auto LIVE, auto NEXT) {
            constexpr bool live = decltype(LIVE)::value;      // tile kt+2 exists: stage it
            constexpr bool next = decltype(NEXT)::value;      // tile kt+1 exists: read its fragments
            const int slot2 = cur == 0 ? 2 : cur - 1;          // (cur + 2) % 3
            const int slot1 = cur == 2 ? 0 : cur + 1;
            const unsigned sa2 = lds_base + slot2 * TILE_BYTES, sb2 = sa2 + BM * 128;
            const int k = (kt + 2) * BK + c * CH;
            int kh = 0, kw = 0, ci = k;
            if (live && spatial) {
                const int khw = k >> p.cin_shift;
                ci = k & (p.Cin - 1);
                kh = (khw * p.kw_rcp) >> 16;
                kw = khw - kh * p.KW;
            }
            lap(1);
            int cnt = 0, pc = 0;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    Mma<T>::run(acc[i][j], wb[0][i], xa[0][j]);
                    ++cnt;
                    if (live && pc < P1 && cnt == G1 * (pc + 1)) {
                        __builtin_amdgcn_sched_barrier(0);
                        piece(pc, sa2, sb2, k, kh, kw, ci);
                        __builtin_amdgcn_sched_barrier(0);
                        ++pc;
                    }
                }
            if (live) {
#pragma unroll
                for (int q = 0; q < P1; ++q)
                    if (q >= pc) piece(q, sa2, sb2, k, kh, kw, ci);
            }
            lap(2);
            if (live) wait_vmcnt<P1>(); else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lap(3);
            __builtin_amdgcn_s_barrier();
            lap(4);
            __builtin_amdgcn_sched_barrier(0);
            if (next) rd(0, slot1);
            __builtin_amdgcn_sched_barrier(0);
            cnt = 0; pc = P1;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    Mma<T>::run(acc[i][j], wb[1][i], xa[1][j]);
                    ++cnt;
                    if (live && pc < LPT && cnt == G2 * (pc - P1 + 1)) {
                        __builtin_amdgcn_sched_barrier(0);
                        piece(pc, sa2, sb2, k, kh, kw, ci);
                        __builtin_amdgcn_sched_barrier(0);
                        ++pc;
                    }
                }
            if (live) {
#pragma unroll
                for (int q = P1; q < LPT; ++q)
                    if (q >= pc) piece(q, sa2, sb2, k, kh, kw, ci);
            }
            if (next) rd(1, slot1);
            cur = slot1;
        };
        int kt = 0;
        for (; kt + 2 < nk; ++kt) body(kt, std::true_type{}, std::true_type{});
        for (; kt + 1 < nk; ++kt) body(kt, std::false_type{}, std::true_type{});
        for (; kt < nk; ++kt) body(kt, std::false_type{}, std::false_type{});
    } else
    if constexpr (ILV == 3) {
        // ILV = 3 (round 4; deep ring, one workgroup per CU): the wave software-pipelines its fragment reads ACROSS the barrier.  ILV = 1 reads a tile's
        // 2 x (TM + TN) fragments at the top of the iteration and every wave then sits in `lgkmcnt` with eight waves hammering the LDS before its first
        // MFMA (~1330 cycles per 128 x 128 K tile for 544 of matrix work).  Here the K half computed next is always in registers: the first half's
        // MFMAs run beside the reads of this tile's second half, the second half's beside the reads of the NEXT tile's first half -- which is why
        // tile kt + 1 must be visible to everybody one barrier earlier than in ILV = 1 (the end-of-iteration wait asks for tile kt + 2: NBUF - 3
        // tiles stay in flight across the barrier).  Same MFMA order per accumulator: bit-identical.
        static_assert(NBUF >= 4, "the pipelined loop needs the deep ring");
        constexpr int NRD = TM + TN;                      // fragment reads per K half
        auto piece = [&](int i, unsigned sa2, unsigned sb2, int k, int kh, int kw, int ci) {
            if (i < A_IT) {
                const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
                const bool ok = (k < p.K) & (a_pix[i] >= 0) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                const unsigned off = (unsigned)((a_pix[i] + iy * p.W + ix) * p.xC + ci) * (unsigned)sizeof(T);
                dma16(sa2 + (wave + NW * i) * 1024, ok ? off : 0xFFFFFFFFu, rx);
            } else {
                const int ib = i - A_IT;
                const int n = n0 + (wave + NW * ib) * 8 + rin;
                const bool ok = (k < p.Kp) & (n < p.N);
                const unsigned off = (unsigned)(n * p.Kp + k) * (unsigned)sizeof(T);
                dma16(sb2 + (wave + NW * ib) * 1024, ok ? off : 0xFFFFFFFFu, rw);
            }
        };
        uint4 xa[2][TM], wb[2][TN];
        auto rd1 = [&](int f, int ks, int slot) {         // fragment f of a K half: f < TM token fragment, else weight fragment f - TM
            const char* sa = smem + slot * TILE_BYTES;
            const char* sb = sa + BM * 128;
            const int chunk = ks * 4 + fg;
            if (f < TM) {
                const int r = wm * (BM / WMc) + f * 16 + fr;
                xa[ks][f] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((chunk ^ (r & 7)) << 4));
            } else {
                const int r = wn * (BN / WNc) + (f - TM) * 16 + fr;
                wb[ks][f - TM] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
        };
        // tile 1 must be visible as well before the first iteration's second half reads it
        {
            int staged = 1;
            for (int t = 1; t < NBUF - 1; ++t) if (t < nk) ++staged;
            wait_tiles(staged - 2 > 0 ? staged - 2 : 0);
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int f = 0; f < NRD; ++f) rd1(f, 0, 0);
        auto body = [&](int kt, auto LIVE, auto NEXT) {
            constexpr bool live = decltype(LIVE)::value;      // tile kt + NBUF - 1 exists: stage it
            constexpr bool next = decltype(NEXT)::value;      // tile kt + 1 exists: read its first K half
            const int slot_new = cur == 0 ? NBUF - 1 : cur - 1;
            const int slot1 = cur == NBUF - 1 ? 0 : cur + 1;
            const unsigned sa2 = lds_base + slot_new * TILE_BYTES, sb2 = sa2 + BM * 128;
            const int k = (kt + NBUF - 1) * BK + c * CH;
            int kh = 0, kw = 0, ci = k;
            if (live && spatial) {
                const int khw = k >> p.cin_shift;
                ci = k & (p.Cin - 1);
                kh = (khw * p.kw_rcp) >> 16;
                kw = khw - kh * p.KW;
            }
            lap(1);
            // ---- first half: K step 0 (set 0) beside the reads of K step 1 (set 1) of this tile, then the first DMA requests
            int cnt = 0, pc = 0;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    Mma<T>::run(acc[i][j], wb[0][i], xa[0][j]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (cnt < NRD) rd1(cnt, 1, cur);
                    else if (live && pc < LPT) { piece(pc, sa2, sb2, k, kh, kw, ci); ++pc; }
                    __builtin_amdgcn_sched_barrier(0);
                    ++cnt;
                }
#pragma unroll
            for (int f = cnt; f < NRD; ++f) rd1(f, 1, cur);       // (tiles with fewer MFMAs per half than fragments)
            // ---- second half: K step 1 (set 1) beside the reads of the NEXT tile's K step 0 (set 0) and the remaining requests
            cnt = 0;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    Mma<T>::run(acc[i][j], wb[1][i], xa[1][j]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (next && cnt < NRD) rd1(cnt, 0, slot1);
                    else if (live && pc < LPT) { piece(pc, sa2, sb2, k, kh, kw, ci); ++pc; }
                    __builtin_amdgcn_sched_barrier(0);
                    ++cnt;
                }
            if (next) {
#pragma unroll
                for (int f = cnt; f < NRD; ++f) rd1(f, 0, slot1);
            }
            if (live) {
#pragma unroll
                for (int q = 0; q < LPT; ++q)
                    if (q >= pc) piece(q, sa2, sb2, k, kh, kw, ci);
            }
            lap(2);
            // tile kt + 2 must have landed (it is read from the next iteration's second half on); younger tiles stay in flight
            if (live) wait_vmcnt<(NBUF - 3) * LPT>(); else wait_tiles(nk - 3 - kt > 0 ? nk - 3 - kt : 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lap(3);
            __builtin_amdgcn_s_barrier();
            lap(4);
            cur = slot1;
        };
        int kt = 0;
        for (; kt + (NBUF - 1) < nk; ++kt) body(kt, std::true_type{}, std::true_type{});
        for (; kt + 1 < nk; ++kt) body(kt, std::false_type{}, std::true_type{});
        for (; kt < nk; ++kt) body(kt, std::false_type{}, std::false_type{});
    } else
    if constexpr (ILV == 1) {
        constexpr int NMMA = 2 * TM * TN;
        constexpr int GAP = NMMA / (LPT + 1) >= 1 ? NMMA / (LPT + 1) : 1;
        // one DMA instruction of tile `kt2` (piece i: A rows first, then B rows)
        auto piece = [&](int i, unsigned sa2, unsigned sb2, int k, int kh, int kw, int ci) {
            if (i < A_IT) {
                const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
                const bool ok = (k < p.K) & (a_pix[i] >= 0) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                const unsigned off = (unsigned)((a_pix[i] + iy * p.W + ix) * p.xC + ci) * (unsigned)sizeof(T);
                dma16(sa2 + (wave + NW * i) * 1024, ok ? off : 0xFFFFFFFFu, rx);
            } else {
                const int ib = i - A_IT;
                const int n = n0 + (wave + NW * ib) * 8 + rin;
                const bool ok = (k < p.Kp) & (n < p.N);
                const unsigned off = (unsigned)(n * p.Kp + k) * (unsigned)sizeof(T);
                dma16(sb2 + (wave + NW * ib) * 1024, ok ? off : 0xFFFFFFFFu, rw);
            }
        };
        auto body = [&](int kt, auto LIVE) {
            constexpr bool live = decltype(LIVE)::value;
            const int slot = cur == 0 ? NBUF - 1 : cur - 1;
            const unsigned sa2 = lds_base + slot * TILE_BYTES, sb2 = sa2 + BM * 128;
            const int k = (kt + NBUF - 1) * BK + c * CH;
            int kh = 0, kw = 0, ci = k;
            if (live && spatial) {
                const int khw = k >> p.cin_shift;
                ci = k & (p.Cin - 1);
                kh = (khw * p.kw_rcp) >> 16;
                kw = khw - kh * p.KW;
            }
            lap(1);
            const char* sa = smem + cur * TILE_BYTES;
            const char* sb = sa + BM * 128;
            uint4 xa[2][TM], wb[2][TN];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int chunk = ks * 4 + fg;
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int r = wm * (BM / WMc) + j * 16 + fr;
                    xa[ks][j] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((chunk ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    const int r = wn * (BN / WNc) + i * 16 + fr;
                    wb[ks][i] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
                }
            }
            int cnt = 0, pc = 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) {
                        Mma<T>::run(acc[i][j], wb[ks][i], xa[ks][j]);
                        ++cnt;
                        if (live && pc < LPT && cnt == GAP * (pc + 1)) {
                            __builtin_amdgcn_sched_barrier(0);
                            piece(pc, sa2, sb2, k, kh, kw, ci);
                            __builtin_amdgcn_sched_barrier(0);
                            ++pc;
                        }
                    }
            if (live) {
#pragma unroll
                for (int q = 0; q < LPT; ++q)
                    if (q >= pc) piece(q, sa2, sb2, k, kh, kw, ci);
            }
            lap(2);
            // tile kt + 1 must have landed; younger: tiles kt + 2 .. min(kt + NBUF - 1, nk - 1)
            if constexpr (NBUF <= 3) { if (NBUF == 3 && live) wait_vmcnt<LPT>(); else wait_vmcnt<0>(); }
            else { if (live) wait_vmcnt<(NBUF - 2) * LPT>(); else wait_tiles(nk - 2 - kt > 0 ? nk - 2 - kt : 0); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lap(3);
            __builtin_amdgcn_s_barrier();
            lap(4);
            cur = cur == NBUF - 1 ? 0 : cur + 1;
        };
        int kt = 0;
        for (; kt + (NBUF - 1) < nk; ++kt) body(kt, std::true_type{});
        for (; kt < nk; ++kt) body(kt, std::false_type{});
    } else
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + (NBUF - 1) < nk;
        // ring slot of tile kt+NBUF-1: it was last read in iteration kt-1, which every wave has left (barrier)
        if (more) stage(kt + NBUF - 1, cur == 0 ? NBUF - 1 : cur - 1);
        lap(1);
        const char* sa = smem + cur * TILE_BYTES;
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 xa[TM], wb[TN];
            const int chunk = ks * 4 + fg;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int r = wm * (BM / WMc) + j * 16 + fr;
                xa[j] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int r = wn * (BN / WNc) + i * 16 + fr;
                wb[i] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) Mma<T>::run(acc[i][j], wb[i], xa[j]);
        }
        // tile kt+1 must be complete before the next iteration reads it; with the 3-deep ring the tile requested in this
        // iteration may stay in flight across the barrier
        lap(2);
        if constexpr (NBUF <= 3) { if (NBUF == 3 && more) wait_vmcnt<LPT>(); else wait_vmcnt<0>(); }
        else { if (more) wait_vmcnt<(NBUF - 2) * LPT>(); else wait_tiles(nk - 2 - kt > 0 ? nk - 2 - kt : 0); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lap(3);
        __builtin_amdgcn_s_barrier();
        lap(4);
        cur = cur == NBUF - 1 ? 0 : cur + 1;
    }
    if constexpr (HPOOL) igemm_epilogue_hpool<T, BM, BN, NW, WMc>(p, acc, smem, m0, n0, tid, wm, wn, fr, fg);
    else {
        if constexpr (kRegsOk) {
            if (regs_epi) igemm_epilogue_regs<T, BM, BN, NW, WMc, E_NP>(p, acc, m0, n0, wm, wn, fr, fg, rpre, have_pre);
            else igemm_epilogue<T, BM, BN, NW, WMc, E_NP>(p, acc, smem, m0, n0, tid, wm, wn, fr, fg, rpre, have_pre);
        } else igemm_epilogue<T, BM, BN, NW, WMc, E_NP>(p, acc, smem, m0, n0, tid, wm, wn, fr, fg, rpre, have_pre);
    }
    if constexpr (PROF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lap(5);
        if (lane == 0) {
            unsigned long long* slot = g_igemm_prof[((blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) & (kProfSlots - 1)];
            for (int i = 0; i < 6; ++i) atomicAdd(&slot[i], pt[i]);          // one wave per slot (collisions only past 65536 waves)
            atomicAdd(&slot[6], 1ull);
            atomicAdd(&slot[7], (unsigned long long)nk);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Fused tail of a BatchNorm-folded ResNet bottleneck (RGB trunk, layer1 / layer2): the 3x3 conv (C1 -> C1, + bias + ReLU) and the
// 1x1 expansion (C1 -> 4*C1, + bias + identity + ReLU) in ONE launch.  A workgroup computes BM pixels x all C1 channels of
// the 3x3 conv (phase A: the ordinary LDS-DMA K loop, 3-deep ring), rounds them to the storage type exactly as the stand-alone conv's
// epilogue would, and parks them in LDS in the swizzled operand layout -- the C1-channel intermediate never goes to HBM (a
// full write + read of the M x C1 tensor per block, and one launch).  Phase B multiplies that tile with the expansion weights, 128
// output channels at a time (weights streamed by LDS-DMA, the next slice requested while the current one's epilogue runs),
// through the shared epilogue in two row slabs.  Same MFMA instruction and k order as the two stand-alone launches: bit-identical.
struct BneckDev {
    IGemmDev a;                  // phase A: x, w (3x3), bias, geometry, M, N = C1, K = Kp = 9*C1, groups / g_x / g_w / g_b
    const char* w3; const float* b3; const char* res; char* y;
    int C3, Kp3, ldy3, ldr3;
    unsigned w3_bytes;
    long long g_w3, g_b3, g_y3;  // per-group element offsets of the expansion weights / bias / (y, res)
};

template <typename T, int BM, int C1>
__global__ __launch_bounds__(512) void bneck23_kernel(BneckDev q) {
    constexpr int NW = 8, WMc = 2, WNc = 4, CH = 8, BK = 64;
    constexpr int TM = BM / WMc / 16;              // both phases: BM / 2 pixel rows per wave
    constexpr int TN1 = C1 / WNc / 16;             // phase A: C1 / 4 channels per wave
    constexpr int TN2 = 128 / WNc / 16;            // phase B: 32 of the slice's 128 channels per wave
    constexpr int A_IT = BM / 8 / NW, B_IT = C1 / 8 / NW;
    constexpr int TILE_BYTES = (BM + C1) * 128;
    constexpr int KT1 = C1 / BK;                   // K tiles of phase B
    constexpr int NT = 4 * C1 / 128;               // 128-channel output slices
    constexpr int T_BYTES = KT1 * BM * 128, W3_BYTES = KT1 * 128 * 128;
    constexpr int E_RPP = 64 * NW / (128 / 8), E_NP = BM / E_RPP;
    static_assert(A_IT >= 1 && B_IT >= 1 && TN1 >= 1 && sizeof(T) == 2, "bneck23 tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    IGemmDev& p = q.a;
    if (p.groups > 1) {
        const long long g = blockIdx.y;
        p.x += g * p.g_x * 2;
        p.w += g * p.g_w * 2;
        p.bias += g * p.g_b;
        q.w3 += g * q.g_w3 * 2;
        q.b3 += g * q.g_b3;
        q.res += g * q.g_y3 * 2;
        q.y += g * q.g_y3 * 2;
    }
    const int m0 = blockIdx.x * BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNc, wn = wave % WNc;
    const int rin = lane >> 3;
    const int c = (lane & 7) ^ rin;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // identity rows of every output slice, requested first: they arrive while phase A runs
    uint4 rpre[NT][E_NP];
    {
        const int n = (tid % 16) * 8;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int pass = 0; pass < E_NP; ++pass) {
                const int m = m0 + pass * E_RPP + tid / 16;
                rpre[nt][pass] = make_uint4(0u, 0u, 0u, 0u);
                if (m < p.M) rpre[nt][pass] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(q.res) + (size_t)m * q.ldr3 + nt * 128 + n);
            }
    }

    int a_pix[A_IT], a_iy0[A_IT], a_ix0[A_IT];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + (wave + NW * i) * 8 + rin;
        if (m < p.M) {
            const int b = m / HoWo;
            const int rem = m - b * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            a_pix[i] = b * p.H * p.W;
            a_iy0[i] = oy * p.stride - p.pad;
            a_ix0[i] = ox * p.stride - p.pad;
        } else {
            a_pix[i] = -1; a_iy0[i] = 0; a_ix0[i] = 0;
        }
    }
    const v4i_t rx = make_rsrc(p.x, p.x_bytes);
    const v4i_t rw = make_rsrc(p.w, p.w_bytes);
    const v4i_t rw3 = make_rsrc(q.w3, q.w3_bytes);

    auto stage = [&](int kt, int buf) {
        const unsigned sa = lds_base + buf * TILE_BYTES;
        const unsigned sb = sa + BM * 128;
        const int k = kt * BK + c * CH;
        const int khw = k >> p.cin_shift;
        const int ci = k & (p.Cin - 1);
        const int kh = (khw * p.kw_rcp) >> 16;
        const int kw = khw - kh * p.KW;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
            const bool ok = (k < p.K) & (a_pix[i] >= 0) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            const unsigned off = (unsigned)((a_pix[i] + iy * p.W + ix) * p.xC + ci) * 2u;
            dma16(sa + (wave + NW * i) * 1024, ok ? off : 0xFFFFFFFFu, rx);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = (wave + NW * i) * 8 + rin;
            const unsigned off = (unsigned)(n * p.Kp + k) * 2u;
            dma16(sb + (wave + NW * i) * 1024, (k < p.Kp) ? off : 0xFFFFFFFFu, rw);
        }
    };
    // expansion weights of output slice nt: KT1 blocks of [128 channels][64 k] behind the parked tile
    auto stage_w3 = [&](int nt) {
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = nt * 128 + (wave + NW * i) * 8 + rin;
                const unsigned off = (unsigned)(n * q.Kp3 + kt * BK + c * CH) * 2u;
                dma16(lds_base + T_BYTES + kt * (128 * 128) + (wave + NW * i) * 1024, off, rw3);
            }
    };

    const int fr = lane & 15;
    const int fg = lane >> 4;
    // ---- phase A: 3x3 conv, BM x C1
    f32x4 acc1[TN1][TM];
#pragma unroll
    for (int i = 0; i < TN1; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = (p.K + BK - 1) / BK;
    constexpr int LPT = A_IT + B_IT;               // DMA instructions per wave per K tile
    stage(0, 0);
    if (nk > 1) { stage(1, 1); wait_vmcnt<LPT>(); } else { wait_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 2 < nk;
        if (more) stage(kt + 2, cur == 0 ? 2 : cur - 1);       // 3-deep ring: tile kt+2 goes where tile kt-1 was
        const char* sa = smem + cur * TILE_BYTES;
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 xa[TM], wb[TN1];
            const int chunk = ks * 4 + fg;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int r = wm * (BM / WMc) + j * 16 + fr;
                xa[j] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN1; ++i) {
                const int r = wn * (C1 / WNc) + i * 16 + fr;
                wb[i] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN1; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) Mma<T>::run(acc1[i][j], wb[i], xa[j]);
        }
        if (more) wait_vmcnt<LPT>(); else wait_vmcnt<0>();     // tile kt+1 complete; the one just requested may stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur == 2 ? 0 : cur + 1;
    }
    // the ring is dead (the loop ended with a barrier): first weight slice on its way, then park relu(conv + bias) as the
    // phase-B operand: row r, channel cc -> block cc / 64, 16-byte chunk (cc % 64) / 8 swizzled by the row like a DMA'd tile
    stage_w3(0);
#pragma unroll
    for (int i = 0; i < TN1; ++i) {
        const int cc = wn * (C1 / WNc) + i * 16 + fg * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + cc);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int r = wm * (BM / WMc) + j * 16 + fr;
            T o4[4];
            Tr<T>::st(&o4[0], relu_f(acc1[i][j][0] + b4.x));
            Tr<T>::st(&o4[1], relu_f(acc1[i][j][1] + b4.y));
            Tr<T>::st(&o4[2], relu_f(acc1[i][j][2] + b4.z));
            Tr<T>::st(&o4[3], relu_f(acc1[i][j][3] + b4.w));
            char* dst = smem + (cc >> 6) * (BM * 128) + r * 128 + ((((cc & 63) >> 3) ^ (r & 7)) << 4) + (cc & 7) * 2;
            *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(o4);
        }
    }
    // ---- phase B: 1x1 expansion, NT slices of BM x 128
    IGemmDev pe = p;
    pe.bias = q.b3; pe.res = q.res; pe.y = q.y; pe.N = q.C3; pe.ldy = q.ldy3; pe.ldr = q.ldr3;
    pe.act = ACT_RELU; pe.out_f32 = 0; pe.gn_cg = 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        wait_vmcnt<0>();                                   // this slice's weights have landed
        __syncthreads();                                   // ... for every wave; the parked tile is complete; the image is free
        f32x4 acc2[TN2][TM];
#pragma unroll
        for (int i = 0; i < TN2; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt) {
            const char* sa = smem + kt * (BM * 128);
            const char* sb = smem + T_BYTES + kt * (128 * 128);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 xa[TM], wb[TN2];
                const int chunk = ks * 4 + fg;
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int r = wm * (BM / WMc) + j * 16 + fr;
                    xa[j] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((chunk ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN2; ++i) {
                    const int r = wn * 32 + i * 16 + fr;
                    wb[i] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN2; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) Mma<T>::run(acc2[i][j], wb[i], xa[j]);
            }
        }
        if (nt + 1 < NT) {
            __syncthreads();                               // every wave has read this slice's weights
            stage_w3(nt + 1);
        }
        igemm_epilogue_split<T, BM, 128, NW, WMc, E_NP, 2>(pe, acc2, smem + T_BYTES + W3_BYTES, m0, nt * 128, tid, wm, wn, fr, fg, rpre[nt], true);
    }
}



// bneck23_kernel plus the NEXT block's 1x1 reduction (4*C1 -> CN channels, + bias + ReLU) computed from the output tile while it is
// still on the CU: that conv would otherwise be a launch of its own whose only HBM-heavy operand is the map this kernel has just
// written.  Phase B runs in 64-channel output slices here: the slice's final (rounded) values are also parked in LDS as an
// MFMA operand block, and acc3 += slice x W1[:, slice] after every slice -- the same k order as the stand-alone conv, whose input
// is exactly those rounded values: bit-identical.  LDS: parked tile + weight slice + image slab + slice block + W1 slice = 65-74 KB.
struct Bneck231Dev {
    BneckDev t;
    const char* w1; const float* b1; char* o1;     // next block's reduction: w1 [CN][4*C1], output [M][ldo]
    int ldo; unsigned w1_bytes;
    long long g_w1, g_b1, g_o1;
    // KD > 0: the block's 1x1 down-sample conv folded into the expansion GEMM (weights K-concatenated [W3 | Wds], bias b3 + bds): the
    // block input xd [B,H,W,*] (KD*64 channels at pixel stride xdC) is a second operand block, no identity tensor exists
    const char* xd; int xdC; unsigned xd_bytes; long long g_xd;
    // xcd_tiles > 0 (two groups, 1-D grid): the hi | lo pair is split over the XCDs -- workgroup id -> XCD id & 7 (dispatch order, observed), group =
    // XCD / 4, pixel tile = (id / 8) * 4 + XCD % 4 -- so that an XCD's L2 streams ONE group's weights (2.2 MB per group at 256 mid channels;
    // both groups' 4.4 MB do not fit the 4 MB of an XCD's L2: 104 us per launch in the step against 83 us with one group's weights)
    int xcd_tiles;
};

template <typename T, int BM, int C1, int CN, int KD = 0>
__global__ __launch_bounds__(512, 4) void bneck231_kernel(Bneck231Dev qq) {     // 4 waves per SIMD = two workgroups per CU
    BneckDev& q = qq.t;
    constexpr int NW = 8, WMc = 2, WNc = 4, CH = 8, BK = 64, SW = 64;
    constexpr int TM = BM / WMc / 16;
    constexpr int TN1 = C1 / WNc / 16;
    constexpr int TN2 = SW / WNc / 16;             // 1: 16 of the slice's 64 channels per wave
    constexpr int TN3 = CN / WNc / 16;             // reduction output: CN / 4 channels per wave
    constexpr int A_IT = BM / 8 / NW, B_IT = C1 / 8 / NW;
    constexpr int TILE_BYTES = (BM + C1) * 128;
    constexpr int KT1 = C1 / BK;
    constexpr int NT = 4 * C1 / SW;
    constexpr int KTB = KT1 + KD;                  // K blocks of phase B: the parked tile, then the down-sample input
    constexpr int T_BYTES = KTB * BM * 128, W3_BYTES = KTB * SW * 128;
    constexpr int E_RPP = 64 * NW / (SW / 8), E_NP = BM / E_RPP;          // 64 rows per pass
    constexpr int ESPLIT = BM / E_RPP;                                     // image slab = one pass = 64 rows
    constexpr int IMG_BYTES = 64 * (128 + 4) * 4 / 2 + 1024;               // >= 64 x (64+4) f32 and >= the reduction's slabs
    constexpr int YS_OFF = T_BYTES + W3_BYTES + IMG_BYTES, YS_BYTES = BM * 128;
    constexpr int W1_OFF = YS_OFF + YS_BYTES;
    constexpr int R_RPP = 64 * NW / (CN / 8), R_SPLIT = BM / R_RPP;        // reduction epilogue: image slabs of one pass
    static_assert(A_IT >= 1 && B_IT >= 1 && TN1 >= 1 && TN3 >= 1 && sizeof(T) == 2, "bneck231 tile");
    static_assert(R_RPP * (CN + 4) * 4 <= IMG_BYTES && 64 * (SW + 4) * 4 <= IMG_BYTES, "image slab");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    IGemmDev& p = q.a;
    if (p.groups > 1) {
        const long long g = blockIdx.y;
        p.x += g * p.g_x * 2;
        p.w += g * p.g_w * 2;
        p.bias += g * p.g_b;
        q.w3 += g * q.g_w3 * 2;
        q.b3 += g * q.g_b3;
        q.res += g * q.g_y3 * 2;
        q.y += g * q.g_y3 * 2;
        qq.w1 += g * qq.g_w1 * 2;
        qq.b1 += g * qq.g_b1;
        qq.o1 += g * qq.g_o1 * 2;
        if (KD) qq.xd += g * qq.g_xd * 2;
    }
    const int m0 = blockIdx.x * BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNc, wn = wave % WNc;
    const int rin = lane >> 3;
    const int c = (lane & 7) ^ rin;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    uint4 rpre[NT][E_NP] = {};
    if constexpr (KD == 0) {
        const int n = (tid % (SW / 8)) * 8;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int pass = 0; pass < E_NP; ++pass) {
                const int m = m0 + pass * E_RPP + tid / (SW / 8);
                rpre[nt][pass] = make_uint4(0u, 0u, 0u, 0u);
                if (m < p.M) rpre[nt][pass] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(q.res) + (size_t)m * q.ldr3 + nt * SW + n);
            }
    }

    int a_pix[A_IT], a_iy0[A_IT], a_ix0[A_IT];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + (wave + NW * i) * 8 + rin;
        if (m < p.M) {
            const int b = m / HoWo;
            const int rem = m - b * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            a_pix[i] = b * p.H * p.W;
            a_iy0[i] = oy * p.stride - p.pad;
            a_ix0[i] = ox * p.stride - p.pad;
        } else {
            a_pix[i] = -1; a_iy0[i] = 0; a_ix0[i] = 0;
        }
    }
    const v4i_t rx = make_rsrc(p.x, p.x_bytes);
    const v4i_t rw = make_rsrc(p.w, p.w_bytes);
    const v4i_t rw3 = make_rsrc(q.w3, q.w3_bytes);
    const v4i_t rw1 = make_rsrc(qq.w1, qq.w1_bytes);
    const v4i_t rxd = make_rsrc(qq.xd, KD ? qq.xd_bytes : 0u);

    auto stage = [&](int kt, int buf) {
        const unsigned sa = lds_base + buf * TILE_BYTES;
        const unsigned sb = sa + BM * 128;
        const int k = kt * BK + c * CH;
        const int khw = k >> p.cin_shift;
        const int ci = k & (p.Cin - 1);
        const int kh = (khw * p.kw_rcp) >> 16;
        const int kw = khw - kh * p.KW;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
            const bool ok = (k < p.K) & (a_pix[i] >= 0) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            const unsigned off = (unsigned)((a_pix[i] + iy * p.W + ix) * p.xC + ci) * 2u;
            dma16(sa + (wave + NW * i) * 1024, ok ? off : 0xFFFFFFFFu, rx);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = (wave + NW * i) * 8 + rin;
            const unsigned off = (unsigned)(n * p.Kp + k) * 2u;
            dma16(sb + (wave + NW * i) * 1024, (k < p.Kp) ? off : 0xFFFFFFFFu, rw);
        }
    };
    // slice nt: expansion weights (KT1 blocks of [64 channels][64 k]) and the reduction's K slice ([CN channels][64 k])
    // the 1x1 down-sample conv's input rows: pixel (oy * stride, ox * stride) of the block input, KD blocks of 64 channels
    auto stage_xd = [&]() {
#pragma unroll
        for (int kt = 0; kt < KD; ++kt)
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const unsigned off = (unsigned)((a_pix[i] + (a_iy0[i] + p.pad) * p.W + a_ix0[i] + p.pad) * qq.xdC + kt * BK + c * CH) * 2u;
                dma16(lds_base + (KT1 + kt) * (BM * 128) + (wave + NW * i) * 1024, a_pix[i] >= 0 ? off : 0xFFFFFFFFu, rxd);
            }
    };
    auto stage_w3 = [&](int nt) {
#pragma unroll
        for (int kt = 0; kt < KTB; ++kt) {
            const int n = nt * SW + wave * 8 + rin;                        // 64 rows = 8 waves x 8
            const unsigned off = (unsigned)(n * q.Kp3 + kt * BK + c * CH) * 2u;
            dma16(lds_base + T_BYTES + kt * (SW * 128) + wave * 1024, off, rw3);
        }
    };
    auto stage_w1 = [&](int nt) {
#pragma unroll
        for (int i = 0; i < CN / 8 / NW; ++i) {
            const int n = (wave + NW * i) * 8 + rin;
            const unsigned off = (unsigned)(n * (4 * C1) + nt * SW + c * CH) * 2u;
            dma16(lds_base + W1_OFF + (wave + NW * i) * 1024, off, rw1);
        }
    };

    const int fr = lane & 15;
    const int fg = lane >> 4;
    // ---- phase A: 3x3 conv, BM x C1 (as in bneck23_kernel)
    f32x4 acc1[TN1][TM];
#pragma unroll
    for (int i = 0; i < TN1; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = (p.K + BK - 1) / BK;
    constexpr int LPT = A_IT + B_IT;
    stage(0, 0);
    if (nk > 1) { stage(1, 1); wait_vmcnt<LPT>(); } else { wait_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 2 < nk;
        if (more) stage(kt + 2, cur == 0 ? 2 : cur - 1);
        const char* sa = smem + cur * TILE_BYTES;
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 xa[TM], wb[TN1];
            const int chunk = ks * 4 + fg;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int r = wm * (BM / WMc) + j * 16 + fr;
                xa[j] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN1; ++i) {
                const int r = wn * (C1 / WNc) + i * 16 + fr;
                wb[i] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN1; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) Mma<T>::run(acc1[i][j], wb[i], xa[j]);
        }
        if (more) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur == 2 ? 0 : cur + 1;
    }
    stage_w3(0);
    stage_w1(0);
    if constexpr (KD > 0) stage_xd();
#pragma unroll
    for (int i = 0; i < TN1; ++i) {
        const int cc = wn * (C1 / WNc) + i * 16 + fg * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + cc);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int r = wm * (BM / WMc) + j * 16 + fr;
            T o4[4];
            Tr<T>::st(&o4[0], relu_f(acc1[i][j][0] + b4.x));
            Tr<T>::st(&o4[1], relu_f(acc1[i][j][1] + b4.y));
            Tr<T>::st(&o4[2], relu_f(acc1[i][j][2] + b4.z));
            Tr<T>::st(&o4[3], relu_f(acc1[i][j][3] + b4.w));
            char* dst = smem + (cc >> 6) * (BM * 128) + r * 128 + ((((cc & 63) >> 3) ^ (r & 7)) << 4) + (cc & 7) * 2;
            *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(o4);
        }
    }
    // ---- phase B: NT slices of BM x 64 output channels, each followed by its K slice of the next block's reduction
    IGemmDev pe = p;
    pe.bias = q.b3; pe.res = KD ? nullptr : q.res; pe.y = q.y; pe.N = q.C3; pe.ldy = q.ldy3; pe.ldr = q.ldr3;
    pe.act = ACT_RELU; pe.out_f32 = 0; pe.gn_cg = 0;
    f32x4 acc3[TN3][TM];
#pragma unroll
    for (int i = 0; i < TN3; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        wait_vmcnt<0>();                                   // this slice's weights (both matrices) have landed
        __syncthreads();                                   // ... for every wave; parked tile complete; image and slice block free
        f32x4 acc2[TN2][TM];
#pragma unroll
        for (int i = 0; i < TN2; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KTB; ++kt) {
            const char* sa = smem + kt * (BM * 128);
            const char* sb = smem + T_BYTES + kt * (SW * 128);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 xa[TM], wb[TN2];
                const int chunk = ks * 4 + fg;
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int r = wm * (BM / WMc) + j * 16 + fr;
                    xa[j] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((chunk ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN2; ++i) {
                    const int r = wn * (SW / WNc) + i * 16 + fr;
                    wb[i] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN2; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) Mma<T>::run(acc2[i][j], wb[i], xa[j]);
            }
        }
        // epilogue of the slice: to HBM and, rounded, into the slice block (its first barrier also says: every wave is done with the weights)
        igemm_epilogue_split<T, BM, SW, NW, WMc, E_NP, ESPLIT>(pe, acc2, smem + T_BYTES + W3_BYTES, m0, nt * SW, tid, wm, wn, fr, fg, rpre[nt], KD == 0,
                                                               smem + YS_OFF);
        if (nt + 1 < NT) stage_w3(nt + 1);                 // (after the epilogue's barriers: the expansion weights are dead)
        __syncthreads();                                   // slice block complete
        {
            const char* sa = smem + YS_OFF;
            const char* sb = smem + W1_OFF;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 xa[TM], wb[TN3];
                const int chunk = ks * 4 + fg;
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int r = wm * (BM / WMc) + j * 16 + fr;
                    xa[j] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((chunk ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN3; ++i) {
                    const int r = wn * (CN / WNc) + i * 16 + fr;
                    wb[i] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN3; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) Mma<T>::run(acc3[i][j], wb[i], xa[j]);
            }
        }
        if (nt + 1 < NT) {
            __syncthreads();                               // every wave has read this K slice of the reduction weights
            stage_w1(nt + 1);
        }
    }
    // the reduction's own epilogue (bias + ReLU), through the same image region
    IGemmDev pr = p;
    pr.bias = qq.b1; pr.res = nullptr; pr.y = qq.o1; pr.N = CN; pr.ldy = qq.ldo; pr.ldr = qq.ldo;
    pr.act = ACT_RELU; pr.out_f32 = 0; pr.gn_cg = 0;
    __syncthreads();
    const uint4 none[1] = {make_uint4(0u, 0u, 0u, 0u)};
    igemm_epilogue_split<T, BM, CN, NW, WMc, 1, R_SPLIT>(pr, acc3, smem + T_BYTES + W3_BYTES, m0, 0, tid, wm, wn, fr, fg, none, false);
}

// bneck231_kernel with REGISTER epilogues (round 3).  The f32 LDS image of the shared epilogue -- write the accumulators, barrier, read
// rows back, barrier, per 64-row slab -- was most of this kernel's 42 barriers per pixel tile and the reason it ran at 0.44 of the HBM rate
// its bytes ask for.  Here phase B re-partitions the 8 waves as 4 (pixels) x 2 (channels): a wave owns BM/4 pixels x 32 channels of a 64-wide
// output slice, i.e. PAIRS of adjacent 16-channel accumulator tiles, and swap_pair (v_permlane16_swap_b32, dev.h) turns a pair into 8
// consecutive channels of one pixel per lane.  Bias + identity + ReLU + the one rounding then happen in registers and the lane stores its
// 16 bytes twice: to y, and into the slice block in LDS that the next block's reduction reads as its MFMA operand.  Per slice that leaves TWO
// barriers (weights landed / slice block complete); the reduction's weight slices are double-buffered so that nothing else has to be waited
// for.  Same MFMA sequence, same f32 operations in the same order as bneck231_kernel: bit-identical to it and to the stand-alone launches.
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// NWB > 0: HALO phase A.  A lone workgroup's phase profile (B = 8: one tile per CU, every byte out of L2 / the Infinity Cache) says what a
// pixel tile's 12 us are: 3 us until the first barrier and 0.43 us per K iteration of the 3x3 conv for 8 MFMAs (0.05 us) -- the iteration
// waits for its 24 KB tap tile, requested two iterations earlier through a 3-deep ring, and an L2 -> LDS request takes ~0.85 us to land: a
// latency chain, L / 2 per tap, that no amount of HBM bandwidth shortens (the launch takes the same 21-23 us per round of resident
// workgroups whether its maps stream from HBM or sit in the Infinity Cache).  Here the tile's input rows are staged ONCE, with their
// one-pixel halo (a tile is whole image rows: (R + 2) x (W + 2) pixels, zero-filled outside the image by the buffer range check), the nine
// taps read shifted windows of that block, and only the 8 KB weight tile of a tap still streams -- through an NWB-deep ring whose first
// NWB - 1 tiles are requested with the halo block, so that all nine are in flight within the first four iterations.  The identity rows are
// requested right BEHIND the last weight tile instead of first: vmcnt retires in order, so with them in front every wait of phase A also
// waited for the tile's 64 KB of identity (an earlier halo form with a 4-deep ring and the identity in front measured no gain for exactly
// that reason).  MFMA sequence and f32 operations unchanged: bit-identical to the NWB = 0 form.  Needs stride 1, C1 = 64, a tile of whole rows
// of one image, W a multiple of 16 (launch_bneck23 checks; else NWB = 0).
// W1B = 1: ONE buffer for the reduction's weight slice instead of two (the 128-pixel tile of the folded-down-sample block needs the 8 KB: its
// parked tile is two K blocks deep).  The slice is then requested behind the top-of-slice barrier -- every wave has left the previous
// reduction -- and waited for at the slice-block barrier, a whole expansion + epilogue later.
template <typename T, int BM, int C1, int CN, int KD = 0, bool PROF = false, int NWB = 0, int W1B = 2>
__global__ __launch_bounds__(512, (C1 >= 256 || (BM == 128 && C1 == 128)) ? 2 : 4) void bneck231r_kernel(Bneck231Dev qq) {
    // phase timing (HCM_IGEMM_PROF=1 builds, read through hcm_debug_igemm_prof): per-wave cycle totals [0] prologue up to the first barrier,
    // [1] phase A K loop, [2] park + top-of-slice waits and barriers, [3] expansion MFMAs, [4] register epilogues, [5] slice-block barrier +
    // reduction MFMAs + final epilogue, [6] waves, [7] K iterations
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
    if constexpr (PROF) t_prev = prof_now();
    auto lap = [&](int slot) {
        if constexpr (PROF) { const unsigned long long t = prof_now(); pt[slot] += t - t_prev; t_prev = t; }
    };
    BneckDev& q = qq.t;
    // phase A wave grid: 2 (pixels) x 4 (channels); the halo form with its 64 mid channels takes 4 x 2 -- 32 pixels x 32 channels per wave, four
    // fragment reads per four MFMAs instead of five: the K loop there is bound by the LDS read rate (80 KB of fragments per tap and tile)
    constexpr int NW = 8, WMc = (NWB > 0 && C1 == 64) ? 4 : 2, WNc = (NWB > 0 && C1 == 64) ? 2 : 4, CH = 8, BK = 64, SW = 64;
    constexpr int TM = BM / WMc / 16;              // phase A: BM / WMc pixels x C1 / WNc channels per wave
    constexpr int TN1 = C1 / WNc / 16;
    constexpr int WM2 = 4, WN2 = 2;                // phase B: BM / 4 pixels x 32 channels (of a slice) or CN / 2 channels (reduction) per wave
    constexpr int TMB = BM / WM2 / 16;
    constexpr int TN2 = SW / WN2 / 16;             // 2: one pair
    constexpr int TN3 = CN / WN2 / 16;             // 2 or 4: one or two pairs
    constexpr int A_IT = BM / 8 / NW, B_IT = C1 / 8 / NW;
    constexpr int TILE_BYTES = (BM + C1) * 128;
    constexpr int KT1 = C1 / BK;
    constexpr int NT = 4 * C1 / SW;
    constexpr int KTB = KT1 + KD;
    constexpr int T_BYTES = KTB * BM * 128, W3_BYTES = KTB * SW * 128;
    constexpr int YS_OFF = T_BYTES + W3_BYTES, YS_BYTES = BM * 128;
    constexpr int W1_OFF = YS_OFF + YS_BYTES, W1_BYTES = CN * 128;      // two buffers
    constexpr int BIAS_OFF = W1_OFF + W1B * W1_BYTES;                   // f32: the expansion's 4 * C1 biases, then the reduction's CN
    static_assert(A_IT >= 1 && B_IT >= 1 && TN1 >= 1 && TMB >= 1 && TN2 == 2 && TN3 % 2 == 0 && sizeof(T) == 2, "bneck231r tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    IGemmDev& p = q.a;
    int tile_id = blockIdx.x;
    if (p.groups > 1) {
        long long g = blockIdx.y;
        if (qq.xcd_tiles > 0) {
            const int xcd = blockIdx.x & 7;
            g = xcd >> 2;
            tile_id = (blockIdx.x >> 3) * 4 + (xcd & 3);
            if (tile_id >= qq.xcd_tiles) return;
        }
        p.x += g * p.g_x * 2;
        p.w += g * p.g_w * 2;
        p.bias += g * p.g_b;
        q.w3 += g * q.g_w3 * 2;
        q.b3 += g * q.g_b3;
        q.res += g * q.g_y3 * 2;
        q.y += g * q.g_y3 * 2;
        qq.w1 += g * qq.g_w1 * 2;
        qq.b1 += g * qq.g_b1;
        qq.o1 += g * qq.g_o1 * 2;
        if (KD) qq.xd += g * qq.g_xd * 2;
    }
    const int m0 = tile_id * BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNc, wn = wave % WNc;
    const int wm2 = wave / WN2, wn2 = wave % WN2;
    const int rin = lane >> 3;
    const int c = (lane & 7) ^ rin;
    const int fr = lane & 15;
    const int fg = lane >> 4;
    const int coff = (fg & 1) * 16 + (fg >> 1) * 8;        // the lane's 8 channels inside a 32-channel pair (swap_pair)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // identity rows of every output slice, requested first: they arrive while phase A runs (16 B per lane, pixel tile and slice)
    // ID_STREAM (256 mid channels: 16 slices x TMB rows = 128 registers if held at once): the identity rows of slice nt + 1 are requested at the
    // top of slice nt instead, behind that slice's reduction-weight request, into a two-slice register ring -- a whole slice (~1.5 us) to land
    constexpr bool ID_STREAM = C1 >= 256 && KD == 0;
    static_assert(!ID_STREAM || W1B == 1, "the streamed identity's wait counts assume the one-buffer reduction-weight form");
    uint4 rpre[ID_STREAM ? 2 : NT][TMB] = {};
    auto load_identity_slice = [&](int nt, int slot) {
#pragma unroll
        for (int j = 0; j < TMB; ++j) {
            const int m = min(m0 + wm2 * (BM / WM2) + j * 16 + fr, p.M - 1);
            rpre[slot][j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(q.res) + (size_t)m * q.ldr3 + nt * SW + wn2 * 32 + coff);
        }
    };
    auto load_identity = [&]() {
        if constexpr (KD == 0 && !ID_STREAM) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < TMB; ++j) {
                    // (rows past M read row M - 1 and are never stored: the loads are unconditional so that every wave has exactly NT * TMB of
                    //  them in its queue -- the counted waits of phase A rely on it)
                    const int m = min(m0 + wm2 * (BM / WM2) + j * 16 + fr, p.M - 1);
                    rpre[nt][j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(q.res) + (size_t)m * q.ldr3 + nt * SW + wn2 * 32 + coff);
                }
        }
    };
    // both bias vectors, one element per thread: parked in LDS behind phase A, so that nothing in phase B is a global load whose wait would
    // also wait for the output stores in front of it (vmcnt retires in order)
    const float b3v = tid < 4 * C1 ? q.b3[tid] : 0.f;
    const float b3w = (4 * C1 > 512 && tid + 512 < 4 * C1) ? q.b3[tid + 512] : 0.f;      // 256 mid channels: 1024 expansion biases, two per thread
    const float b1v = tid < CN ? qq.b1[tid] : 0.f;

    int a_pix[A_IT], a_iy0[A_IT], a_ix0[A_IT];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + (wave + NW * i) * 8 + rin;
        if (m < p.M) {
            const int b = m / HoWo;
            const int rem = m - b * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            a_pix[i] = b * p.H * p.W;
            a_iy0[i] = oy * p.stride - p.pad;
            a_ix0[i] = ox * p.stride - p.pad;
        } else {
            a_pix[i] = -1; a_iy0[i] = 0; a_ix0[i] = 0;
        }
    }
    const v4i_t rx = make_rsrc(p.x, p.x_bytes);
    const v4i_t rw = make_rsrc(p.w, p.w_bytes);
    const v4i_t rw3 = make_rsrc(q.w3, q.w3_bytes);
    const v4i_t rw1 = make_rsrc(qq.w1, qq.w1_bytes);
    const v4i_t rxd = make_rsrc(qq.xd, KD ? qq.xd_bytes : 0u);

    auto stage = [&](int kt, int buf) {
        const unsigned sa = lds_base + buf * TILE_BYTES;
        const unsigned sb = sa + BM * 128;
        const int k = kt * BK + c * CH;
        const int khw = k >> p.cin_shift;
        const int ci = k & (p.Cin - 1);
        const int kh = (khw * p.kw_rcp) >> 16;
        const int kw = khw - kh * p.KW;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
            const bool ok = (k < p.K) & (a_pix[i] >= 0) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            const unsigned off = (unsigned)((a_pix[i] + iy * p.W + ix) * p.xC + ci) * 2u;
            dma16(sa + (wave + NW * i) * 1024, ok ? off : 0xFFFFFFFFu, rx);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = (wave + NW * i) * 8 + rin;
            const unsigned off = (unsigned)(n * p.Kp + k) * 2u;
            dma16(sb + (wave + NW * i) * 1024, (k < p.Kp) ? off : 0xFFFFFFFFu, rw);
        }
    };
    auto stage_xd = [&]() {
#pragma unroll
        for (int kt = 0; kt < KD; ++kt)
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const unsigned off = (unsigned)((a_pix[i] + (a_iy0[i] + p.pad) * p.W + a_ix0[i] + p.pad) * qq.xdC + kt * BK + c * CH) * 2u;
                dma16(lds_base + (KT1 + kt) * (BM * 128) + (wave + NW * i) * 1024, a_pix[i] >= 0 ? off : 0xFFFFFFFFu, rxd);
            }
    };
    auto stage_w3 = [&](int nt) {
#pragma unroll
        for (int kt = 0; kt < KTB; ++kt) {
            const int n = nt * SW + wave * 8 + rin;
            const unsigned off = (unsigned)(n * q.Kp3 + kt * BK + c * CH) * 2u;
            dma16(lds_base + T_BYTES + kt * (SW * 128) + wave * 1024, off, rw3);
        }
    };
    auto stage_w1 = [&](int nt) {
#pragma unroll
        for (int i = 0; i < CN / 8 / NW; ++i) {
            const int n = (wave + NW * i) * 8 + rin;
            const unsigned off = (unsigned)(n * (4 * C1) + nt * SW + c * CH) * 2u;
            dma16(lds_base + W1_OFF + (nt % W1B) * W1_BYTES + (wave + NW * i) * 1024, off, rw1);
        }
    };

    // ---- phase A: 3x3 conv, BM x C1 (as in bneck23_kernel)
    f32x4 acc1[TN1][TM];
#pragma unroll
    for (int i = 0; i < TN1; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = (p.K + BK - 1) / BK;
    constexpr int LPT = A_IT + B_IT;
    if constexpr (NWB > 0) {
        // weight UNITS of 8 KB: a whole tap (64 output rows x 64 k, 128-byte rows) with 64 mid channels; with 128 mid channels a quarter tap
        // (128 output rows x 32 k, 64-BYTE rows: one MFMA K step) -- 16 KB tiles would leave room for a 2-deep ring only beside the 35 KB halo block
        constexpr int UK = C1 == 64 ? 64 : 32;               // k per unit
        constexpr int UPT = C1 / UK;                         // units per tap: 1 / 4
        constexpr int NK = 9 * UPT, D = NWB - 1, NID = (KD == 0 && !ID_STREAM) ? NT * TMB : 0;
        static_assert(NWB >= 2 && (C1 == 64 || C1 == 128) && C1 * UK * 2 == 8192, "halo phase A: 8 KB weight units");
        constexpr int ND0 = D < NK ? D : NK;                 // weight units requested with the halo block
        constexpr int ID_AT = NK - 1 - D;                    // the iteration that requests the last weight unit (< 0: all went out up front)
        // halo geometry: the tile is R whole rows of one image's W-wide map; halo rows hy = 0 .. R + 1 <-> input rows oy0 - 1 + hy, columns
        // hx = 0 .. W + 1 <-> input columns hx - 1; HRP (a multiple of 8) LDS rows of 128 B per 64-channel block
        const int Wm = p.W, Wp = Wm + 2, R = BM / Wm, HR = (R + 2) * Wp, HRP = (HR + 7) & ~7;
        const int img = m0 / HoWo, oy0 = (m0 - img * HoWo) / Wm, pix0 = img * p.H * p.W;
        const unsigned WR_OFF = (unsigned)(KT1 * HRP * 128);
        {
            const int n_instr = KT1 * HRP / 8;
            for (int i = wave; i < n_instr; i += NW) {
                const int L = i * 8 + rin;
                const int blk = L / HRP, hr = L - blk * HRP;
                const int hy = hr / Wp, hx = hr - hy * Wp;
                const int iy = oy0 - 1 + hy, ix = hx - 1;
                const bool ok = (hr < HR) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                const unsigned off = (unsigned)((pix0 + iy * p.W + ix) * p.xC + blk * BK + c * CH) * 2u;
                dma16(__builtin_amdgcn_readfirstlane(lds_base + i * 1024), ok ? off : 0xFFFFFFFFu, rx);
            }
        }
        // 64-byte weight rows: 16-byte position = chunk ^ g(row quad), g = (0, 3, 2, 1) -- with ds_read_b128's lane groups (lanes {0-3, 12-15} of one
        // K group with lanes {4-11} of the next) the 16 lanes of a group then hit 16 distinct 16-byte slots of the 256-byte bank window
        auto stage_w = [&](int u, int slot) {
            if constexpr (C1 == 64) {
                const int n = wave * 8 + rin;
                const unsigned off = (unsigned)(n * p.Kp + u * BK + c * CH) * 2u;
                dma16(__builtin_amdgcn_readfirstlane(lds_base + WR_OFF + slot * 8192 + wave * 1024), off, rw);
            } else {
                const int row = lane >> 2, n = wave * 16 + row;
                const int src = (lane & 3) ^ ((4 - (row >> 2)) & 3);
                const unsigned off = (unsigned)(n * p.Kp + u * UK + src * CH) * 2u;
                dma16(__builtin_amdgcn_readfirstlane(lds_base + WR_OFF + slot * 8192 + wave * 1024), off, rw);
            }
        };
        int hr0[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int r = wm * (BM / WMc) + j * 16 + fr;
            const int ry = r / Wm;
            hr0[j] = ry * Wp + (r - ry * Wm);
        }
        auto unit_mma = [&](auto U) {
            constexpr int u = decltype(U)::value;
            constexpr int tap = u / UPT, kh = tap / 3, kw = tap - kh * 3;
            const int dhr = kh * Wp + kw;
            const char* sb = smem + WR_OFF + (u % NWB) * 8192;
            if constexpr (C1 == 64) {
                const char* sa = smem;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    uint4 xa[TM], wb[TN1];
                    const int chunk = ks * 4 + fg;
#pragma unroll
                    for (int j = 0; j < TM; ++j) {
                        const int hr = hr0[j] + dhr;
                        xa[j] = *reinterpret_cast<const uint4*>(sa + hr * 128 + ((chunk ^ (hr & 7)) << 4));
                    }
#pragma unroll
                    for (int i = 0; i < TN1; ++i) {
                        const int r = wn * (C1 / WNc) + i * 16 + fr;
                        wb[i] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
                    }
#pragma unroll
                    for (int i = 0; i < TN1; ++i)
#pragma unroll
                        for (int j = 0; j < TM; ++j) Mma<T>::run(acc1[i][j], wb[i], xa[j]);
                }
            } else {
                constexpr int blk = (u % UPT) / 2, ks = u % 2;       // 64-channel block of the halo rows, 32-k half inside it
                const char* sa = smem + blk * (HRP * 128);
                uint4 xa[TM], wb[TN1];
                const int chunk = ks * 4 + fg;
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int hr = hr0[j] + dhr;
                    xa[j] = *reinterpret_cast<const uint4*>(sa + hr * 128 + ((chunk ^ (hr & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN1; ++i) {
                    const int r = wn * (C1 / WNc) + i * 16 + fr;
                    wb[i] = *reinterpret_cast<const uint4*>(sb + r * 64 + ((fg ^ ((4 - (r >> 2)) & 3)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN1; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) Mma<T>::run(acc1[i][j], wb[i], xa[j]);
            }
        };
#pragma unroll
        for (int t = 0; t < ND0; ++t) stage_w(t, t);
        if constexpr (ID_AT < 0) load_identity();
        wait_vmcnt<(ND0 - 1) + (ID_AT < 0 ? NID : 0)>();      // the halo block and weight unit 0 have landed
        __builtin_amdgcn_s_barrier();
        lap(0);
        static_for<0, NK>([&](auto U) {
            constexpr int u = decltype(U)::value;
            if constexpr (u + D < NK) stage_w(u + D, (u + D) % NWB);        // into the slot unit u - 1 was read from (every wave has left it)
            if constexpr (u == ID_AT) load_identity();                      // behind the last operand request
            unit_mma(U);
            if constexpr (u + 1 < NK) {
                // weight unit u + 1 must have landed; what was requested after it (younger units, the identity rows) stays in flight
                constexpr int last = u + D < NK - 1 ? u + D : NK - 1;
                wait_vmcnt<(last - (u + 1)) + (u >= ID_AT ? NID : 0)>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        });
    } else {
    // the 3-deep ring of whole tap tiles.  The identity rows are requested right behind the LAST tile request (iteration nk - 3): in front,
    // every wait of the loop would also wait for them (vmcnt retires in order) -- the first tile then comes up a whole identity fetch late
    constexpr int NID = (KD == 0 && !ID_STREAM) ? NT * TMB : 0;
    stage(0, 0);
    if (nk > 1) { stage(1, 1); if (nk == 2) load_identity(); }
    else load_identity();
    if (nk > 2) wait_vmcnt<LPT>(); else if (nk == 2) wait_vmcnt<LPT + NID>(); else wait_vmcnt<NID>();
    __builtin_amdgcn_s_barrier();
    lap(0);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 2 < nk;
        if (more) stage(kt + 2, cur == 0 ? 2 : cur - 1);
        if (kt + 3 == nk) load_identity();
        const char* sa = smem + cur * TILE_BYTES;
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 xa[TM], wb[TN1];
            const int chunk = ks * 4 + fg;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int r = wm * (BM / WMc) + j * 16 + fr;
                xa[j] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN1; ++i) {
                const int r = wn * (C1 / WNc) + i * 16 + fr;
                wb[i] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN1; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) Mma<T>::run(acc1[i][j], wb[i], xa[j]);
        }
        // tile kt + 1 has landed; younger: tile kt + 2 (if any) and, from iteration nk - 3 on, the identity rows
        if (more) { if (kt + 3 == nk) wait_vmcnt<LPT + NID>(); else wait_vmcnt<LPT>(); }
        else if (kt + 1 < nk) wait_vmcnt<NID>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur == 2 ? 0 : cur + 1;
    }
    }
    lap(1);
    stage_w3(0);
    stage_w1(0);
    if constexpr (ID_STREAM) load_identity_slice(0, 0);
    {
        float* sbias = reinterpret_cast<float*>(smem + BIAS_OFF);
        if (tid < 4 * C1) sbias[tid] = b3v;
        if constexpr (4 * C1 > 512) { if (tid + 512 < 4 * C1) sbias[tid + 512] = b3w; }
        if (tid < CN) sbias[4 * C1 + tid] = b1v;
    }
    if constexpr (KD > 0) stage_xd();
#pragma unroll
    for (int i = 0; i < TN1; ++i) {
        const int cc = wn * (C1 / WNc) + i * 16 + fg * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + cc);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int r = wm * (BM / WMc) + j * 16 + fr;
            T o4[4];
            Tr<T>::st(&o4[0], relu_f(acc1[i][j][0] + b4.x));
            Tr<T>::st(&o4[1], relu_f(acc1[i][j][1] + b4.y));
            Tr<T>::st(&o4[2], relu_f(acc1[i][j][2] + b4.z));
            Tr<T>::st(&o4[3], relu_f(acc1[i][j][3] + b4.w));
            char* dst = smem + (cc >> 6) * (BM * 128) + r * 128 + ((((cc & 63) >> 3) ^ (r & 7)) << 4) + (cc & 7) * 2;
            *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(o4);
        }
    }
    // ---- phase B: NT slices of BM x 64 output channels, each followed by its K slice of the next block's reduction
    f32x4 acc3[TN3][TMB];
#pragma unroll
    for (int i = 0; i < TN3; ++i)
#pragma unroll
        for (int j = 0; j < TMB; ++j) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    T* const yp = reinterpret_cast<T*>(q.y);
    // store instructions this wave issues per slice: its row groups j (16 rows each, ascending) that start below M -- wave-uniform
    int n_st = 0;
#pragma unroll
    for (int j = 0; j < TMB; ++j) n_st += (m0 + wm2 * (BM / WM2) + j * 16 < p.M) ? 1 : 0;
    n_st = __builtin_amdgcn_readfirstlane(n_st);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int nb = nt * SW + wn2 * 32 + coff;
        // this slice's weights (both matrices) have landed.  They were requested BEFORE the previous slice's output stores were issued (below),
        // so the wait leaves those TMB stores per lane in flight: a slice never waits for the write acknowledgements of the one before
        // (a wave whose pixel rows lie partly or wholly past M -- the ragged last tile -- issued FEWER store instructions: a row group of 16 whose lanes are
        //  all past M is branched around.  Its count of stores in flight is n_st, not TMB; waiting `vmcnt(TMB)` there would leave the wave's own weight
        //  pieces in flight across the barrier, and the OTHER waves read them: a rare wrong tile, found by round 4's race screen)
        if (nt == 0) wait_vmcnt<0>();
        else if (n_st >= TMB) wait_vmcnt<TMB>();
        else if (TMB > 2 && n_st == 3) wait_vmcnt<3>();
        else if (TMB > 1 && n_st == 2) wait_vmcnt<2>();
        else if (n_st == 1) wait_vmcnt<1>();
        else wait_vmcnt<0>();
        __syncthreads();                                   // ... for every wave; parked tile complete; the previous reduction has left the slice block
        if constexpr (W1B == 1) { if (nt > 0) stage_w1(nt); }
        if constexpr (ID_STREAM) { if (nt + 1 < NT) load_identity_slice(nt + 1, (nt + 1) & 1); }      // younger than this slice's reduction weights
        lap(2);
        const float4 b30 = *reinterpret_cast<const float4*>(smem + BIAS_OFF + nb * 4), b31 = *reinterpret_cast<const float4*>(smem + BIAS_OFF + nb * 4 + 16);
        f32x4 acc2[TN2][TMB];
#pragma unroll
        for (int i = 0; i < TN2; ++i)
#pragma unroll
            for (int j = 0; j < TMB; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KTB; ++kt) {
            const char* sa = smem + kt * (BM * 128);
            const char* sb = smem + T_BYTES + kt * (SW * 128);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 xa[TMB], wb[TN2];
                const int chunk = ks * 4 + fg;
#pragma unroll
                for (int j = 0; j < TMB; ++j) {
                    const int r = wm2 * (BM / WM2) + j * 16 + fr;
                    xa[j] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((chunk ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN2; ++i) {
                    const int r = wn2 * 32 + i * 16 + fr;
                    wb[i] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN2; ++i)
#pragma unroll
                    for (int j = 0; j < TMB; ++j) Mma<T>::run(acc2[i][j], wb[i], xa[j]);
            }
        }
        lap(3);
        // register epilogue of the slice: bias + identity + ReLU, one rounding, 16 bytes per lane to y and into the slice block
        const float bias8[8] = {b30.x, b30.y, b30.z, b30.w, b31.x, b31.y, b31.z, b31.w};
        uint4 oreg[TMB];
#pragma unroll
        for (int j = 0; j < TMB; ++j) {
            float v[8];
            swap_pair(acc2[0][j], acc2[1][j], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias8[e];
            if constexpr (KD == 0) {
                float rr[8];
                cvt_chunk<T>(rpre[ID_STREAM ? (nt & 1) : nt][j], rr);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rr[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e]);
            const uint4 o = pack_chunk<T>(v);
            const int r = wm2 * (BM / WM2) + j * 16 + fr;
            const int ch = (wn2 * 32 + coff) >> 3;
            *reinterpret_cast<uint4*>(smem + YS_OFF + r * 128 + ((ch ^ (r & 7)) << 4)) = o;
            oreg[j] = o;
        }
        lap(4);
        // this slice's reduction weights (requested at the top of the slice; the stores in front of them are a slice old); the streamed identity rows
        // of the next slice, requested behind them, stay in flight
        if constexpr (W1B == 1) { if (ID_STREAM && nt + 1 < NT) wait_vmcnt<TMB>(); else wait_vmcnt<0>(); }
        __syncthreads();                                   // slice block complete; every wave is done with the expansion weights
        if (nt + 1 < NT) { stage_w3(nt + 1); if constexpr (W1B == 2) stage_w1(nt + 1); }      // (the reduction's weight slices alternate between two buffers)
        // ... and only now the slice's 16-byte stores to y: younger than the weight requests above, they stay in flight across the next slice's wait
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < TMB; ++j) {
            const int m = m0 + wm2 * (BM / WM2) + j * 16 + fr;
            if (m < p.M) *reinterpret_cast<uint4*>(yp + (size_t)m * q.ldy3 + nb) = oreg[j];
        }
        {
            const char* sa = smem + YS_OFF;
            const char* sb = smem + W1_OFF + (nt % W1B) * W1_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 xa[TMB], wb[TN3];
                const int chunk = ks * 4 + fg;
#pragma unroll
                for (int j = 0; j < TMB; ++j) {
                    const int r = wm2 * (BM / WM2) + j * 16 + fr;
                    xa[j] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((chunk ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN3; ++i) {
                    const int r = wn2 * (CN / WN2) + i * 16 + fr;
                    wb[i] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((chunk ^ (r & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN3; ++i)
#pragma unroll
                    for (int j = 0; j < TMB; ++j) Mma<T>::run(acc3[i][j], wb[i], xa[j]);
            }
        }
        lap(5);
    }
    // the reduction's own epilogue (bias + ReLU), in registers as well
    T* const op = reinterpret_cast<T*>(qq.o1);
#pragma unroll
    for (int i = 0; i < TN3; i += 2) {
        const int nb = wn2 * (CN / WN2) + i * 16 + coff;
        const float4 b0 = *reinterpret_cast<const float4*>(smem + BIAS_OFF + (4 * C1 + nb) * 4), b1 = *reinterpret_cast<const float4*>(smem + BIAS_OFF + (4 * C1 + nb) * 4 + 16);
        const float bias8[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < TMB; ++j) {
            float v[8];
            swap_pair(acc3[i][j], acc3[i + 1][j], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e] + bias8[e]);
            const int m = m0 + wm2 * (BM / WM2) + j * 16 + fr;
            if (m < p.M) *reinterpret_cast<uint4*>(op + (size_t)m * qq.ldo + nb) = pack_chunk<T>(v);
        }
    }
    if constexpr (PROF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lap(5);
        if (lane == 0) {
            unsigned long long* slot = g_igemm_prof[((blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) & (kProfSlots - 1)];
            for (int i = 0; i < 6; ++i) atomicAdd(&slot[i], pt[i]);
            atomicAdd(&slot[6], 1ull);
            atomicAdd(&slot[7], (unsigned long long)nk);
        }
    }
}

hipError_t igemm_prof_read(unsigned long long* host8, bool reset) {
    std::vector<unsigned long long> h((size_t)kProfSlots * 8);
    hipError_t e = hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_igemm_prof), h.size() * 8);
    if (e != hipSuccess) return e;
    for (int i = 0; i < 8; ++i) host8[i] = 0;
    for (size_t k = 0; k < h.size(); ++k) host8[k & 7] += h[k];
    if (!reset) return hipSuccess;
    std::fill(h.begin(), h.end(), 0ull);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_igemm_prof), h.data(), h.size() * 8);
}
static bool prof_on() { static const bool on = dev_env("HCM_IGEMM_PROF") != nullptr; return on; }

// variant: 0 = register-staged 2-buffer, 1 = LDS-DMA 2-buffer, 2 = LDS-DMA 3-deep ring, 3 = register-staged with two
// register sets (prefetch distance 2), 4 / 5 = variants 1 / 2 with 8 waves per workgroup, 6 / 7 = 4 / 5 with the DMA
// instructions interleaved into the MFMA stream
template <typename T, int BM, int BN>
static hipError_t launch_cfg(IGemmDev d, int variant, hipStream_t s) {
    d.tilesM = (d.M + BM - 1) / BM;
    d.tilesN = (d.N + BN - 1) / BN;
    const int tm8 = (d.tilesM + 7) / 8;
    int grid = tm8 * 8 * d.tilesN;
    static const char* fmap = dev_env("HCM_IGEMM_MAP");
    d.map = fmap ? atoi(fmap) : (((size_t)d.N * d.Kp * sizeof(T) > (2u << 20)) && d.tilesN >= 8 ? 1 : 0);
    if (d.map == 1) {
        if (d.tilesN < 8) d.map = 0;
        else grid = 8 * d.tilesM * ((d.tilesN + 7) / 8);       // every XCD gets ceil(tilesN/8) slots per pixel tile
    }
    size_t lds = ((variant == 2 || variant == 5 || variant >= 7) ? 3 : 2) * (size_t)(BM + BN) * 128;
    const size_t lds_c = (size_t)BM * (BN + 4) * 4 + 1024 + (d.cs_part ? 64 * 8 * 2 * 4 : 0);    // f32 output-tile image of the epilogue (+ fused-GroupNorm statistics, + column-sum slices)
    if (lds_c > lds) lds = lds_c;
    if (variant == 9 || variant == 10) {
        // the DEEP ring (igemm_dma_kernel, NBUF > 3): 6 tiles (variant 9; as many as 160 KB hold) or 4 (variant 10), one workgroup per CU
        constexpr int TB = (BM + BN) * 128;
        constexpr int ND6 = 6 * TB <= 160 * 1024 ? 6 : (5 * TB <= 160 * 1024 ? 5 : 4);
        const int nb = variant == 9 ? ND6 : 4;
        lds = (size_t)nb * TB;
        if (lds_c > lds) lds = lds_c;
        const void* fn;
        int threads;
        if constexpr (BN >= 64 && sizeof(T) == 2) {
            // the pipelined loop (ILV = 3, fragment reads of the next K half beside the current half's MFMAs) is the default; HCM_DEEP_ILV3=0 of the
            // development build selects the read-then-multiply loop (ILV = 1) for the toggle test
            static const bool ilv3 = !(dev_env("HCM_DEEP_ILV3") != nullptr && atoi(dev_env("HCM_DEEP_ILV3")) == 0);
            if (ilv3)
                fn = nb == ND6 ? reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, ND6, 8, 2, false, 3>) : reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 4, 8, 2, false, 3>);
            else
                fn = nb == ND6 ? reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, ND6, 8, 2, false, 1>) : reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 4, 8, 2, false, 1>);
            threads = 512;
        } else {
            fn = nb == ND6 ? reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, ND6>) : reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 4>);
            threads = 256;
        }
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        void* args[] = {&d};
        return hipLaunchKernel(fn, dim3(grid, d.groups), dim3(threads), args, lds, s);
    }
    static DeviceOnce attr_once;                            // one flag per template instantiation
    if (attr_once.need()) {
        const void* fns[4] = {reinterpret_cast<const void*>(igemm_kernel<T, BM, BN>),
                              reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 2>),
                              reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 3>),
                              reinterpret_cast<const void*>(igemm_kernel<T, BM, BN, void, 2>)};
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        if constexpr (BN >= 64) {
            const void* f8[2] = {reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 2, 8>),
                                 reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 3, 8>)};
            for (const void* f : f8) {
                hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
            }
        }
        attr_once.done();
    }
    if constexpr (BN >= 64) {
        if constexpr (std::is_same<T, bf16>::value && BN == 128) {
            if (prof_on() && variant >= 4 && variant <= 8) {
                const void* fp[5] = {reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 2, 8, 2, true>),
                                     reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 3, 8, 2, true>),
                                     reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 2, 8, 2, true, 1>),
                                     reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 3, 8, 2, true, 1>),
                                     reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 3, 8, 2, true, 2>)};
                for (const void* f : fp) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (variant == 4) hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 2, 8, 2, true>), dim3(grid, d.groups), dim3(512), lds, s, d);
                else if (variant == 5) hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 3, 8, 2, true>), dim3(grid, d.groups), dim3(512), lds, s, d);
                else if (variant == 6) hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 2, 8, 2, true, 1>), dim3(grid, d.groups), dim3(512), lds, s, d);
                else if (variant == 7) hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 3, 8, 2, true, 1>), dim3(grid, d.groups), dim3(512), lds, s, d);
                else hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 3, 8, 2, true, 2>), dim3(grid, d.groups), dim3(512), lds, s, d);
                return hipGetLastError();
            }
        }
        if (variant == 4) { hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 2, 8>), dim3(grid, d.groups), dim3(512), lds, s, d); return hipGetLastError(); }
        if (variant == 5) { hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 3, 8>), dim3(grid, d.groups), dim3(512), lds, s, d); return hipGetLastError(); }
        if constexpr (sizeof(T) == 2) {
            if (variant >= 6 && variant <= 8) {
                static bool ilv_attr = false;
                if (!ilv_attr) {
                    const void* fi[3] = {reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 2, 8, 2, false, 1>),
                                         reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 3, 8, 2, false, 1>),
                                         reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, 3, 8, 2, false, 2>)};
                    for (const void* f : fi) {
                        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        if (e != hipSuccess) return e;
                    }
                    ilv_attr = true;
                }
                if (variant == 6) hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 2, 8, 2, false, 1>), dim3(grid, d.groups), dim3(512), lds, s, d);
                else if (variant == 7) hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 3, 8, 2, false, 1>), dim3(grid, d.groups), dim3(512), lds, s, d);
                else hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 3, 8, 2, false, 2>), dim3(grid, d.groups), dim3(512), lds, s, d);
                return hipGetLastError();
            }
        } else {
            if (variant >= 6) variant -= 2;
        }
    } else {
        if (variant >= 6) variant -= 2;
        if (variant >= 4) variant -= 3;             // no 8-wave instantiation for 32-wide channel tiles
    }
    if (variant == 0) hipLaunchKernelGGL((igemm_kernel<T, BM, BN>), dim3(grid, d.groups), dim3(256), lds, s, d);
    else if (variant == 3) hipLaunchKernelGGL((igemm_kernel<T, BM, BN, void, 2>), dim3(grid, d.groups), dim3(256), lds, s, d);
    else if (variant == 1) hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 2>), dim3(grid, d.groups), dim3(256), lds, s, d);
    else hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, 3>), dim3(grid, d.groups), dim3(256), lds, s, d);
    return hipGetLastError();
}

// 256x128 tile, 8 waves as 4 (pixel) x 2 (channel): wave tile 64x64 like the 4-wave 128x128 kernel, but half the
// L2->LDS staging bytes per flop of a 128x128 tile (a 128x128x64 step needs 32 KB for 512 SIMD-cycles of MFMA -- exactly
// the 64 B/clk/CU a CU can pull; 256x128 needs 48 KB for 1024).  One workgroup per CU (96 / 144 KB of LDS).
template <typename T, int BM = 256, int BN = 128, int WMc = 4>
static hipError_t launch_big(IGemmDev d, int ring, int ilv, hipStream_t s) {
    d.tilesM = (d.M + BM - 1) / BM;
    d.tilesN = (d.N + BN - 1) / BN;
    const int tm8 = (d.tilesM + 7) / 8;
    int grid = tm8 * 8 * d.tilesN;
    static const char* fmap = dev_env("HCM_IGEMM_MAP");
    d.map = fmap ? atoi(fmap) : (((size_t)d.N * d.Kp * sizeof(T) > (2u << 20)) && d.tilesN >= 8 ? 1 : 0);
    if (d.map == 1) {
        if (d.tilesN < 8) d.map = 0;
        else grid = 8 * d.tilesM * ((d.tilesN + 7) / 8);
    }
    size_t lds = (size_t)ring * (BM + BN) * 128;
    const size_t lds_c = (size_t)BM * (BN + 4) * 4 + 1024;
    if (lds_c > lds) lds = lds_c;
    const bool prof = std::is_same<T, bf16>::value && prof_on();
    const void* fn;
#define HCM_BIG(R, P, I) reinterpret_cast<const void*>(igemm_dma_kernel<T, BM, BN, R, 8, WMc, P, I>)
    if (ilv == 2) ring = 3;
    if constexpr (std::is_same<T, bf16>::value) {
        if (prof) fn = ilv == 2 ? HCM_BIG(3, true, 2) : ring == 3 ? (ilv ? HCM_BIG(3, true, 1) : HCM_BIG(3, true, 0)) : (ilv ? HCM_BIG(2, true, 1) : HCM_BIG(2, true, 0));
        else fn = ilv == 2 ? HCM_BIG(3, false, 2) : ring == 3 ? (ilv ? HCM_BIG(3, false, 1) : HCM_BIG(3, false, 0)) : (ilv ? HCM_BIG(2, false, 1) : HCM_BIG(2, false, 0));
    } else {
        fn = ilv == 2 ? HCM_BIG(3, false, 2) : ring == 3 ? (ilv ? HCM_BIG(3, false, 1) : HCM_BIG(3, false, 0)) : (ilv ? HCM_BIG(2, false, 1) : HCM_BIG(2, false, 0));
    }
#undef HCM_BIG
    size_t lds2 = (size_t)ring * (BM + BN) * 128;
    if (lds2 > lds) lds = lds2;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    void* args[] = {&d};
    return hipLaunchKernel(fn, dim3(grid, d.groups), dim3(512), args, lds, s);
}

// first-layer (narrow-channel) launch: register-staged variant with the element-wise gather; N is 32 or 64
template <typename T, typename S>
static hipError_t launch_narrow(IGemmDev d, hipStream_t s) {
    constexpr int BM = 128;
    d.tilesM = (d.M + BM - 1) / BM;
    d.map = 0;
    const int tm8 = (d.tilesM + 7) / 8;
    if (d.N <= 32) {
        d.tilesN = 1;
        const size_t lds = 2 * (size_t)(BM + 32) * 128;
        hipLaunchKernelGGL((igemm_kernel<T, BM, 32, S>), dim3(tm8 * 8), dim3(256), lds, s, d);
    } else if (d.N <= 64) {
        d.tilesN = 1;
        const size_t lds = 2 * (size_t)(BM + 64) * 128;
        hipLaunchKernelGGL((igemm_kernel<T, BM, 64, S>), dim3(tm8 * 8), dim3(256), lds, s, d);
    } else {
        // 128-wide channel tiles: the (expensive) raw-frame gather is done once per pixel tile for up to 128 output channels
        // (the hi|lo pair stem has 2 x 64)
        d.tilesN = (d.N + 127) / 128;
        size_t lds = 2 * (size_t)(BM + 128) * 128;
        const size_t lds_c = (size_t)BM * (128 + 4) * 4;
        if (lds_c > lds) lds = lds_c;
        static DeviceOnce attr_once;
        if (attr_once.need()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_kernel<T, BM, 128, S>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_once.done();
        }
        hipLaunchKernelGGL((igemm_kernel<T, BM, 128, S>), dim3(tm8 * 8 * d.tilesN), dim3(256), lds, s, d);
    }
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_narrow_src(const IGemmDev& d, int src_dt, int dt, hipStream_t s) {
    if (src_dt == DT_F32) return launch_narrow<T, float>(d, s);
    if (src_dt == DT_U8) return launch_narrow<T, uint8_t>(d, s);
    if (src_dt == dt) return launch_narrow<T, T>(d, s);
    return hipErrorInvalidValue;
}

static const int kTiles[6][2] = {{128, 128}, {128, 64}, {64, 64}, {64, 32}, {128, 32}, {64, 128}};

template <typename T>
static hipError_t launch_choice(const IGemmDev& d, int choice, hipStream_t s) {
    const int variant = choice / 6;
    switch (choice % 6) {
        case 0: return launch_cfg<T, 128, 128>(d, variant, s);
        case 1: return launch_cfg<T, 128, 64>(d, variant, s);
        case 2: return launch_cfg<T, 64, 64>(d, variant, s);
        case 3: return launch_cfg<T, 64, 32>(d, variant, s);
        case 4: return launch_cfg<T, 128, 32>(d, variant, s);
        default: return launch_cfg<T, 64, 128>(d, variant, s);
    }
}
static hipError_t launch_dt(const IGemmDev& d, int dt, int choice, hipStream_t s) {
    if (d.cs_part && choice >= 100) choice = 24;           // the big-tile launchers do not reserve the column-sum scratch
    if (choice >= 110) {                                // 110..115: 64 pixels x 256 channels (whole 512-byte output rows per workgroup)
        const int c2 = choice - 110;
        const int ring = 2 + (c2 & 1);
        const int ilv = c2 >= 4 ? 2 : c2 >= 2 ? 1 : 0;
        if (dt == DT_BF16) return launch_big<bf16, 64, 256, 2>(d, ring, ilv, s);
        if (dt == DT_F16) return launch_big<f16, 64, 256, 2>(d, ring, ilv, s);
        return hipErrorInvalidValue;
    }
    if (choice >= 100) {                                // 100 / 101: 256x128 tile with a 2- / 3-deep LDS ring; 102 / 103: interleaved DMA
        const int ring = 2 + (choice & 1);
        const int ilv = choice >= 104 ? 2 : choice >= 102 ? 1 : 0;       // 105: rotated loop
        if (dt == DT_BF16) return launch_big<bf16>(d, ring, ilv, s);
        if (dt == DT_F16) return launch_big<f16>(d, ring, ilv, s);
        return hipErrorInvalidValue;
    }
    if (dt == DT_BF16) return launch_choice<bf16>(d, choice, s);
    if (dt == DT_F16) return launch_choice<f16>(d, choice, s);
    if (dt == DT_F32) return launch_choice<float>(d, choice, s);
    return hipErrorInvalidValue;
}

// Default (tile, staging variant) per shape, distilled from the per-shape sweep in profiles/igemm_sweep_r1.md:
// LDS-DMA with 2 buffers nearly everywhere (2 workgroups per CU); the 3-deep ring only for long K loops on small
// tiles (few workgroups, latency bound); 64-row pixel tiles whenever 128x128 would leave CUs idle or K is so short
// that the kernel is a streaming copy with a matmul attached.
static inline int sizeof_dt(int dt) { return dt == DT_F32 ? 4 : 2; }
static int heuristic_choice(const IGemmDev& d, int dt) {
    auto cdiv = [](long a, long b) { return (a + b - 1) / b; };
    const long b128 = cdiv(d.M, 128) * cdiv(d.N, 128) * d.groups;
    const long b64128 = cdiv(d.M, 64) * cdiv(d.N, 128) * d.groups;
    const long b64 = cdiv(d.M, 64) * cdiv(d.N, 64) * d.groups;
    // tile index into kTiles: 0 128x128, 1 128x64, 2 64x64, 3 64x32, 4 128x32, 5 64x128
    // variant: 1 dma2, 2 dma3 (4 waves); 4 dma2, 5 dma3 (8 waves)
    int tile, variant;
    // write-dominated expansions (small K, >= 256 output channels, big M): 64 x 256 tiles write whole 512-byte pixel rows.
    // Stand-alone (no residual, one group) layer1's 64 -> 256 goes 55 -> 45 us; in the step (hi|lo groups, residual) it
    // measured 1.5 % SLOWER end to end, so it stays opt-in.
    static const int row_tiles = dev_env("HCM_IGEMM_ROW256") ? atoi(dev_env("HCM_IGEMM_ROW256")) : 0;
    if (row_tiles && dt != DT_F32 && d.N >= 256 && d.N % 256 == 0 && d.K <= 256 && (long)d.M * d.groups >= 32768) return 110;
    const bool longk = d.K >= 768;
    if (d.N <= 32) { tile = 3; variant = d.K >= 2048 ? 2 : 1; }
    else if (d.N <= 64) {
        tile = d.M >= 65536 ? 1 : (b64 >= 256 ? 2 : 3);
        variant = tile == 3 ? (d.K >= 2048 ? 2 : 1) : 4;
    }
    else if (d.M <= 64) { tile = 3; variant = d.K >= 512 ? 2 : 1; }     // skinny, latency-bound: deeper ring
    else if (b128 >= 512) { tile = 0; variant = 4; }     // (the rotated 3-ring variant 8 wins stand-alone for K >= 2048 but
                                                         //  loses end to end: 1 workgroup per CU blocks the other streams)
    // 192-511 big tiles and a long K (BERT's 3072 -> 768, layer4's 3x3): one 128x128 workgroup per CU with the interleaved 3-deep ring.
    // Stand-alone the 64x128 split is as fast; inside the step the whole-step rate is 1-2 % higher with the big tile (measured per shape
    // with HCM_IGEMM_SHAPE_FORCE: 12 334 -> 12 603 env-steps/s for the FFN shape alone) -- the second workgroup slot of every CU stays free
    // for the other chains' kernels
    else if (b128 >= 192 && d.K >= 2048 && sizeof_dt(dt) == 2) { tile = 0; variant = 7; }
    else if (b64128 >= 128) { tile = 5; variant = longk ? (sizeof_dt(dt) == 2 ? 7 : 5) : 4; }   // long K: interleaved DMA issue
    else { tile = b64 >= 256 ? 2 : 3; variant = d.K >= 2048 ? 2 : 1; }
    // launches whose whole grid fits the chip one workgroup per CU, with a K loop worth pipelining: the deep ring (LDS is free there)
    static const int deep = dev_env("HCM_IGEMM_DEEP") ? atoi(dev_env("HCM_IGEMM_DEEP")) : 9;     // (development build: 0 off, 9 six tiles, 10 four)
    if (deep && variant != 0 && variant != 3) {
        const long blocks = cdiv(d.M, kTiles[tile][0]) * cdiv(d.N, kTiles[tile][1]) * d.groups;
        if (blocks <= 256 && d.K >= 256 && !d.hpool) return deep * 6 + tile;
    }
    return variant * 6 + tile;
}

// ---- per-shape autotuner (run once per handle at hcm_finalize on the real layer shapes) ----
static bool g_tuning = false;
static std::unordered_map<std::string, int> g_choice;
static std::mutex g_choice_mu;

void igemm_set_tuning(bool on) { g_tuning = on; }
size_t igemm_tuned_shapes() { std::lock_guard<std::mutex> l(g_choice_mu); return g_choice.size(); }

static std::string shape_key(const IGemmDev& d, int dt) {
    char buf[160];
    snprintf(buf, sizeof buf, "%d|%d,%d,%d|%d,%d,%d,%d|%d,%d,%d,%d,%d|%d,%d", dt, d.M, d.N, d.K, d.H, d.W, d.Cin, d.xC, d.KH, d.KW, d.stride,
             d.stride_w, d.pad, d.res != nullptr, d.out_f32);
    return buf;
}

static bool candidate_ok(const IGemmDev& d, int choice) {
    const int bm = kTiles[choice % 6][0], bn = kTiles[choice % 6][1];
    if (bn >= 64 && d.N <= bn / 2) return false;            // mostly-empty channel tile
    if (bm == 128 && d.M <= 64) return false;
    return true;
}

static hipError_t tune_shape(const IGemmDev& d, int dt, hipStream_t s, int* best_out) {
    hipEvent_t e0, e1;
    hipError_t rc = hipEventCreate(&e0);
    if (rc != hipSuccess) return rc;
    rc = hipEventCreate(&e1);
    if (rc != hipSuccess) return rc;
    float best = 1e30f;
    int best_c = heuristic_choice(d, dt);
    for (int c = 0; c < 36; ++c) {
        if (!candidate_ok(d, c)) continue;
        if ((rc = launch_dt(d, dt, c, s)) != hipSuccess) break;         // warm-up (also sets the LDS attribute)
        float tmin = 1e30f;
        for (int rep = 0; rep < 3 && rc == hipSuccess; ++rep) {
            (void)hipEventRecord(e0, s);
            rc = launch_dt(d, dt, c, s);
            (void)hipEventRecord(e1, s);
            if (rc != hipSuccess) break;
            if ((rc = hipEventSynchronize(e1)) != hipSuccess) break;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < tmin) tmin = ms;
        }
        if (rc != hipSuccess) break;
        if (tmin < best) { best = tmin; best_c = c; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *best_out = best_c;
    return rc;
}

#ifdef HCM_DEV_KNOBS
// make DEV=1 builds: HCM_IGEMM_TIME=1 brackets every launch with events (synchronising: stand-alone durations) and prints a per-shape
// table at exit -- how profiles/r2e_igemm_shapes.md was made
static hipError_t launch_igemm_impl(const IGemm& g, int dt, hipStream_t s);
namespace {
struct ShapeTime { double us = 0; long n = 0; double flop = 0; double bytes = 0; };
std::unordered_map<std::string, ShapeTime> g_shape_time;
struct ShapeTimePrinter {
    ~ShapeTimePrinter() {
        if (g_shape_time.empty()) return;
        std::vector<std::pair<std::string, ShapeTime>> v(g_shape_time.begin(), g_shape_time.end());
        std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.second.us > b.second.us; });
        double tot = 0;
        for (auto& kv : v) tot += kv.second.us;
        // algorithmic bytes: input map + weights + output (+ identity) once each; fractions against 2.5 PFLOP/s dense 16-bit (f32: 1/16) and 8 TB/s
        fprintf(stderr, "| shape | launches | total us | avg us | TFLOP/s | of MFMA peak | alg MB / launch | TB/s | of HBM peak |\n|---|---|---|---|---|---|---|---|---|\n");
        for (auto& kv : v) {
            const double tf = kv.second.flop / kv.second.us / 1e6, tb = kv.second.bytes / kv.second.us / 1e6;
            const bool f32 = kv.first.rfind("dt0", 0) == 0;
            fprintf(stderr, "| %s | %ld | %.0f | %.1f | %.0f | %.3f | %.1f | %.2f | %.3f |\n", kv.first.c_str(), kv.second.n, kv.second.us, kv.second.us / kv.second.n,
                    tf, tf / (f32 ? 156.0 : 2500.0), kv.second.bytes / kv.second.n / 1e6, tb, tb / 8.0);
        }
        fprintf(stderr, "total %.0f us\n", tot);
    }
} g_shape_time_printer;
}
hipError_t launch_igemm(const IGemm& g, int dt, hipStream_t s) {
    static const bool timing = dev_env("HCM_IGEMM_TIME") != nullptr;
    if (!timing) return launch_igemm_impl(g, dt, s);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipDeviceSynchronize();
    // the launch is issued twice and the SECOND one timed: while the first runs, the host has queued e0 and the second launch behind it, so
    // the interval holds the kernel and not the host's submission latency (5-15 us on an idle stream).  Timing aid only: a launch that
    // accumulates in place runs its accumulation twice.
    (void)launch_igemm_impl(g, dt, s);
    (void)hipEventRecord(e0, s);
    hipError_t rc = launch_igemm_impl(g, dt, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    char key[160];
    snprintf(key, sizeof key, "dt%d M=%d N=%d K=%d g=%d %dx%d s%d%s%s%s", dt, g.M, g.N, g.K, g.groups, g.KH, g.KW, g.stride, g.res ? " res" : "",
             g.gn_gamma ? " gn" : "", g.cs_part ? " cs" : "");
    ShapeTime& t = g_shape_time[key];
    const double gr = g.groups > 1 ? g.groups : 1, esz = dt == DT_F32 ? 4 : 2;
    t.us += ms * 1e3; t.n += 1; t.flop += 2.0 * g.M * g.N * g.K * gr;
    const double in_elems = (g.KH * g.KW > 1 || g.H > 1) ? (double)g.B * g.H * g.W * g.Cin : (double)g.M * g.K;
    t.bytes += gr * ((in_elems + (double)g.N * g.K) * esz + (double)g.M * g.N * (g.out_f32 ? 4 : esz) + (g.res ? (double)g.M * g.N * esz : 0));
    return rc;
}
static hipError_t launch_igemm_impl(const IGemm& g, int dt, hipStream_t s) {
#else
hipError_t launch_igemm(const IGemm& g, int dt, hipStream_t s) {
#endif
    if ((g.impl & 15) == 2) return launch_gemm256(g, dt, s);
    if ((g.impl & 15) == 3) return launch_skinny(g, dt, s);
    if (g.impl == 0) {
        static const bool no256 = dev_env("HCM_NO_GEMM256") != nullptr;
        if (!no256 && gemm256_applicable(g, dt)) return launch_gemm256(g, dt, s);
        // few rows (round 5): a wave per 16 x 16 output tile, operands straight into registers (skinny.hip; bit-identical).  HCM_NO_SKINNY=1
        // (development build): the implicit-GEMM tiles, for the A/B and the toggle test
        static const bool no_skinny = dev_env("HCM_NO_SKINNY") != nullptr;
        if (!no_skinny && g.force_choice < 0 && skinny_applicable(g, dt)) return launch_skinny(g, dt, s);
    }
    IGemmDev d;
    d.x = (const char*)g.x; d.w = (const char*)g.w; d.bias = g.bias; d.res = (const char*)g.res; d.y = (char*)g.y;
    d.B = g.B; d.H = g.H; d.W = g.W; d.Cin = g.Cin; d.xC = g.xC ? g.xC : g.Cin;
    d.Ho = g.Ho; d.Wo = g.Wo; d.KH = g.KH; d.KW = g.KW; d.stride = g.stride; d.pad = g.pad;
    d.stride_w = g.stride_w > 0 ? g.stride_w : g.stride;
    d.M = g.M; d.N = g.N; d.K = g.K; d.Kp = g.Kp ? g.Kp : g.K;
    d.ldy = g.ldy ? g.ldy : g.N; d.ldr = g.ldr ? g.ldr : g.N; d.act = g.act; d.out_f32 = g.out_f32; d.res_f32 = g.res_f32;
    static const bool image_epi = dev_env("HCM_IGEMM_IMAGE") != nullptr;
    d.image_epi = image_epi ? 1 : 0;
    const int CH = dt_chunk(dt);
    d.cin_shift = 0;
    d.kw_rcp = (65536 + g.KW - 1) / g.KW;
    if (g.KH * g.KW > 1 && g.x_src_dt < 0) {
        if (g.Cin & (g.Cin - 1)) return hipErrorInvalidValue;     // spatial kernels need power-of-two Cin
        while ((1 << d.cin_shift) < g.Cin) ++d.cin_shift;
    }
    if (d.M <= 0 || d.N <= 0 || d.K <= 0) return hipErrorInvalidValue;
    d.x_scale = g.x_scale;
    d.rowrun = 0;
    d.hpool = 0;
    d.cs_part = g.cs_part; d.cs_cg = g.cs_cg; d.cs_hw = g.cs_hw; d.cs_G = g.cs_G;
    d.groups = g.groups > 1 ? g.groups : 1;
    d.g_x = g.g_x; d.g_w = g.g_w; d.g_b = g.g_b; d.g_y = g.g_y;
    d.gn_gamma = g.gn_gamma; d.gn_beta = g.gn_beta; d.gn_cg = g.gn_gamma ? g.gn_cg : 0; d.gn_hw = g.gn_hw; d.gn_eps = g.gn_eps;
    if (g.x_src_dt >= 0) {
        // narrow-channel first layer: element-wise gather from the raw frame
        d.rowrun = g.x_rowrun;
        if ((d.Kp % CH) || (d.N % 4) || (d.ldy % 4) || d.res) return hipErrorInvalidValue;
        if (d.rowrun ? (d.K != d.KH * 24 || d.Cin != 3 || d.KW * 3 > 24 || g.x_src_dt != DT_F32) : (d.KH * d.KW * d.Cin != d.K))
            return hipErrorInvalidValue;
        d.w_bytes = 0;
        {
            const size_t xb = (size_t)d.B * d.H * d.W * d.xC * (g.x_src_dt == DT_F32 ? 4 : g.x_src_dt == DT_U8 ? 1 : dt_size(dt));
            if (xb >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
            d.x_bytes = (unsigned)xb;
        }
        if (dt == DT_BF16) return launch_narrow_src<bf16>(d, g.x_src_dt, dt, s);
        if (dt == DT_F16) return launch_narrow_src<f16>(d, g.x_src_dt, dt, s);
        if (dt == DT_F32) return launch_narrow_src<float>(d, g.x_src_dt, dt, s);
        return hipErrorInvalidValue;
    }
    {
        const size_t esz = dt_size(dt);
        const size_t xb = (((size_t)d.B * d.H * d.W - 1) * d.xC + d.Cin) * esz;
        const size_t wb = (size_t)d.N * d.Kp * esz;
        if (xb >= 0xFFFFFFF0ull || wb >= 0xFFFFFFF0ull) return hipErrorInvalidValue;   // 32-bit buffer offsets
        d.x_bytes = (unsigned)xb;
        d.w_bytes = (unsigned)wb;
    }
    // pixel stride: a multiple of the 16-byte chunk, or (LDS-DMA variants only: dword-aligned source addresses suffice) of 4 bytes
    const bool narrow_stride = (d.xC % CH) != 0;
    if ((g.Cin % CH) || (narrow_stride && (d.xC * (int)dt_size(dt)) % 4) || (d.K % CH) || (d.Kp % CH) || (d.N % 4) || (d.ldy % 4) || (d.ldr % 4))
        return hipErrorInvalidValue;
    if (dt != DT_BF16 && dt != DT_F16 && dt != DT_F32) return hipErrorInvalidValue;
    if (d.cs_part && (d.gn_cg || d.bias || g.x_src_dt >= 0 || d.cs_cg < 1 || 32 % d.cs_cg || d.N % d.cs_cg || d.cs_hw % 64 || d.M % d.cs_hw))
        return hipErrorInvalidValue;
    d.rln_stats = g.rln_stats; d.rln_gamma = g.rln_gamma; d.rln_beta = g.rln_beta;
    if (g.rln_stats && (!g.res || (g.res_f32 && !g.out_f32) || dt == DT_F32 || !g.rln_gamma || !g.rln_beta || (d.N % 8) || (d.ldr % 8) || g.groups > 1 || d.gn_cg || d.cs_part))
        return hipErrorInvalidValue;              // the LayerNorm-on-the-fly residual exists in the 16-byte residual paths of the 16-bit epilogues only
    if (g.ln_s) return hipErrorInvalidValue;      // the folded-LayerNorm consumer is gemm256f_kernel only (launch_gemm256)
    d.ln_part = g.ln_part_out; d.ln_part_P = g.ln_part_P;
    if (g.ln_part_out) {
        // partial row statistics come out of the REGISTER epilogue of the 8-wave 128-channel tiles (a wave column = 32 channels = one slot)
        const int fc = g.force_choice;
        const bool tile_ok = fc >= 0 && fc < 100 && (fc % 6 == 0 || fc % 6 == 5) && (fc / 6 == 4 || fc / 6 == 5 || fc / 6 == 6 || fc / 6 == 7 || fc / 6 == 9 || fc / 6 == 10);
        if (!tile_ok || dt == DT_F32 || g.out_f32 || (d.N % 32) || g.ln_part_P != d.N / 32 || d.gn_cg || d.cs_part || g.groups > 1 || (d.ldy % 8) || d.image_epi ||
            (g.res && (d.ldr % 8)))
            return hipErrorInvalidValue;
    }
    d.gi_stats = nullptr;
    if (g.gi_stats) {
        if (!igemm_gnin_ok(g, dt)) return hipErrorInvalidValue;
        if (d.gn_cg && (d.gn_hw <= 0 || 64 % d.gn_hw || d.M % d.gn_hw || d.gn_cg % 8 || 128 % d.gn_cg || d.N % d.gn_cg || d.bias || !d.gn_beta))
            return hipErrorInvalidValue;
        d.gi_stats = g.gi_stats; d.gi_gamma = g.gi_gamma; d.gi_beta = g.gi_beta; d.gi_ps = g.gi_ps; d.gi_cg = g.gi_cg; d.gi_G = g.gi_G;
        d.gi_hw = g.gi_hw; d.gi_relu = g.gi_relu; d.gi_eps = g.gi_eps; d.gi_res = (const char*)g.gi_res; d.gi_out = (char*)g.gi_out;
        return dt == DT_BF16 ? launch_gnin<bf16>(d, s) : launch_gnin<f16>(d, s);
    }
    if (d.gn_cg) {
        // fused GroupNorm: 64-row tiles of whole samples, 128 channels of whole groups (igemm_epilogue)
        if (g.x_src_dt >= 0 || d.gn_hw <= 0 || 64 % d.gn_hw || d.M % d.gn_hw || d.gn_cg % 8 || 128 % d.gn_cg || d.N % d.gn_cg || d.bias ||
            d.out_f32 || !d.gn_beta)
            return hipErrorInvalidValue;
        // (16-bit, the whole grid one workgroup per CU: the deep ring -- see igemm_dma_kernel, NBUF > 3)
        static const int deep_gn = dev_env("HCM_IGEMM_DEEP") ? atoi(dev_env("HCM_IGEMM_DEEP")) : 9;
        const long blocks_gn = (long)((d.M + 63) / 64) * ((d.N + 127) / 128) * d.groups;
        if (deep_gn && dt != DT_F32 && blocks_gn <= 256 && d.K >= 256) return launch_dt(d, dt, deep_gn * 6 + 5, s);
        return launch_dt(d, dt, ((d.K >= 768 && dt != DT_F32) ? 7 : 4) * 6 + 5, s);
    }
    if (g.hpool) {
        // stem conv + horizontal max-pool: tiles of 128 whole-row pixels (8-wave, 2-deep ring), N in whole 64 / 128-channel tiles
        if ((dt != DT_BF16 && dt != DT_F16) || d.res || d.out_f32 || d.Wo < 2 || (d.Wo & (d.Wo - 1)) || d.Wo > 128 || (d.M % d.Wo) || (d.N % 64) ||
            (d.ldy % 8) || (d.act != ACT_RELU && d.act != ACT_NONE))
            return hipErrorInvalidValue;
        d.hpool = 1;
        const bool wide = d.N % 128 == 0;
        const int BN = wide ? 128 : 64;
        d.tilesM = (d.M + 127) / 128;
        d.tilesN = d.N / BN;
        d.map = 0;
        const int grid = ((d.tilesM + 7) / 8) * 8 * d.tilesN;
        size_t lds = 2 * (size_t)(128 + BN) * 128;
        const size_t lds_c = (size_t)128 * (BN + 4) * 4;
        if (lds_c > lds) lds = lds_c;
        const void* fn;
        if (dt == DT_BF16) fn = wide ? reinterpret_cast<const void*>(igemm_dma_kernel<bf16, 128, 128, 2, 8, 2, false, 0, true>)
                                     : reinterpret_cast<const void*>(igemm_dma_kernel<bf16, 128, 64, 2, 8, 2, false, 0, true>);
        else fn = wide ? reinterpret_cast<const void*>(igemm_dma_kernel<f16, 128, 128, 2, 8, 2, false, 0, true>)
                       : reinterpret_cast<const void*>(igemm_dma_kernel<f16, 128, 64, 2, 8, 2, false, 0, true>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        void* args[] = {&d};
        return hipLaunchKernel(fn, dim3(grid, d.groups), dim3(512), args, lds, s);
    }
    // tuning aid: HCM_IGEMM_SHAPE_FORCE="M,N,K:choice;M,N,K:choice" forces the tile / staging choice of single GEMM shapes inside a whole step
    static const std::unordered_map<std::string, int> shape_force = [] {
        std::unordered_map<std::string, int> m;
        const char* e = dev_env("HCM_IGEMM_SHAPE_FORCE");
        if (e) {
            std::string v(e);
            size_t pos = 0;
            while (pos < v.size()) {
                const size_t end = v.find(';', pos);
                const std::string item = v.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
                const size_t c = item.find(':');
                if (c != std::string::npos) m[item.substr(0, c)] = atoi(item.c_str() + c + 1);
                if (end == std::string::npos) break;
                pos = end + 1;
            }
        }
        return m;
    }();
    if (g.force_choice >= 0) return launch_dt(d, dt, g.force_choice, s);
    if (!shape_force.empty() && !narrow_stride) {
        auto it = shape_force.find(std::to_string(d.M) + "," + std::to_string(d.N) + "," + std::to_string(d.K));
        if (it != shape_force.end()) return launch_dt(d, dt, it->second, s);
    }
    static const char* force = dev_env("HCM_IGEMM_FORCE");      // debugging: variant*6 + tile
    if (force && !(narrow_stride && (atoi(force) / 6 == 0 || atoi(force) / 6 == 3))) return launch_dt(d, dt, atoi(force), s);
    const std::string key = shape_key(d, dt);
    int choice = -1;
    {
        std::lock_guard<std::mutex> l(g_choice_mu);
        auto it = g_choice.find(key);
        if (it != g_choice.end()) choice = it->second;
    }
    if (choice < 0) {
        if (g_tuning) {
            hipError_t rc = tune_shape(d, dt, s, &choice);
            if (rc != hipSuccess) return rc;
            std::lock_guard<std::mutex> l(g_choice_mu);
            g_choice[key] = choice;
        } else {
            choice = heuristic_choice(d, dt);
        }
    }
    static const bool log_shapes = dev_env("HCM_IGEMM_LOG") != nullptr;       // tuning aid: every new shape and its choice, once
    if (log_shapes) {
        static std::unordered_map<std::string, int> seen;
        std::lock_guard<std::mutex> l(g_choice_mu);
        if (seen.emplace(key, choice).second)
            fprintf(stderr, "[igemm] dt=%d M=%d N=%d K=%d groups=%d KHxKW=%dx%d stride=%d res=%d -> choice %d\n", dt, d.M, d.N, d.K, d.groups, d.KH, d.KW,
                    d.stride, d.res != nullptr, choice);
    }
    return launch_dt(d, dt, choice, s);
}


constexpr int kHaloRing = 5, kHaloRingD = 5;       // weight-tile ring depth of the halo phase A (128-pixel tiles / the 64-pixel tile of the folded down-sample block)

hipError_t launch_bneck23(const Bneck23& b, int dt, hipStream_t s) {
    if (dt != DT_BF16 && dt != DT_F16) return hipErrorInvalidValue;
    if ((b.C1 != 64 && b.C1 != 128 && b.C1 != 256) || (!b.res && !b.xd) || !b.b2 || !b.b3 || b.stride < 1) return hipErrorInvalidValue;
    if (b.C1 == 256 && (!b.w1 || b.CN != 256 || b.xd)) return hipErrorInvalidValue;        // 256 mid channels: only the tail + next-reduction form is built
    if (b.xd && (b.C1 != 64 || b.KD != 1 || !b.w1 || b.CN != 64 || (b.xdC % 8))) return hipErrorInvalidValue;   // the one folded-down-sample shape built
    const int C3 = 4 * b.C1;
    const int ldy = b.ldy ? b.ldy : C3, ldr = b.ldr ? b.ldr : C3, xC = b.xC ? b.xC : b.C1;
    if ((ldy % 8) || (ldr % 8) || (xC % 8)) return hipErrorInvalidValue;
    BneckDev q;
    IGemmDev& d = q.a;
    memset((void*)&q, 0, sizeof(q));
    d.x = (const char*)b.x; d.w = (const char*)b.w2; d.bias = b.b2;
    d.B = b.B; d.H = b.H; d.W = b.W; d.Cin = b.C1; d.xC = xC;
    d.Ho = (b.H + 2 - 3) / b.stride + 1; d.Wo = (b.W + 2 - 3) / b.stride + 1;
    d.KH = 3; d.KW = 3; d.stride = b.stride; d.stride_w = b.stride; d.pad = 1;
    d.M = b.B * d.Ho * d.Wo; d.N = b.C1; d.K = 9 * b.C1; d.Kp = 9 * b.C1;
    d.cin_shift = b.C1 == 64 ? 6 : b.C1 == 128 ? 7 : 8;
    d.kw_rcp = (65536 + 3 - 1) / 3;
    d.groups = b.groups > 1 ? b.groups : 1;
    d.g_x = b.g_x; d.g_w = b.g_w2; d.g_b = b.g_b2;
    {
        const size_t xb = (((size_t)d.B * d.H * d.W - 1) * d.xC + d.Cin) * 2, wb = (size_t)d.N * d.Kp * 2, w3b = (size_t)C3 * b.C1 * 2;
        if (xb >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
        d.x_bytes = (unsigned)xb; d.w_bytes = (unsigned)wb; q.w3_bytes = (unsigned)w3b;
    }
    q.w3 = (const char*)b.w3; q.b3 = b.b3; q.res = (const char*)b.res; q.y = (char*)b.y;
    q.C3 = C3; q.Kp3 = b.C1 + (b.xd ? b.KD * 64 : 0); q.ldy3 = ldy; q.ldr3 = ldr;
    q.w3_bytes = (unsigned)((size_t)C3 * q.Kp3 * 2);
    q.g_w3 = b.g_w3; q.g_b3 = b.g_b3; q.g_y3 = b.g_y;
    const int BM = b.C1 == 128 ? 64 : 128;
    const int KT1 = b.C1 / 64;
    if (b.w1) {
        if (!b.b1 || !b.o1 || (b.CN != 64 && b.CN != 128 && b.CN != 256) || (b.C1 == 128 && b.CN != 128) || ((b.C1 == 256) != (b.CN == 256))) return hipErrorInvalidValue;
        const int ldo = b.ldo ? b.ldo : b.CN;
        if (ldo % 8) return hipErrorInvalidValue;
        Bneck231Dev qq;
        qq.t = q;
        qq.w1 = (const char*)b.w1; qq.b1 = b.b1; qq.o1 = (char*)b.o1; qq.ldo = ldo;
        qq.w1_bytes = (unsigned)((size_t)b.CN * C3 * 2);
        qq.g_w1 = b.g_w1; qq.g_b1 = b.g_b1; qq.g_o1 = b.g_o1;
        qq.xd = nullptr; qq.xdC = 0; qq.xd_bytes = 0; qq.g_xd = 0;
        qq.xcd_tiles = 0;
        if (b.xd) {
            const size_t xdb = (((size_t)d.B * d.H * d.W - 1) * b.xdC + b.KD * 64) * 2;
            if (xdb >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
            qq.xd = (const char*)b.xd; qq.xdC = b.xdC; qq.xd_bytes = (unsigned)xdb; qq.g_xd = b.g_xd;
            const int BMd = 64;
            // register-epilogue form (bneck231r_kernel, the default) or the LDS-image form (HCM_BNECK_IMAGE=1: A/B and the toggle test); bit-identical
            static const bool image_d = dev_env("HCM_BNECK_IMAGE") != nullptr;
            size_t ldsd = image_d ? (size_t)2 * BMd * 128 + (size_t)2 * 64 * 128 + (64 * 132 * 4 / 2 + 1024) + (size_t)BMd * 128 + (size_t)b.CN * 128
                                  : (size_t)2 * BMd * 128 + (size_t)2 * 64 * 128 + (size_t)BMd * 128 + (size_t)2 * b.CN * 128 + (size_t)(4 * b.C1 + b.CN) * 4;
            const size_t ringd = 3 * (size_t)(BMd + b.C1) * 128;
            if (ringd > ldsd) ldsd = ringd;
            const void* fd = image_d ? (dt == DT_BF16 ? reinterpret_cast<const void*>(bneck231_kernel<bf16, 64, 64, 64, 1>) : reinterpret_cast<const void*>(bneck231_kernel<f16, 64, 64, 64, 1>))
                                     : (dt == DT_BF16 ? reinterpret_cast<const void*>(bneck231r_kernel<bf16, 64, 64, 64, 1>) : reinterpret_cast<const void*>(bneck231r_kernel<f16, 64, 64, 64, 1>));
            // halo phase A (see bneck231r_kernel): a tile of whole rows of one image
            static const bool no_halo_d = dev_env("HCM_NO_BNECK_HALO") != nullptr;
            // 128-pixel tiles (one buffer for the reduction's weight slice, halo phase A) where the map allows: per-tile fixed costs halve
            static const bool no_big_d = dev_env("HCM_NO_BNECK_DS128") != nullptr;
            if (!image_d && !no_halo_d && !no_big_d && b.stride == 1 && d.W % 16 == 0 && 128 % d.W == 0 && (d.Ho * d.Wo) % 128 == 0) {
                const size_t halo = (size_t)((((128 / d.W + 2) * (d.W + 2)) + 7) & ~7) * 128 + (size_t)kHaloRing * 64 * 128;
                size_t lds = (size_t)2 * 128 * 128 + (size_t)2 * 64 * 128 + (size_t)128 * 128 + (size_t)b.CN * 128 + (size_t)(4 * b.C1 + b.CN) * 4;
                if (halo > lds) lds = halo;
                if (lds <= 80 * 1024) {
                    const void* fb = dt == DT_BF16 ? reinterpret_cast<const void*>(bneck231r_kernel<bf16, 128, 64, 64, 1, false, kHaloRing, 1>)
                                                   : reinterpret_cast<const void*>(bneck231r_kernel<f16, 128, 64, 64, 1, false, kHaloRing, 1>);
                    hipError_t eb = hipFuncSetAttribute(fb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    if (eb != hipSuccess) return eb;
                    void* ab[] = {&qq};
                    return hipLaunchKernel(fb, dim3((d.M + 127) / 128, d.groups), dim3(512), ab, lds, s);
                }
            }
            if (!image_d && !no_halo_d && b.stride == 1 && d.W % 16 == 0 && BMd % d.W == 0 && (d.Ho * d.Wo) % BMd == 0) {
                const size_t halo = (size_t)((((BMd / d.W + 2) * (d.W + 2)) + 7) & ~7) * 128 + (size_t)kHaloRingD * 64 * 128;
                if (halo <= 80 * 1024) {
                    ldsd = (size_t)2 * BMd * 128 + (size_t)2 * 64 * 128 + (size_t)BMd * 128 + (size_t)2 * b.CN * 128 + (size_t)(4 * b.C1 + b.CN) * 4;
                    if (halo > ldsd) ldsd = halo;
                    fd = dt == DT_BF16 ? reinterpret_cast<const void*>(bneck231r_kernel<bf16, 64, 64, 64, 1, false, kHaloRingD>)
                                       : reinterpret_cast<const void*>(bneck231r_kernel<f16, 64, 64, 64, 1, false, kHaloRingD>);
                }
            }
            hipError_t ed = hipFuncSetAttribute(fd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (ed != hipSuccess) return ed;
            void* ad[] = {&qq};
            return hipLaunchKernel(fd, dim3((d.M + BMd - 1) / BMd, d.groups), dim3(512), ad, ldsd, s);
        }
        if (b.C1 == 256) {
            // RGB layer3 (16 x 16 maps at 256-pixel frames): 128-pixel tiles, one workgroup per CU (149 KB: the parked 128 x 256 tile 64 KB, a 32 KB
            // expansion-weight slice, the 16 KB slice block, ONE 32 KB buffer for the reduction's weight slice, 5 KB of biases; phase A's 3-deep ring
            // of 48 KB tap tiles lives in the same bytes before that), identity rows streamed two slices deep (ID_STREAM)
            const size_t lb = (size_t)KT1 * BM * 128 + (size_t)KT1 * 64 * 128 + (size_t)BM * 128 + (size_t)b.CN * 128 + (size_t)(4 * b.C1 + b.CN) * 4;
            const size_t ra = 3 * (size_t)(BM + b.C1) * 128;
            const size_t lds256 = lb > ra ? lb : ra;
            if (lds256 > 160 * 1024) return hipErrorInvalidValue;
            const void* f2 = dt == DT_BF16 ? reinterpret_cast<const void*>(bneck231r_kernel<bf16, 128, 256, 256, 0, false, 0, 1>)
                                           : reinterpret_cast<const void*>(bneck231r_kernel<f16, 128, 256, 256, 0, false, 0, 1>);
            hipError_t e2 = hipFuncSetAttribute(f2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e2 != hipSuccess) return e2;
            void* a2[] = {&qq};
            static const bool no_xcd = dev_env("HCM_NO_BNECK_XCD") != nullptr;
            if (d.groups == 2 && !no_xcd) {
                const int tiles = (d.M + BM - 1) / BM;
                qq.xcd_tiles = tiles;
                return hipLaunchKernel(f2, dim3(((tiles + 3) / 4) * 8, 1), dim3(512), a2, lds256, s);
            }
            return hipLaunchKernel(f2, dim3((d.M + BM - 1) / BM, d.groups), dim3(512), a2, lds256, s);
        }
        static const bool image = dev_env("HCM_BNECK_IMAGE") != nullptr;
        size_t lds1 = image ? (size_t)KT1 * BM * 128 + (size_t)KT1 * 64 * 128 + (64 * 132 * 4 / 2 + 1024) + (size_t)BM * 128 + (size_t)b.CN * 128
                            : (size_t)KT1 * BM * 128 + (size_t)KT1 * 64 * 128 + (size_t)BM * 128 + (size_t)2 * b.CN * 128 + (size_t)(4 * b.C1 + b.CN) * 4;
        const size_t ring1 = 3 * (size_t)(BM + b.C1) * 128;
        if (ring1 > lds1) lds1 = ring1;
        const void* f1;
        if (image) {
            if (dt == DT_BF16) f1 = b.C1 == 128 ? reinterpret_cast<const void*>(bneck231_kernel<bf16, 64, 128, 128>)
                                  : b.CN == 64 ? reinterpret_cast<const void*>(bneck231_kernel<bf16, 128, 64, 64>) : reinterpret_cast<const void*>(bneck231_kernel<bf16, 128, 64, 128>);
            else f1 = b.C1 == 128 ? reinterpret_cast<const void*>(bneck231_kernel<f16, 64, 128, 128>)
                    : b.CN == 64 ? reinterpret_cast<const void*>(bneck231_kernel<f16, 128, 64, 64>) : reinterpret_cast<const void*>(bneck231_kernel<f16, 128, 64, 128>);
        } else {
            if (dt == DT_BF16) f1 = b.C1 == 128 ? reinterpret_cast<const void*>(bneck231r_kernel<bf16, 64, 128, 128>)
                                  : b.CN == 64 ? reinterpret_cast<const void*>(bneck231r_kernel<bf16, 128, 64, 64>) : reinterpret_cast<const void*>(bneck231r_kernel<bf16, 128, 64, 128>);
            else f1 = b.C1 == 128 ? reinterpret_cast<const void*>(bneck231r_kernel<f16, 64, 128, 128>)
                    : b.CN == 64 ? reinterpret_cast<const void*>(bneck231r_kernel<f16, 128, 64, 64>) : reinterpret_cast<const void*>(bneck231r_kernel<f16, 128, 64, 128>);
        }
        static const bool no_halo = dev_env("HCM_NO_BNECK_HALO") != nullptr;
        // round 4: 128 mid channels (layer2) on 128-PIXEL tiles, one 92 KB workgroup per CU instead of two 64-pixel ones.  A tile streams the block's
        // 557 KB of weights (3x3 295 KB + expansion 131 KB + next reduction 131 KB) from L2 whatever its size: 2048 64-pixel tiles = 1.14 GB per launch
        // (the pair at B = 64) against 0.39 GB of activations -- the launch was bound by that stream (134 us = 2.9 TB/s of HBM traffic, 0.28 of the
        // matrix rate).  HCM_BNECK128_BM64=1 (development build): the 64-pixel tiles.
        static const bool bm64 = dev_env("HCM_BNECK128_BM64") != nullptr;
        if (!image && !no_halo && !bm64 && b.C1 == 128 && b.CN == 128 && b.stride == 1 && d.W % 16 == 0 && 128 % d.W == 0 && (d.Ho * d.Wo) % 128 == 0 &&
            (long)d.groups * (d.M / 128) >= 192) {
            const size_t halo = (size_t)KT1 * ((((128 / d.W + 2) * (d.W + 2)) + 7) & ~7) * 128 + (size_t)kHaloRing * 8192;
            size_t lds = (size_t)KT1 * 128 * 128 + (size_t)KT1 * 64 * 128 + (size_t)128 * 128 + (size_t)b.CN * 128 + (size_t)(4 * b.C1 + b.CN) * 4;
            if (halo > lds) lds = halo;
            if (lds <= 96 * 1024) {
                const void* fb = dt == DT_BF16 ? reinterpret_cast<const void*>(bneck231r_kernel<bf16, 128, 128, 128, 0, false, kHaloRing, 1>)
                                               : reinterpret_cast<const void*>(bneck231r_kernel<f16, 128, 128, 128, 0, false, kHaloRing, 1>);
                hipError_t eb = hipFuncSetAttribute(fb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (eb != hipSuccess) return eb;
                void* ab[] = {&qq};
                return hipLaunchKernel(fb, dim3(d.M / 128, d.groups), dim3(512), ab, lds, s);
            }
        }
        bool halo_on = false;
        if (!image && !no_halo && b.stride == 1 && d.W % 16 == 0 && BM % d.W == 0 && (d.Ho * d.Wo) % BM == 0) {
            const size_t halo = (size_t)KT1 * ((((BM / d.W + 2) * (d.W + 2)) + 7) & ~7) * 128 + (size_t)kHaloRing * 8192;
            if (halo <= 80 * 1024) {
                halo_on = true;
                // (the 3-deep ring of whole tap tiles is not used by this form: the phase-B regions or the halo block + weight ring decide)
                lds1 = (size_t)KT1 * BM * 128 + (size_t)KT1 * 64 * 128 + (size_t)BM * 128 + (size_t)2 * b.CN * 128 + (size_t)(4 * b.C1 + b.CN) * 4;
                if (halo > lds1) lds1 = halo;
                if (dt == DT_BF16) f1 = b.C1 == 128 ? reinterpret_cast<const void*>(bneck231r_kernel<bf16, 64, 128, 128, 0, false, kHaloRing>)
                                      : b.CN == 64 ? reinterpret_cast<const void*>(bneck231r_kernel<bf16, 128, 64, 64, 0, false, kHaloRing>)
                                                   : reinterpret_cast<const void*>(bneck231r_kernel<bf16, 128, 64, 128, 0, false, kHaloRing>);
                else f1 = b.C1 == 128 ? reinterpret_cast<const void*>(bneck231r_kernel<f16, 64, 128, 128, 0, false, kHaloRing>)
                        : b.CN == 64 ? reinterpret_cast<const void*>(bneck231r_kernel<f16, 128, 64, 64, 0, false, kHaloRing>)
                                     : reinterpret_cast<const void*>(bneck231r_kernel<f16, 128, 64, 128, 0, false, kHaloRing>);
            }
        }
#ifdef HCM_DEV_KNOBS
        if (!image && prof_on() && dt == DT_F16 && b.C1 == 64 && b.CN == 64)
            f1 = halo_on ? reinterpret_cast<const void*>(bneck231r_kernel<f16, 128, 64, 64, 0, true, kHaloRing>) : reinterpret_cast<const void*>(bneck231r_kernel<f16, 128, 64, 64, 0, true>);
        if (!image && prof_on() && dt == DT_F16 && b.C1 == 128 && b.CN == 128)
            f1 = halo_on ? reinterpret_cast<const void*>(bneck231r_kernel<f16, 64, 128, 128, 0, true, kHaloRing>) : reinterpret_cast<const void*>(bneck231r_kernel<f16, 64, 128, 128, 0, true>);
#endif
        (void)halo_on;
        hipError_t e1 = hipFuncSetAttribute(f1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e1 != hipSuccess) return e1;
        void* a1[] = {&qq};
        return hipLaunchKernel(f1, dim3((d.M + BM - 1) / BM, d.groups), dim3(512), a1, lds1, s);
    }
    size_t lds = (size_t)KT1 * BM * 128 + (size_t)KT1 * 128 * 128 + (size_t)(BM / 2) * (128 + 4) * 4;
    const size_t ring = 3 * (size_t)(BM + b.C1) * 128;     // phase A: 3-deep ring
    if (ring > lds) lds = ring;
    const void* fn;
    if (dt == DT_BF16) fn = b.C1 == 64 ? reinterpret_cast<const void*>(bneck23_kernel<bf16, 128, 64>) : reinterpret_cast<const void*>(bneck23_kernel<bf16, 64, 128>);
    else fn = b.C1 == 64 ? reinterpret_cast<const void*>(bneck23_kernel<f16, 128, 64>) : reinterpret_cast<const void*>(bneck23_kernel<f16, 64, 128>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    void* args[] = {&q};
    return hipLaunchKernel(fn, dim3((d.M + BM - 1) / BM, d.groups), dim3(512), args, lds, s);
}


// ------------------------------------------------------------------------------------------------------------------------------------
// depth_l3_kernel: a run of identity bottlenecks of the depth GroupNorm trunk's layer3 (8 x 8 maps, 512 channels, 128 mid channels, 16 groups
// per trunk; habitat ResNetEncoder as used at resnet_encoders.py:27-62) in ONE launch.  As separate launches every conv of these blocks is a
// 64 x 128-tile GEMM over a few dozen workgroups whose 14 us are fixed costs (launch, first-tile latency, the fused-GroupNorm epilogue's LDS
// image and barriers), 15 launches in a row.  Here a workgroup owns ONE sample of one trunk for the whole run: the block input (64 pixels x
// 512 channels = 64 KB) and the two mid tensors (16 KB each) stay in LDS in the MFMA operand layout (64-channel blocks of 128-byte pixel rows,
// XOR-swizzled), the weights come from L2 straight into registers as MFMA fragments (16 bytes per lane: 8 k of one output channel), a few K
// steps ahead, and GroupNorm needs no LDS at all -- a wave's accumulators hold whole groups (8 channels x 64 pixels for the 128-wide convs,
// 32 x 64 for the expansion), so the statistics are a butterfly over lanes.  3 barriers per block.  Same f32 operations as the fused epilogue
// of igemm_dma_kernel (statistics of the f32 accumulators, (v - mean) * rstd * gamma + beta, + identity, ReLU, one rounding) in a different
// (fixed) summation order: equal to that path to f32 round-off of the statistics, not bit for bit.
struct DepthL3Dev {
    const char* x; char* y; int ld, nblocks;
    const char* w1[6]; const char* w2[6]; const char* w3[6];
    const float* g1[6]; const float* b1[6]; const float* g2[6]; const float* b2[6]; const float* g3[6]; const float* b3[6];
    float eps1[6], eps2[6], eps3[6];
};

template <typename T, bool PROF = false>
__global__ __launch_bounds__(512) void depth_l3_kernel(DepthL3Dev p) {
    // phase timing (HCM_IGEMM_PROF=1, development build; read through hcm_debug_igemm_prof): per-wave cycle totals [0] input staging, [1] conv1 K loop,
    // [2] its GroupNorm + barrier, [3] conv2 K loop, [4] its GroupNorm + barrier, [5] conv3 + GroupNorm + identity + barrier + output, [6] waves
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
    if constexpr (PROF) t_prev = prof_now();
    auto lap = [&](int slot) {
        if constexpr (PROF) { const unsigned long long t = prof_now(); pt[slot] += t - t_prev; t_prev = t; }
    };
    constexpr int XO = 0, O1 = 65536, O2 = 81920;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, g = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    // ---- block input -> LDS
    {
        const char* src = p.x + ((size_t)b * 64 * p.ld + (size_t)g * 512) * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + 512 * i;
            const int px = c >> 6, ch = c & 63;                     // 64 chunks of 8 channels per pixel
            const uint4 v = *reinterpret_cast<const uint4*>(src + ((size_t)px * p.ld + ch * 8) * 2);
            *reinterpret_cast<uint4*>(smem + XO + (ch >> 3) * 8192 + px * 128 + (((ch & 7) ^ (px & 7)) << 4)) = v;
        }
    }
    __syncthreads();
    // the lane's A-fragment rows: pixel r = j * 16 + fr
    auto a_addr = [&](int base, int r, int kstep) {             // k step of 32 channels: 64-channel block kstep / 2, half kstep % 2
        const int chunk = (kstep & 1) * 4 + fg;
        return base + (kstep >> 1) * 8192 + r * 128 + ((chunk ^ (r & 7)) << 4);
    };
    // per-lane sum over the 16 lanes of its K group pair and both K groups of a pair (bits 0-4), or over the whole wave
    auto red32 = [&](float v) {
#pragma unroll
        for (int o = 1; o <= 16; o <<= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    auto store4 = [&](int base, int r, int c, const float (&v)[4]) {        // 4 consecutive channels c .. c + 3 of pixel r into an operand block
        T o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) Tr<T>::st(&o4[e], v[e]);
        *reinterpret_cast<uint2*>(smem + base + (c >> 6) * 8192 + r * 128 + ((((c & 63) >> 3) ^ (r & 7)) << 4) + (c & 7) * 2) = *reinterpret_cast<const uint2*>(o4);
    };
    // ---- the weight stream.  Per block a wave consumes 68 PIECES of 1 KB -- its 16 output rows x 32 k (64-byte rows) of conv1's 16 K steps,
    // conv2's 36, and conv3's 4 K steps x 4 fragments -- in a fixed order; every wave has its own R-slot ring in LDS (no barrier: producer and
    // consumer are the same wave), filled by `buffer_load ... lds` R - 1 pieces ahead, across the conv and block boundaries, with counted waits.
    // 64-byte rows: 16-byte position = K group ^ g(row quad), g = (0, 3, 2, 1), as in the bottleneck kernel's quarter-tap units.
    constexpr int R = 8, PPB = 68, WR = 98304;
    const int total = p.nblocks * PPB;
    const unsigned ring = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + WR + wave * (R * 1024);
    const int drow = lane >> 2, dsrc = (lane & 3) ^ ((4 - (drow >> 2)) & 3);
    // per-lane byte offsets of a piece inside its weight matrix, without the K step
    const unsigned vo1 = (unsigned)((wave * 16 + drow) * 512 + dsrc * 8) * 2u, vo2 = (unsigned)((wave * 16 + drow) * 1152 + dsrc * 8) * 2u,
                   vo3 = (unsigned)((wave * 64 + drow) * 128 + dsrc * 8) * 2u;
    v4i_t rs1, rs2, rs3, rs1n;                                  // buffer resources of this block's three matrices and the next block's first
    int par = 0;                                                // ring position of the block's first piece: (blk * 68) % 8
    bool has_next = false;
    auto load_rsrc = [&](int blk) {
        rs1 = make_rsrc(p.w1[blk] + (size_t)g * 128 * 512 * 2, 128u * 512u * 2u);
        rs2 = make_rsrc(p.w2[blk] + (size_t)g * 128 * 1152 * 2, 128u * 1152u * 2u);
        rs3 = make_rsrc(p.w3[blk] + (size_t)g * 512 * 128 * 2, 512u * 128u * 2u);
        has_next = blk + 1 < p.nblocks;
        rs1n = make_rsrc(p.w1[has_next ? blk + 1 : blk] + (size_t)g * 128 * 512 * 2, 128u * 512u * 2u);
        par = (blk & 1) * 4;
    };
    auto issue = [&](int idx) {                                 // piece idx of the current block (compile-time after unrolling); >= 68: of the next
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring + (unsigned)(((par + idx) & (R - 1)) * 1024));
        if (idx >= PPB) { if (has_next) dma16(dst, vo1 + (unsigned)(idx - PPB) * 64u, rs1n); }
        else if (idx < 16) dma16(dst, vo1 + (unsigned)idx * 64u, rs1);
        else if (idx < 52) dma16(dst, vo2 + (unsigned)(idx - 16) * 64u, rs2);
        else dma16(dst, vo3 + (unsigned)(((idx - 52) & 3) * 16 * 128 * 2 + ((idx - 52) >> 2) * 64), rs3);
    };
    // the fragment of piece idx: wait for it and read it; the slot of the piece BEFORE it (whose fragment the previous MFMAs have consumed, so its
    // read is long complete) is refilled with the piece R - 1 ahead of this one -- R - 2 pieces in flight behind the one waited for
    // ROUND 6: "its read is long complete" was an assumption about the compiler, not a guarantee.  The MFMAs that consume piece idx - 1 depend on
    // nothing in take(idx), so hipcc is free to sink them -- and the lgkmcnt wait in front of them -- BELOW take(idx)'s request: with
    // -ffp-contract=on it did exactly that in conv3's fragment loop (ds_read slot 1; request -> slot 0; ds_read slot 2; request -> slot 1; only then
    // s_waitcnt lgkmcnt), i.e. a request into a slot whose ds_read was still in flight.  Harmless on an idle CU (a request needs >= 250 cycles to
    // land), but with the other chains of a step competing for the CU's LDS queue the read could lose: configs[4] at B = 128 disagreed with itself
    // run to run (tests/test_fullsize_gpu.py::test_config4_full_size_properties; bisected to this kernel with per-file objects and HCM_NO_DEPTH_L3).
    // The LDS wait now stands in the source, in front of the request: everything this wave has read so far -- piece idx - 1's fragment and the
    // A fragments the next MFMAs need anyway -- has arrived before the slot is handed back.
    auto take = [&](int idx) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (idx + R - 2 < PPB || has_next) wait_vmcnt<R - 2>(); else wait_vmcnt<0>();
        const uint4 wb = *reinterpret_cast<const uint4*>(smem + WR + wave * (R * 1024) + ((par + idx) & (R - 1)) * 1024 + fr * 64 +
                                                         ((fg ^ ((4 - (fr >> 2)) & 3)) << 4));
        issue(idx + R - 1);
        return wb;
    };
    load_rsrc(0);
#pragma unroll
    for (int i = 0; i < R - 1; ++i) issue(i);
    lap(0);
    for (int blk = 0; blk < p.nblocks; ++blk) {
        if (blk > 0) load_rsrc(blk);
        // =================== conv1: 1x1, 512 -> 128 (wave: channels wave * 16 .. + 15, all 64 pixels)
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // (fragment reads one K step ahead, by hand: the asm waits / requests of take() are barriers to the compiler's own pipelining)
        {
            uint4 xa[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xa[0][j] = *reinterpret_cast<const uint4*>(smem + a_addr(XO, j * 16 + fr, 0));
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (ks + 1 < 16) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) xa[(ks + 1) & 1][j] = *reinterpret_cast<const uint4*>(smem + a_addr(XO, j * 16 + fr, ks + 1));
                }
                const uint4 wb = take(ks);
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<T>::run(acc[j], wb, xa[ks & 1][j]);
            }
        }
        lap(1);
        auto gn128 = [&](const float* gamma, const float* beta, float eps, int dst) {
            // groups of 8 channels: the lane's 4 channels with those of its neighbour K group (fg ^ 1), over 64 pixels
            const int c = wave * 16 + fg * 4;
            const float4 ga = *reinterpret_cast<const float4*>(gamma + g * 128 + c), be = *reinterpret_cast<const float4*>(beta + g * 128 + c);
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float v = acc[j][e]; a += v; q += v * v; }
            a = red32(a); q = red32(q);
            const float inv = 1.0f / 512.0f;
            const float mean = a * inv;
            const float rstd = rsqrtf(relu_f(q * inv - mean * mean) + eps);
            const float gaa[4] = {ga.x, ga.y, ga.z, ga.w}, bea[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = relu_f((acc[j][e] - mean) * rstd * gaa[e] + bea[e]);
                store4(dst, j * 16 + fr, c, v);
            }
        };
        gn128(p.g1[blk], p.b1[blk], p.eps1[blk], O1);
        __syncthreads();
        lap(2);
        // =================== conv2: 3x3 (pad 1), 128 -> 128 over the 8 x 8 map: k = tap * 128 + ci
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            uint4 xa[2][4];
            auto ld2 = [&](int step, uint4 (&dst)[4]) {            // A fragments of K step `step` = (tap, 32-channel quarter)
                const int tap = step >> 2, kc = step & 3;
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = j * 16 + fr;
                    const int sy = (r >> 3) + dy, sx = (r & 7) + dx;
                    const bool ok = ((unsigned)sy < 8u) & ((unsigned)sx < 8u);
                    uint4 v = *reinterpret_cast<const uint4*>(smem + a_addr(O1, ok ? sy * 8 + sx : 0, kc));
                    if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
                    dst[j] = v;
                }
            };
            ld2(0, xa[0]);
#pragma unroll
            for (int ks = 0; ks < 36; ++ks) {
                if (ks + 1 < 36) ld2(ks + 1, xa[(ks + 1) & 1]);
                const uint4 wb = take(16 + ks);
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<T>::run(acc[j], wb, xa[ks & 1][j]);
            }
        }
        lap(3);
        gn128(p.g2[blk], p.b2[blk], p.eps2[blk], O2);
        __syncthreads();
        lap(4);
        // =================== conv3: 1x1, 128 -> 512, GroupNorm (groups of 32 channels = fragment pairs), + identity, ReLU, in place
        {
            f32x4 acc3[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                uint4 xa[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xa[j] = *reinterpret_cast<const uint4*>(smem + a_addr(O2, j * 16 + fr, ks));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint4 wb = take(52 + ks * 4 + i);
#pragma unroll
                    for (int j = 0; j < 4; ++j) Mma<T>::run(acc3[i][j], wb, xa[j]);
                }
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {                       // the wave's two groups: fragments 2 pr, 2 pr + 1
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float v = acc3[pr * 2 + ii][j][e]; a += v; q += v * v; }
                a = wave_sum(a); q = wave_sum(q);
                const float inv = 1.0f / 2048.0f;
                const float mean = a * inv;
                const float rstd = rsqrtf(relu_f(q * inv - mean * mean) + p.eps3[blk]);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int i = pr * 2 + ii;
                    const int c = wave * 64 + i * 16 + fg * 4;
                    const float4 ga = *reinterpret_cast<const float4*>(p.g3[blk] + g * 512 + c), be = *reinterpret_cast<const float4*>(p.b3[blk] + g * 512 + c);
                    const float gaa[4] = {ga.x, ga.y, ga.z, ga.w}, bea[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = j * 16 + fr;
                        char* xp = smem + XO + (c >> 6) * 8192 + r * 128 + ((((c & 63) >> 3) ^ (r & 7)) << 4) + (c & 7) * 2;
                        const uint2 idt = *reinterpret_cast<const uint2*>(xp);
                        const T* it = reinterpret_cast<const T*>(&idt);
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = relu_f((acc3[i][j][e] - mean) * rstd * gaa[e] + bea[e] + Tr<T>::ld(it + e));
                        T o4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) Tr<T>::st(&o4[e], v[e]);
                        *reinterpret_cast<uint2*>(xp) = *reinterpret_cast<const uint2*>(o4);
                    }
                }
            }
        }
        __syncthreads();
        lap(5);
    }
    // ---- block output -> y
    {
        char* dst = p.y + ((size_t)b * 64 * p.ld + (size_t)g * 512) * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + 512 * i;
            const int px = c >> 6, ch = c & 63;
            *reinterpret_cast<uint4*>(dst + ((size_t)px * p.ld + ch * 8) * 2) =
                *reinterpret_cast<const uint4*>(smem + XO + (ch >> 3) * 8192 + px * 128 + (((ch & 7) ^ (px & 7)) << 4));
        }
    }
    if constexpr (PROF) {
        lap(5);
        if (lane == 0) {
            unsigned long long* slot = g_igemm_prof[((blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) & (kProfSlots - 1)];
            for (int i = 0; i < 6; ++i) atomicAdd(&slot[i], pt[i]);
            atomicAdd(&slot[6], 1ull);
            atomicAdd(&slot[7], (unsigned long long)p.nblocks);
        }
    }
}

hipError_t launch_depth_l3(const DepthL3& d, int dt, hipStream_t s) {
    if ((dt != DT_F16 && dt != DT_BF16) || d.nblocks < 1 || d.nblocks > 6 || d.B < 1 || d.groups < 1 || (d.ld % 8) || d.ld < d.groups * 512 || !d.x || !d.y)
        return hipErrorInvalidValue;
    DepthL3Dev q;
    q.x = (const char*)d.x; q.y = (char*)d.y; q.ld = d.ld; q.nblocks = d.nblocks;
    for (int i = 0; i < 6; ++i) {
        q.w1[i] = (const char*)d.w1[i]; q.w2[i] = (const char*)d.w2[i]; q.w3[i] = (const char*)d.w3[i];
        q.g1[i] = d.g1[i]; q.b1[i] = d.b1[i]; q.g2[i] = d.g2[i]; q.b2[i] = d.b2[i]; q.g3[i] = d.g3[i]; q.b3[i] = d.b3[i];
        q.eps1[i] = d.eps1[i]; q.eps2[i] = d.eps2[i]; q.eps3[i] = d.eps3[i];
    }
    const void* fn = dt == DT_BF16 ? reinterpret_cast<const void*>(depth_l3_kernel<bf16>) : reinterpret_cast<const void*>(depth_l3_kernel<f16>);
#ifdef HCM_DEV_KNOBS
    if (prof_on() && dt == DT_F16) fn = reinterpret_cast<const void*>(depth_l3_kernel<f16, true>);
#endif
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    void* args[] = {&q};
    return hipLaunchKernel(fn, dim3(d.B, d.groups), dim3(512), args, 98304 + 8 * 8 * 1024, s);
}

}  // namespace hcm
