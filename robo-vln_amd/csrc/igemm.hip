// Implicit-GEMM convolution / linear layer on MFMA (gfx950).
//
//   D[n][m] = sum_k W[n][k] * A[m][k]     W = weights [N][Kp], A = im2col(x) gathered on the fly (NHWC)
//
// The weights are the MFMA "A" operand and the activations the "B" operand, so that in the 16x16
// accumulator tile (col = lane&15, row = (lane>>4)*4 + reg) a lane holds FOUR CONSECUTIVE output
// channels of ONE output pixel: the epilogue (bias + residual + activation) then stores 8 B (bf16) or
// 16 B (f32) per lane into the NHWC output with no transpose.
//
// Tile: BM output pixels x BN output channels x 128 bytes of K per step (64 bf16 / 32 f32), 4 waves (2x2),
// LDS double-buffered, global->register->LDS staging (the im2col gather needs per-chunk predication, so
// the LDS-DMA path with its lane-linear destination is not used here).  LDS rows are 128 B; the 16-byte
// chunk index is XOR-swizzled with (row & 7), which makes both the ds_write_b128 staging and the
// ds_read_b128 fragment reads bank-conflict free (cdna_hip_programming.md T2).
//
// K ordering inside one MFMA step is permuted consistently for both operands (lane group g consumes the
// g-th 16-byte chunk of the 64-byte half row), which a dot product is invariant to; this lets the f32
// path feed four v_mfma_f32_16x16x4_f32 from one ds_read_b128.
//
// Reference ops replaced: cuDNN conv fwd + BatchNorm(eval) + ReLU of torchvision resnet50, the GN-ResNet
// convs, and every nn.Linear / Conv1d(k=1) on the path.
#include <cstdlib>
#include "kernels.h"
#include "dev.h"

namespace hcm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct IGemmDev {
    const char* x; const char* w; const float* bias; const char* res; char* y;
    int B, H, W, Cin, xC, Ho, Wo, KH, KW, stride, pad;
    int M, N, K, Kp, ldy, ldr, act, out_f32;
    int cin_shift, kw_rcp, tilesM, tilesN;
};

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

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void igemm_kernel(IGemmDev p) {
    constexpr int CH = Tr<T>::CH;          // elements per 16-byte chunk
    constexpr int BK = 8 * CH;             // elements per 128-byte tile row
    constexpr int TM = BM / 32;            // 16-wide pixel tiles per wave
    constexpr int TN = BN / 32;            // 16-wide channel tiles per wave
    constexpr int A_IT = BM / 32;          // 16-byte chunks staged per thread
    constexpr int B_IT = BN / 32;
    constexpr int TILE_BYTES = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // XCD-aware tile order: block b runs on XCD b%8; give each XCD whole pixel-tiles (all channel tiles of
    // one pixel tile back to back) so the gathered activation rows are re-read from that XCD's L2.
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int local = bid >> 3;
    const int tile_n = local % p.tilesN;
    const int tile_m = (local / p.tilesN) * 8 + xcd;
    if (tile_m >= p.tilesM) return;
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
            a_ix0[i] = ox * p.stride - p.pad;
        } else {
            a_pix[i] = -1; a_iy0[i] = 0; a_ix0[i] = 0;
        }
    }
    const T* xg = reinterpret_cast<const T*>(p.x);
    const T* wg = reinterpret_cast<const T*>(p.w);
    const bool spatial = (p.KH * p.KW) > 1;

    uint4 ra[A_IT], rb[B_IT];
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
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
            const bool ok = kvalid && a_pix[i] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = *reinterpret_cast<const uint4*>(xg + (size_t)(a_pix[i] + iy * p.W + ix) * p.xC + ci);
            ra[i] = v;
        }
        const bool kvalid_w = k < p.Kp;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = n0 + r0 + 32 * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kvalid_w && n < p.N) v = *reinterpret_cast<const uint4*>(wg + (size_t)n * p.Kp + k);
            rb[i] = v;
        }
    };
    auto store_tiles = [&](int buf) {
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
    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int fr = lane & 15;      // fragment row (pixel for A-tile, channel for W-tile)
    const int fg = lane >> 4;      // 16-byte chunk within the 64-byte half row

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
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
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

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
            const int r = wm * (BM / 2) + j * 16 + fr;
            const int cc = wn * (BN / 2) + i * 16 + fg * 4;
            *reinterpret_cast<float4*>(sc + r * LDC + cc) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    __syncthreads();
    constexpr int TPR = BN / 8;            // threads per tile row
    constexpr int RPP = 256 / TPR;         // rows per pass
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
        if (p.res) {
            const T* rp = reinterpret_cast<const T*>(p.res) + (size_t)m * p.ldr + n;
            if constexpr (sizeof(T) == 2) {
                if (wide16r) {
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
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
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

template <typename T, int BM, int BN>
static hipError_t launch_cfg(IGemmDev d, hipStream_t s) {
    d.tilesM = (d.M + BM - 1) / BM;
    d.tilesN = (d.N + BN - 1) / BN;
    const int tm8 = (d.tilesM + 7) / 8;
    const int grid = tm8 * 8 * d.tilesN;
    size_t lds = 2 * (size_t)(BM + BN) * 128;
    const size_t lds_c = (size_t)BM * (BN + 4) * 4;           // f32 output-tile image of the epilogue
    if (lds_c > lds) lds = lds_c;
    if (lds > 64 * 1024) {
        static bool attr_done = false;                        // one flag per template instantiation
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_kernel<T, BM, BN>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
    }
    hipLaunchKernelGGL((igemm_kernel<T, BM, BN>), dim3(grid), dim3(256), lds, s, d);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_forced(const IGemmDev& d, int cfg, hipStream_t s) {
    switch (cfg) {
        case 0: return launch_cfg<T, 128, 128>(d, s);
        case 1: return launch_cfg<T, 128, 64>(d, s);
        case 2: return launch_cfg<T, 64, 64>(d, s);
        case 3: return launch_cfg<T, 64, 32>(d, s);
        case 4: return launch_cfg<T, 128, 32>(d, s);
        case 5: return launch_cfg<T, 64, 128>(d, s);
        default: return hipErrorInvalidValue;
    }
}

template <typename T>
static hipError_t launch_t(const IGemmDev& d, hipStream_t s) {
    static const char* force = getenv("HCM_IGEMM_FORCE");
    if (force) return launch_forced<T>(d, atoi(force), s);
    // tile choice: largest tile that still gives the 256 CUs at least ~2 workgroups each
    const long blocks128 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128);
    const long blocks12864 = (long)((d.M + 127) / 128) * ((d.N + 63) / 64);
    const long blocks64 = (long)((d.M + 63) / 64) * ((d.N + 63) / 64);
    if (d.N >= 128 && blocks128 >= 512) return launch_cfg<T, 128, 128>(d, s);
    if (d.N >= 64 && blocks12864 >= 512) return launch_cfg<T, 128, 64>(d, s);
    if (d.N <= 32) {
        if ((long)((d.M + 127) / 128) >= 512) return launch_cfg<T, 128, 32>(d, s);
        return launch_cfg<T, 64, 32>(d, s);
    }
    if (blocks64 >= 256 || d.N <= 64) return launch_cfg<T, 64, 64>(d, s);
    return launch_cfg<T, 64, 32>(d, s);
}

hipError_t launch_igemm(const IGemm& g, int dt, hipStream_t s) {
    IGemmDev d;
    d.x = (const char*)g.x; d.w = (const char*)g.w; d.bias = g.bias; d.res = (const char*)g.res; d.y = (char*)g.y;
    d.B = g.B; d.H = g.H; d.W = g.W; d.Cin = g.Cin; d.xC = g.xC ? g.xC : g.Cin;
    d.Ho = g.Ho; d.Wo = g.Wo; d.KH = g.KH; d.KW = g.KW; d.stride = g.stride; d.pad = g.pad;
    d.M = g.M; d.N = g.N; d.K = g.K; d.Kp = g.Kp ? g.Kp : g.K;
    d.ldy = g.ldy ? g.ldy : g.N; d.ldr = g.ldr ? g.ldr : g.N; d.act = g.act; d.out_f32 = g.out_f32;
    const int CH = dt_chunk(dt);
    d.cin_shift = 0;
    d.kw_rcp = (65536 + g.KW - 1) / g.KW;
    if (g.KH * g.KW > 1) {
        if (g.Cin & (g.Cin - 1)) return hipErrorInvalidValue;     // spatial kernels need power-of-two Cin
        while ((1 << d.cin_shift) < g.Cin) ++d.cin_shift;
    }
    if (d.M <= 0 || d.N <= 0 || d.K <= 0) return hipErrorInvalidValue;
    if ((g.Cin % CH) || (d.xC % CH) || (d.K % CH) || (d.Kp % CH) || (d.N % 4) || (d.ldy % 4) || (d.ldr % 4))
        return hipErrorInvalidValue;
    if (dt == DT_BF16) return launch_t<bf16>(d, s);
    if (dt == DT_F16) return launch_t<f16>(d, s);
    if (dt == DT_F32) return launch_t<float>(d, s);
    return hipErrorInvalidValue;
}

}  // namespace hcm
