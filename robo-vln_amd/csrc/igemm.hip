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

    // ---- epilogue: lane holds channels n..n+3 of pixel m ----
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * (BM / 2) + j * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * (BN / 2) + i * 16 + fg * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.bias) {
                const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            if (p.res) {
                const T* rp = reinterpret_cast<const T*>(p.res) + (size_t)m * p.ldr + n;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += Tr<T>::ld(rp + e);
            }
            if (p.act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (p.act == ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            }
            if (p.out_f32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (size_t)m * p.ldy + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (size_t)m * p.ldy + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                T o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) Tr<T>::st(&o4[e], v[e]);
                *reinterpret_cast<uint2*>(reinterpret_cast<T*>(p.y) + (size_t)m * p.ldy + n) = *reinterpret_cast<const uint2*>(o4);
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
    const size_t lds = 2 * (size_t)(BM + BN) * 128;
    hipLaunchKernelGGL((igemm_kernel<T, BM, BN>), dim3(grid), dim3(256), lds, s, d);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_t(const IGemmDev& d, hipStream_t s) {
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
