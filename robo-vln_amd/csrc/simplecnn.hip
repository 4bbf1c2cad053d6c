// SimpleDepthCNN's first layer -- Conv2d(1, 32, kernel 8, stride 4) + bias + ReLU (models/encoders/simple_cnns.py:76-84, the depth branch
// of :104-125) -- straight from the raw f32 depth frame (B, H, H, 1): HBM-bound (4 B in and ~4 B out per input pixel, 64 MACs per output
// value), so one pass over the frame is the whole cost.  The generic route (convert the frame to 16 bit, then an LDS-DMA implicit GEMM over
// the converted copy) reads and writes the frame once more than necessary: 58 us at B = 256 against 27 us here.
//
// A workgroup (4 waves) owns RB = 4 output rows of one frame: the 4*RB + 4 input rows they cover are read once as float4, rounded to the
// storage type exactly as the conversion kernel rounds them (RNE) and parked in LDS; an output pixel's operand for kernel row kh is then
// the 8 consecutive values at (4 oy + kh, 4 ox ..), i.e. one 16-byte MFMA operand chunk.  K = 64 = (kh, kw) in two 16x16x32 steps, weights
// [32][64] as the A operand (4 chunks per lane, loaded once), pixels as B: a lane ends with 4 consecutive channels of one pixel, the same
// instruction, operand roles and k order as the implicit GEMM -- bit-identical to it.
#include <cstdlib>
#include "kernels.h"
#include "dev.h"

namespace hcm {

typedef float s_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 s_bf16x8 __attribute__((ext_vector_type(8)));
template <typename T> struct SMma;
template <> struct SMma<bf16> {
    static __device__ __forceinline__ void run(s_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s_bf16x8, a), __builtin_bit_cast(s_bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct SMma<f16> {
    static __device__ __forceinline__ void run(s_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), acc, 0, 0, 0);
    }
};

constexpr int S_RB = 4;                  // output rows per workgroup
constexpr int S_ROWS = 4 * S_RB + 4;     // input rows they cover

template <typename T>
__global__ __launch_bounds__(256) void depth_conv8x8s4_kernel(const float* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                                                             T* __restrict__ y, int H, int h1, int act) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sx = reinterpret_cast<T*>(smem);                        // [S_ROWS][H]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int oy0 = blockIdx.x * S_RB, b = blockIdx.y;
    const float* xb = x + (size_t)b * H * H + (size_t)oy0 * 4 * H;
    const int rows_in = min(S_ROWS, H - oy0 * 4);
    const int q4 = H >> 2;                                     // float4 per input row
    for (int e = tid; e < rows_in * q4; e += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)e * 4);      // rows are contiguous: e*4 = row*H + col
        T o[4];
        Tr<T>::st(&o[0], v.x); Tr<T>::st(&o[1], v.y); Tr<T>::st(&o[2], v.z); Tr<T>::st(&o[3], v.w);
        *reinterpret_cast<uint2*>(sx + (size_t)e * 4) = *reinterpret_cast<const uint2*>(o);
    }
    // weights: A operand, channel i*16 + fr, k = ks*32 + fg*8 ..
    uint4 wa[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wa[i][ks] = *reinterpret_cast<const uint4*>(w + (size_t)(i * 16 + fr) * 64 + ks * 32 + fg * 8);
    float4 b4[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) b4[i] = bias ? *reinterpret_cast<const float4*>(bias + i * 16 + fg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int rows_out = min(S_RB, h1 - oy0);
    const int npix = rows_out * h1;
    T* yb = y + ((size_t)b * h1 + oy0) * h1 * 32;
    for (int f = wave; f * 16 < npix; f += 4) {
        const int p = f * 16 + fr;                              // this lane's pixel as the B operand row
        const int pc = min(p, npix - 1);
        const int r = pc / h1, ox = pc - r * h1;
        s_f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const T* src = sx + (size_t)(4 * r + ks * 4 + fg) * H + 4 * ox;          // 8-byte aligned
            const uint2 lo = *reinterpret_cast<const uint2*>(src);
            const uint2 hi = *reinterpret_cast<const uint2*>(src + 4);
            const uint4 xb4 = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
            for (int i = 0; i < 2; ++i) SMma<T>::run(acc[i], wa[i][ks], xb4);
        }
        // acc[i][e]: channel i*16 + fg*4 + e of pixel f*16 + fr
        if (p < npix) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v0 = acc[i][0] + b4[i].x, v1 = acc[i][1] + b4[i].y, v2 = acc[i][2] + b4[i].z, v3 = acc[i][3] + b4[i].w;
                if (act == ACT_RELU) { v0 = relu_f(v0); v1 = relu_f(v1); v2 = relu_f(v2); v3 = relu_f(v3); }
                T o[4];
                Tr<T>::st(&o[0], v0); Tr<T>::st(&o[1], v1); Tr<T>::st(&o[2], v2); Tr<T>::st(&o[3], v3);
                *reinterpret_cast<uint2*>(yb + (size_t)p * 32 + i * 16 + fg * 4) = *reinterpret_cast<const uint2*>(o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// SimpleDepthCNN's three convolutions in ONE launch (round 5; models/encoders/simple_cnns.py:76-100, depth branch :104-125):
//   Conv 8x8/4 (1 -> 32) + ReLU -> Conv 4x4/2 (32 -> 64) + ReLU -> Conv 3x3/1 (64 -> 32),
// only the raw f32 frame comes in and only the 32-channel map that feeds the Linear goes out.  As three launches the 63 x 63 x 32 and the
// 30 x 30 x 64 map make a round trip through HBM each (configs[3], B = 256: 144 + 107 MB per step of the 0.63 GB the step moved, against
// 0.107 GB of algorithmic bytes).
//
// A workgroup owns a BAND of N3 = 7 rows of the last map of one frame (four bands at 256 x 256) and everything above it: 9 rows of the
// second map, 20 of the first, 84 input rows -- halo rows are recomputed (x 1.27 / x 1.2 of the first two convs' work, which is cheap: the
// path is bounded by the one pass over the frame and by latency).  The first conv reads its operands straight from the f32 frame (no staging);
// LDS holds the first map with 80-byte pixels and the second with 144-byte pixels (16 bytes of padding each: the B-operand reads of 16 consecutive output
// pixels -- 2 or 1 input pixels apart -- then fall on 16 different bank quads).  Weights stay in REGISTERS as A-operand fragments, loaded from
// the fragment-order copies (launch_pack_frag: 1 KB contiguous per fragment): conv2 a wave holds two of the four 16-channel tiles (32
// fragments) and works on every fourth pixel tile, conv3 one of the two (18 fragments).  k order, operand roles, bias + ReLU + the one
// rounding are those of depth_conv8x8s4_kernel / igemm_dma_kernel: BIT-IDENTICAL to the three launches.
constexpr int S3_N3 = 7;                                  // rows of the last map per workgroup
constexpr int S3_P1 = 80, S3_P2 = 144;                    // bytes per pixel of the first / second map in LDS
struct SimpleCnn3Dev {
    const float* x;                                       // [B][H][H] f32
    const void* w0; const float* b0;                      // [32][64] (k = kh*8 + kw)
    const void* w1f; const float* b1;                     // conv2 [64][512] in fragment order
    const void* w2f; const float* b2;                     // conv3 [32][576] in fragment order
    void* y;                                              // [B][h3][h3][32]
    int H, h1, h2, h3, bands;
    unsigned long long* prof;                             // development build (HCM_S3_PROF_PTR): [workgroup][wave][5] s_memtime stamps at the phase boundaries
};
__device__ __forceinline__ unsigned long long s3_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__host__ __device__ inline size_t s3_lds_bytes(int H, int h1, int h2) {
    const int n2 = S3_N3 + 2, n1 = 2 * n2 + 2;
    (void)H;
    return ((size_t)n2 * h2 * S3_P2 + 15) / 16 * 16 + (size_t)n1 * h1 * S3_P1;
}

template <typename T>
__global__ __launch_bounds__(512) void simplecnn3_kernel(SimpleCnn3Dev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = p.H, h1 = p.h1, h2 = p.h2, h3 = p.h3;
    const int band = blockIdx.x % p.bands, b = blockIdx.x / p.bands;
    const int r3_0 = band * S3_N3;
    const int n3 = min(S3_N3, h3 - r3_0), n2 = n3 + 2, n1 = 2 * n2 + 2;
    const int n2max = S3_N3 + 2, n1max = 2 * n2max + 2;
    char* c2map = smem;                                                   // [n2][h2] pixels of 144 B
    char* c1map = smem + ((size_t)n2max * h2 * S3_P2 + 15) / 16 * 16;     // [n1][h1] pixels of 80 B
    (void)n1max;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    unsigned long long st[5] = {0, 0, 0, 0, 0};
    if (p.prof) st[0] = s3_now();

    // conv1 weights (A operand: channel i*16 + fr, k = ks*32 + fg*8 ..) and conv2's fragments of this wave's two channel tiles
    const T* w0 = reinterpret_cast<const T*>(p.w0);
    uint4 wa0[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wa0[i][ks] = *reinterpret_cast<const uint4*>(w0 + (size_t)(i * 16 + fr) * 64 + ks * 32 + fg * 8);
    float4 b40[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) b40[i] = *reinterpret_cast<const float4*>(p.b0 + i * 16 + fg * 4);
    if (p.prof) st[1] = s3_now();

    // ---- conv1: 8x8 / 4, K = 64 in two steps, pixels of the band's n1 rows.  The operand of pixel (r, ox), kernel row kh is the 8 consecutive
    // f32 values x[4r + kh][4ox ..]: read STRAIGHT from the frame (32 contiguous bytes per lane; the 16 lanes of a fragment row cover one 272-byte
    // run, re-reads are L1 hits), rounded to T exactly as the conversion kernel rounds (RNE).  No LDS staging and no barrier in front of the first
    // MFMA, and -- with one workgroup per CU -- the point of it: the loads of TWO pixel tiles per wave are in flight at a time, so the HBM round
    // trips of a band overlap each other and the MFMAs instead of standing in front of them (first version: 7 of 16.5 us per workgroup).
    {
        const float* xb = p.x + (size_t)b * H * H + (size_t)(8 * r3_0) * H;    // input row of conv1 row 2 * r3_0
        const int npix = n1 * h1;
        const int ntile = (npix + 15) >> 4;
        auto issue = [&](int f, float4 (&d)[2][2]) {
            const int pc = min(f * 16 + fr, npix - 1);
            const int r = pc / h1, ox = pc - r * h1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float* src = xb + (size_t)(4 * r + ks * 4 + fg) * H + 4 * ox;
                d[ks][0] = *reinterpret_cast<const float4*>(src);
                d[ks][1] = *reinterpret_cast<const float4*>(src + 4);
            }
        };
        auto finish = [&](int f, const float4 (&d)[2][2]) {
            const int pq = f * 16 + fr;
            s_f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                T o[8];
                Tr<T>::st(&o[0], d[ks][0].x); Tr<T>::st(&o[1], d[ks][0].y); Tr<T>::st(&o[2], d[ks][0].z); Tr<T>::st(&o[3], d[ks][0].w);
                Tr<T>::st(&o[4], d[ks][1].x); Tr<T>::st(&o[5], d[ks][1].y); Tr<T>::st(&o[6], d[ks][1].z); Tr<T>::st(&o[7], d[ks][1].w);
                const uint4 xb4 = *reinterpret_cast<const uint4*>(o);
#pragma unroll
                for (int i = 0; i < 2; ++i) SMma<T>::run(acc[i], wa0[i][ks], xb4);
            }
            if (pq < npix) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    T o[4];
                    Tr<T>::st(&o[0], relu_f(acc[i][0] + b40[i].x)); Tr<T>::st(&o[1], relu_f(acc[i][1] + b40[i].y));
                    Tr<T>::st(&o[2], relu_f(acc[i][2] + b40[i].z)); Tr<T>::st(&o[3], relu_f(acc[i][3] + b40[i].w));
                    *reinterpret_cast<uint2*>(c1map + (size_t)pq * S3_P1 + (i * 16 + fg * 4) * 2) = *reinterpret_cast<const uint2*>(o);
                }
            }
        };
        // software pipeline, three tiles deep: tiles f, f + 8, f + 16 of this wave are in flight while the oldest is multiplied (deeper is SLOWER:
        // all ten tiles of a wave at once took conv1 from 6.7 to 8.3 us per workgroup -- the 32-byte windows of many rows thrash the CU's L1)
        float4 d0[2][2], d1[2][2], d2[2][2];
        int f = wave;
        if (f < ntile) issue(f, d0);
        if (f + 8 < ntile) issue(f + 8, d1);
        for (; f < ntile; f += 24) {
            if (f + 16 < ntile) issue(f + 16, d2);
            finish(f, d0);
            if (f + 8 >= ntile) break;
            if (f + 24 < ntile) issue(f + 24, d0);
            finish(f + 8, d1);
            if (f + 16 >= ntile) break;
            if (f + 32 < ntile) issue(f + 32, d1);
            finish(f + 16, d2);
        }
    }
    // conv2's fragments of this wave's two channel tiles (requested before the barrier: the round trip runs beside the other waves' last tiles)
    const int np = wave & 1, mq = wave >> 1;
    uint4 wa1[2][16];
    {
        const T* w1f = reinterpret_cast<const T*>(p.w1f);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) wa1[t][ks] = *reinterpret_cast<const uint4*>(w1f + (((size_t)ks * 4 + 2 * np + t) * 64 + lane) * 8);
    }
    float4 b41[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) b41[t] = *reinterpret_cast<const float4*>(p.b1 + (2 * np + t) * 16 + fg * 4);
    __syncthreads();                                       // the first map is complete
    if (p.prof) st[2] = s3_now();

    // ---- conv2: 4x4 / 2, one K step per tap (32 channels); a tile's sixteen operand reads are issued together, each feeds two MFMAs (one
    // channel tile per wave with two pixel tiles per round doubles the operand reads: 4.0 -> 5.1 us per workgroup, LDS-bound)
    {
        const int npix = n2 * h2;
        for (int f = mq; f * 16 < npix; f += 4) {
            const int pq = f * 16 + fr;
            const int pc = min(pq, npix - 1);
            const int oy = pc / h2, ox = pc - oy * h2;
            const char* base = c1map + ((size_t)(2 * oy) * h1 + 2 * ox) * S3_P1 + fg * 16;
            uint4 xv[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) xv[ks] = *reinterpret_cast<const uint4*>(base + ((size_t)(ks >> 2) * h1 + (ks & 3)) * S3_P1);
            s_f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) SMma<T>::run(acc[t], wa1[t][ks], xv[ks]);
            if (pq < npix) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    T o[4];
                    Tr<T>::st(&o[0], relu_f(acc[t][0] + b41[t].x)); Tr<T>::st(&o[1], relu_f(acc[t][1] + b41[t].y));
                    Tr<T>::st(&o[2], relu_f(acc[t][2] + b41[t].z)); Tr<T>::st(&o[3], relu_f(acc[t][3] + b41[t].w));
                    *reinterpret_cast<uint2*>(c2map + (size_t)pq * S3_P2 + ((2 * np + t) * 16 + fg * 4) * 2) = *reinterpret_cast<const uint2*>(o);
                }
            }
        }
    }
    // conv3's fragments of this wave's channel tile (the conv2 fragments are dead: their registers are free again)
    const int nt = wave & 1;
    uint4 wa2[18];
    {
        const T* w2f = reinterpret_cast<const T*>(p.w2f);
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) wa2[ks] = *reinterpret_cast<const uint4*>(w2f + (((size_t)ks * 2 + nt) * 64 + lane) * 8);
    }
    const float4 b42 = *reinterpret_cast<const float4*>(p.b2 + nt * 16 + fg * 4);
    __syncthreads();                                       // the second map is complete
    if (p.prof) st[3] = s3_now();

    // ---- conv3: 3x3 / 1, two K steps per tap (64 channels), no activation; two pixel tiles per round (two independent accumulator chains)
    {
        const int npix = n3 * h3;
        const int ntile = (npix + 15) >> 4;
        T* yb = reinterpret_cast<T*>(p.y) + ((size_t)b * h3 + r3_0) * h3 * 32;
        for (int f = mq; f < ntile; f += 8) {
            const bool two = f + 4 < ntile;
            const char* base[2];
            int pq[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pq[u] = (f + 4 * u) * 16 + fr;
                const int pc = min(pq[u], npix - 1);
                const int oy = pc / h3, ox = pc - oy * h3;
                base[u] = c2map + ((size_t)oy * h2 + ox) * S3_P2 + fg * 16;
            }
            s_f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                uint4 xv[2][9];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int k9 = 0; k9 < 9; ++k9) {
                        const int ks = half * 9 + k9, tap = ks >> 1, kh = tap / 3, kw = tap - kh * 3;
                        xv[u][k9] = *reinterpret_cast<const uint4*>(base[u] + ((size_t)kh * h2 + kw) * S3_P2 + (ks & 1) * 64);
                    }
#pragma unroll
                for (int k9 = 0; k9 < 9; ++k9) {
                    SMma<T>::run(acc[0], wa2[half * 9 + k9], xv[0][k9]);
                    if (two) SMma<T>::run(acc[1], wa2[half * 9 + k9], xv[1][k9]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (pq[u] < npix && (u == 0 || two)) {
                    T o[4];
                    Tr<T>::st(&o[0], acc[u][0] + b42.x); Tr<T>::st(&o[1], acc[u][1] + b42.y); Tr<T>::st(&o[2], acc[u][2] + b42.z); Tr<T>::st(&o[3], acc[u][3] + b42.w);
                    *reinterpret_cast<uint2*>(yb + (size_t)pq[u] * 32 + nt * 16 + fg * 4) = *reinterpret_cast<const uint2*>(o);
                }
        }
    }
    if (p.prof && lane == 0) {
        unsigned long long* o = p.prof + ((size_t)blockIdx.x * 8 + wave) * 5;
        o[0] = st[0]; o[1] = st[1]; o[2] = st[2]; o[3] = st[3]; o[4] = s3_now();
    }
}

bool simplecnn3_ok(int dt, int H) {
    if ((dt != DT_BF16 && dt != DT_F16) || H < 36 || H % 4) return false;
    const int h1 = (H - 8) / 4 + 1, h2 = (h1 - 4) / 2 + 1, h3 = h2 - 2;
    return h3 >= 1 && s3_lds_bytes(H, h1, h2) <= 160 * 1024;
}

// y [B][h3][h3][32] = conv3(relu(conv2(relu(conv1(x))))) of SimpleDepthCNN; w1f / w2f: the 4x4 and 3x3 weights ([64][512], [32][576], k = (kh*KW + kw)*Cin + ci)
// in fragment order (launch_pack_frag)
hipError_t launch_simplecnn3(const float* x, const void* w0, const float* b0, const void* w1f, const float* b1, const void* w2f, const float* b2, void* y,
                             int dt, int B, int H, hipStream_t s) {
    if (!simplecnn3_ok(dt, H) || B < 1) return hipErrorInvalidValue;
    SimpleCnn3Dev p;
    p.x = x; p.w0 = w0; p.b0 = b0; p.w1f = w1f; p.b1 = b1; p.w2f = w2f; p.b2 = b2; p.y = y;
    p.H = H; p.h1 = (H - 8) / 4 + 1; p.h2 = (p.h1 - 4) / 2 + 1; p.h3 = p.h2 - 2; p.bands = (p.h3 + S3_N3 - 1) / S3_N3;
    const size_t lds = s3_lds_bytes(H, p.h1, p.h2);
    static const char* prof_env = dev_env("HCM_S3_PROF_PTR");
    p.prof = prof_env ? reinterpret_cast<unsigned long long*>(strtoull(prof_env, nullptr, 0)) : nullptr;
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(simplecnn3_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(simplecnn3_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_once.done();
    }
    if (dt == DT_BF16) hipLaunchKernelGGL(simplecnn3_kernel<bf16>, dim3(B * p.bands), dim3(512), lds, s, p);
    else hipLaunchKernelGGL(simplecnn3_kernel<f16>, dim3(B * p.bands), dim3(512), lds, s, p);
    return hipGetLastError();
}

bool depth_conv8x8s4_ok(int dt, int H, int act) {
    return (dt == DT_BF16 || dt == DT_F16) && H >= 8 && H % 4 == 0 && (size_t)S_ROWS * H * 2 <= 64 * 1024 && (act == ACT_NONE || act == ACT_RELU);
}

hipError_t launch_depth_conv8x8s4(const float* x, const void* w, const float* bias, void* y, int dt, int B, int H, int act, hipStream_t s) {
    if (!depth_conv8x8s4_ok(dt, H, act)) return hipErrorInvalidValue;
    const int h1 = (H - 8) / 4 + 1;
    const dim3 grid((h1 + S_RB - 1) / S_RB, B);
    const size_t lds = (size_t)S_ROWS * H * 2;
    if (dt == DT_BF16) hipLaunchKernelGGL(depth_conv8x8s4_kernel<bf16>, grid, dim3(256), lds, s, x, (const bf16*)w, bias, (bf16*)y, H, h1, act);
    else hipLaunchKernelGGL(depth_conv8x8s4_kernel<f16>, grid, dim3(256), lds, s, x, (const f16*)w, bias, (f16*)y, H, h1, act);
    return hipGetLastError();
}

}  // namespace hcm
