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
