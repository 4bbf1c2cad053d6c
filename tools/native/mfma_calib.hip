// Calibration of the matrix-pipe counters and of the sustained MFMA issue rate: a kernel that is NOTHING but back-to-back
// v_mfma_f32_16x16x32_f16 on independent accumulators (operands from registers, random bits), NW waves per workgroup (4 = one wave
// per SIMD, 8 = two), one workgroup per CU.  Known work: iters * 32 MFMAs per wave -> SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS
// and SQ_BUSY_CYCLES can be read against an exact count (tools/mfma_calib.py; profiles/r4_mfma_counter_calibration.md).
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/native/mfma_calib.hip -o tools/native/libmfma_calib.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NW>
__global__ __launch_bounds__(64 * NW) void mfma_only(const uint4* seed, float* sink, int iters) {
    const int tid = threadIdx.x;
    uint4 a0 = seed[tid & 63], b0 = seed[64 + (tid & 63)];
    half8 a[4], b[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { uint4 t = a0; t.x ^= i * 0x01010101u; a[i] = __builtin_bit_cast(half8, t); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { uint4 t = b0; t.y ^= j * 0x00010001u; b[j] = __builtin_bit_cast(half8, t); }
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) sink[0] = s;
}

extern "C" int run_mfma_only(int waves, const void* seed, void* sink, int iters, int blocks, void* stream) {
    if (waves == 4) hipLaunchKernelGGL(mfma_only<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)seed, (float*)sink, iters);
    else hipLaunchKernelGGL(mfma_only<8>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const uint4*)seed, (float*)sink, iters);
    return (int)hipGetLastError();
}

// Round 6 (review item 2, last sentence): the K loop of gemm256f_kernel on the OTHER matrix instruction.  A wave's 128-token x 64-channel tile per
// 32-deep K step is 12 fragment reads (ds_read_b128, 1 KB each) + 32 x v_mfma_f32_16x16x32_f16 -- or the same 12 reads + 16 x v_mfma_f32_32x32x16_f16
// (4 x 2 blocks of 32 x 32, two 16-deep halves).  Same bytes, same FLOPs, 8 waves per CU, fragments from a 64 KB LDS image, two register sets
// (the next step's fragments are read while this step multiplies), no DMA and no barrier: what the instruction shape alone is worth in that loop.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3c003c00u + i, 0x38003800u, 0x34003400u + lane, 0x3c003c00u);
    __syncthreads();
    uint4 f[2][12];
    auto rd = [&](int step, uint4 (&d)[12]) {
#pragma unroll
        for (int j = 0; j < 12; ++j) d[j] = *reinterpret_cast<const uint4*>(smem + ((step * 12 + j + wave * 5) & 63) * 1024 + lane * 16);
    };
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rd(0, f[0]);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                rd(2 * it + h + 1, f[h ^ 1]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, f[h][i]), __builtin_bit_cast(half8, f[h][4 + j]), acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        rd(0, f[0]);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                rd(2 * it + h + 1, f[h ^ 1]);
                // fragments 0-3: the two channel blocks x two K halves, 4-11: the four token blocks x two K halves
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, f[h][i * 2 + kh]), __builtin_bit_cast(half8, f[h][4 + j * 2 + kh]), acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    if (s == 12345.678f) sink[0] = s;
}
extern "C" int run_mfma_loop(int shape, void* sink, int iters, int blocks, void* stream) {
    if (shape == 16) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_loop<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
                       hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(512), 65536, (hipStream_t)stream, (float*)sink, iters); }
    else { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_loop<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
           hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(512), 65536, (hipStream_t)stream, (float*)sink, iters); }
    return (int)hipGetLastError();
}
