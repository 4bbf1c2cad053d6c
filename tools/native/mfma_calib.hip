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
