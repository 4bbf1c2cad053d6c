// Memory-pattern probe for the write-dominated expansion convs (tools/tile_stream.py): a workgroup moves the same bytes as one
// 128-pixel x 128-channel output tile of a 1x1 conv with K = 64 and N = 256 -- reads the 128x64 activation tile, reads the
// residual tile, writes the output tile (256-byte runs at 512-byte stride) -- with no LDS, no barriers, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(512) void tile_stream(const uint4* __restrict__ x, const uint4* __restrict__ res, uint4* __restrict__ y,
                                                              int M, int Cin8, int N8, int use_res, int rows_per_block, int cols8) {
    // block -> (m tile, n tile); thread -> (row within pass, 16-byte column)
    const int tilesN = N8 / cols8;
    const int tile_m = blockIdx.x / tilesN, tile_n = blockIdx.x % tilesN;
    const int tpr = cols8;                               // threads per row (16 B each)
    const int rpp = blockDim.x / tpr;
    const int c = threadIdx.x % tpr, r0 = threadIdx.x / tpr;
    uint4 acc = make_uint4(0, 0, 0, 0);
    // activation tile: rows_per_block x Cin8 chunks, spread over the threads
    for (int i = threadIdx.x; i < rows_per_block * Cin8; i += blockDim.x) {
        const int r = i / Cin8, cc = i % Cin8;
        const size_t m = (size_t)tile_m * rows_per_block + r;
        if (m < (size_t)M) { const uint4 v = x[m * Cin8 + cc]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    }
    for (int r = r0; r < rows_per_block; r += rpp) {
        const size_t m = (size_t)tile_m * rows_per_block + r;
        if (m >= (size_t)M) continue;
        const size_t o = m * N8 + (size_t)tile_n * cols8 + c;
        uint4 v = acc;
        if (use_res) { const uint4 q = res[o]; v.x ^= q.x; v.y ^= q.y; v.z ^= q.z; v.w ^= q.w; }
        y[o] = v;
    }
}
// lds_bytes > 0: that much dynamic LDS per workgroup, to pin the number of resident workgroups per CU (160 KB / lds_bytes)
extern "C" int run_tile_stream_lds(const void* x, const void* res, void* y, int M, int Cin, int N, int use_res, int rows, int cols, int lds_bytes, void* stream) {
    const int tilesM = (M + rows - 1) / rows, tilesN = N / cols;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tile_stream), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(tile_stream, dim3(tilesM * tilesN), dim3(512), lds_bytes, (hipStream_t)stream, (const uint4*)x, (const uint4*)res, (uint4*)y, M, Cin / 8,
                       N / 8, use_res, rows, cols / 8);
    return (int)hipGetLastError();
}
extern "C" int run_tile_stream(const void* x, const void* res, void* y, int M, int Cin, int N, int use_res, int rows, int cols, void* stream) {
    const int tilesM = (M + rows - 1) / rows, tilesN = N / cols;
    hipLaunchKernelGGL(tile_stream, dim3(tilesM * tilesN), dim3(512), 0, (hipStream_t)stream, (const uint4*)x, (const uint4*)res, (uint4*)y, M, Cin / 8,
                       N / 8, use_res, rows, cols / 8);
    return (int)hipGetLastError();
}
