// Micro-benchmark: per-CU read rate from L2 (buffer resident in every XCD's L2) for (0) LDS-DMA `buffer_load_dwordx4 ... lds`,
// (1) plain `global_load_dwordx4` into VGPRs, (2) `buffer_load_dwordx4` into VGPRs.  One 512-thread workgroup per CU.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/native/l2_rate.hip -o tools/native/libl2_rate.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i make_rsrc(const void* p, unsigned bytes) {
    const uint64_t a = (uint64_t)p;
    v4i r; r.x = (int)(uint32_t)a; r.y = (int)(uint32_t)(a >> 32); r.z = (int)bytes; r.w = 0x00020000; return r;
}
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, v4i rsrc) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(const char* buf, unsigned bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const v4i rs = make_rsrc(buf, bytes);
    // each iteration the workgroup reads 64 KB: wave w reads 8 x 1 KB rows; offsets walk through the buffer
    unsigned off = (blockIdx.x * 65536u) % bytes;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) dma16(lds_base + (wave * 8 + j) * 1024, off + (wave * 8 + j) * 1024 + lane * 16, rs);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (MODE == 1) {
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const uint4*>(buf + off + (wave * 8 + j) * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
        } else {
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned vo = off + (wave * 8 + j) * 1024 + lane * 16;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[j]) : "v"(vo), "s"(rs) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
        }
        off += 65536u;
        if (off + 65536u > bytes) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE == 0) acc.x = *reinterpret_cast<const unsigned*>(smem + tid * 4);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}
// GEMM-operand pattern: the workgroup walks 256-row panels of a row-major matrix (row stride `ld` bytes), 128 B per row per K step
// (one wave instruction = 8 rows x 128 B, as the LDS-DMA operand fetch of the GEMM kernels), K steps across the row, then the next panel.
template <int W>
__global__ __launch_bounds__(512) void panel_kernel(const char* buf, unsigned bytes, unsigned ld, unsigned rowbytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const v4i rs = make_rsrc(buf, bytes);
    const unsigned rows = bytes / ld, panels = rows / 256, ksteps = rowbytes / 128;
    unsigned panel = blockIdx.x % panels, k = 0;
    const unsigned rin = lane >> 3, c = lane & 7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned row = panel * 256 + (wave * 4 + j) * 8 + rin;
            dma16(lds_base + (wave * 4 + j) * 1024, row * ld + k * 128 + c * 16, rs);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(W) : "memory");
        if (++k == ksteps) { k = 0; panel = (panel + 1) % panels; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (*reinterpret_cast<const unsigned*>(smem + tid * 4) == 0x12345u && lane == 99) sink[0] = 1;
}
extern "C" int run_panel(const void* buf, unsigned bytes, unsigned ld, unsigned rowbytes, int iters, int blocks, int window, void* sink, void* stream) {
#define PK(Wn) hipLaunchKernelGGL(panel_kernel<Wn>, dim3(blocks), dim3(512), 65536, (hipStream_t)stream, (const char*)buf, bytes, ld, rowbytes, iters, (unsigned*)sink)
    if (window == 0) PK(0); else if (window == 4) PK(4); else if (window == 8) PK(8); else if (window == 12) PK(12); else if (window == 20) PK(20); else PK(28);
    return (int)hipGetLastError();
}
extern "C" int run_rate(int mode, const void* buf, unsigned bytes, int iters, int blocks, void* sink, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(512), 65536, s, (const char*)buf, bytes, iters, (unsigned*)sink);
    else if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(512), 65536, s, (const char*)buf, bytes, iters, (unsigned*)sink);
    else hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(512), 65536, s, (const char*)buf, bytes, iters, (unsigned*)sink);
    return (int)hipGetLastError();
}
