// RGB stem as ONE launch (round 6): torchvision resnet50's conv1 (7x7 / 2, pad 3, BatchNorm folded) + ReLU + MaxPool2d(3, 2, 1)
// -- `TorchVisionResNet50.cnn`'s first four modules, resnet_encoders.py:144-231 of the reference -- on the packed 16-bit frame of
// kernels.h: launch_pack_frame, for 256-pixel-wide frames.  Replaces {implicit-GEMM stem conv with the horizontal pool half in its epilogue
// -> vpool3s2_kernel}: 129 + 41 us and 0.4 GB of traffic per step at B = 64 (profiles/r5_seq_rgb_b64.txt), of which the conv was bound by
// its REQUESTS, not by its arithmetic: a 128 x 128 output tile re-fetched the 57 KB of weights and gathered its 128 x 7 kernel rows as
// 112 one-KB LDS-DMA pieces of 16 rows x 64 B for 448 MFMAs (profiles/r6_request_rate.md: such pieces retire at 28 per us and CU once they
// miss L2).  Here nothing is fetched twice:
//   * a workgroup (4 waves) owns a band of 16 pooled rows of ONE image for ONE 64-channel group (the hi | lo pair trunk has two) and walks
//     it top to bottom, two conv rows (= one pooled row) per step;
//   * the 64 x 224 weights of the group live in REGISTERS for the whole band (7 kernel rows x 4 channel fragments x 4 VGPRs = 112 per lane,
//     loaded once in MFMA operand layout);
//   * the packed frame streams through a 16-row LDS ring: 4 new rows (one contiguous 8448-byte block, 9 LDS-DMA pieces) per step, requested
//     a step ahead; a kernel row of an output pixel is 64 contiguous bytes of a ring row, so the B operand of `v_mfma_f32_16x16x32` is ONE
//     ds_read_b128 per lane at byte 16 (pixel + k-group) -- conflict-free, 28 reads per wave for its 112 MFMAs;
//   * bias + ReLU + one rounding in registers, the two conv rows go to an LDS image, and the pool is taken there: horizontal 3-max of both
//     rows, vertical max with the previous step's (horizontally pooled) odd row, which the same lane keeps in registers.  Values behind the ReLU are >= +0 or NaN,
//     so the maximum of the 16-bit patterns as UNSIGNED integers is the IEEE maximum (NaN patterns compare above every number: a NaN still
//     wins, as torch's max-pool and the library's other pools have it) -- v_pk_max_u16, no conversions.
// Two workgroups per CU (77 KB of LDS incl. the reduction's 8 KB of weights, <= 256 VGPRs): while one is in its epilogue / pool phase the other one's MFMAs have the matrix pipe.
// Same MFMA instruction, operand roles (weights = src0, pixels = src1), k order (kernel rows 0..6) and epilogue operations as the
// implicit-GEMM form, max is exact: the pooled map is BIT-IDENTICAL to the two launches it replaces (tests/test_ops_gpu.py).
#include "kernels.h"
#include "dev.h"

namespace hcm {

namespace {
typedef float sp_f32x4 __attribute__((ext_vector_type(4)));
typedef int sp_v4i __attribute__((ext_vector_type(4)));
typedef unsigned short sp_u16x2 __attribute__((ext_vector_type(2)));

constexpr int SP_W = 256;                          // frame width this kernel is built for
constexpr int SP_WO = SP_W / 2;                    // conv map width (128)
constexpr int SP_WP = SP_WO / 2;                   // pooled map width (64)
constexpr int SP_ROWB = (SP_W + 8) * 8;            // bytes per packed frame row (4 channels x 2 bytes per pixel): 2112
constexpr int SP_GROUPB = 4 * SP_ROWB;             // one ring group = 4 packed rows = 8448 bytes, contiguous in HBM and in LDS
constexpr int SP_RING = 4 * SP_GROUPB;             // 16 rows: the 9 rows of a step (3 groups) + the group requested for the next step
constexpr int SP_PXB = 144;                        // bytes per pixel of the LDS row images (64 channels x 2 + 16: 16-byte aligned chunks)
constexpr int SP_IMG = SP_WO * SP_PXB;             // one full-width conv row image
constexpr int SP_W1 = 64 * 64 * 2, SP_B1 = 64 * 4;         // RED: the reduction's weights in fragment order, and its bias
constexpr int SP_LDS = SP_RING + 2 * SP_IMG + SP_W1 + SP_B1;   // 33792 + 36864 + 8192 + 256 = 79104 (two workgroups per CU)
constexpr int SP_BAND = 16;                        // pooled rows per workgroup at full batches (small batches: shorter bands, launch_rgb_stem_pool)

struct StemPoolDev {
    const char* pk;          // packed frame [B][H+6][W+8][4] T
    const char* w;           // [C][224] T, k = kh*32 + kw*4 + ci
    const float* bias;       // [C]
    char* y;                 // pooled map [B][Hp][64][C] T
    int B, H, C, Hp, nbands, ngroups, band;
    unsigned img_bytes;      // (H+6) * SP_ROWB
    // RED: layer1 block 0's 1x1 reduction (64 -> 64 per channel group, BatchNorm folded, ReLU) taken from the pooled row while it is in registers
    const char* w1;          // [C][64] T (group g: rows g*64 .., k = the group's own 64 channels)
    const float* b1;         // [C]
    char* o1;                // [B][Hp][64][C] T
};

template <typename T> struct SpMma;
template <> struct SpMma<bf16> {
    static __device__ __forceinline__ void run(sp_f32x4& acc, const uint4& a, const uint4& b) {
        typedef __bf16 v8 __attribute__((ext_vector_type(8)));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), acc, 0, 0, 0);
    }
};
template <> struct SpMma<f16> {
    static __device__ __forceinline__ void run(sp_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), acc, 0, 0, 0);
    }
};
__device__ __forceinline__ void sp_dma16(unsigned lds_addr, unsigned voff, sp_v4i rsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ uint4 sp_max(const uint4& a, const uint4& b) {
    uint4 r;
    r.x = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(sp_u16x2, a.x), __builtin_bit_cast(sp_u16x2, b.x)));
    r.y = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(sp_u16x2, a.y), __builtin_bit_cast(sp_u16x2, b.y)));
    r.z = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(sp_u16x2, a.z), __builtin_bit_cast(sp_u16x2, b.z)));
    r.w = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(sp_u16x2, a.w), __builtin_bit_cast(sp_u16x2, b.w)));
    return r;
}

template <typename T, bool RED>
__global__ __launch_bounds__(256, 2) void rgb_stem_pool_kernel(StemPoolDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    // block -> (band, channel group, image): the two channel groups of a band are neighbours in the grid (they read the same frame rows)
    int bid = blockIdx.x;
    const int cg = bid % p.ngroups; bid /= p.ngroups;
    const int band = bid % p.nbands;
    const int b = bid / p.nbands;
    const int p0 = band * p.band, p1 = min(p0 + p.band, p.Hp);
    const int rs = p0 > 0 ? p0 - 1 : 0;                       // a band below the first one recomputes the odd conv row above it (no output)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    char* img = smem + SP_RING;                               // two full-width conv rows [2][128 px][SP_PXB]
    char* w1s = img + 2 * SP_IMG;                             // RED: W1 of the channel group, [k step][channel fragment][lane] x 16 bytes (a fragment = 1 KB)
    float* b1s = reinterpret_cast<float*>(w1s + SP_W1);
    sp_v4i rsrc;
    {
        const unsigned long long a = (unsigned long long)(p.pk + (size_t)b * p.img_bytes);
        rsrc[0] = (int)(unsigned)a; rsrc[1] = (int)((unsigned)(a >> 32) & 0xFFFFu); rsrc[2] = (int)p.img_bytes; rsrc[3] = 0x00020000;
    }
    // one ring group (4 packed rows = 528 sixteen-byte chunks): pieces 0..8, piece j by wave j % 4; rows past the frame read as zeros (num_records)
    auto request_group = [&](int g) {
        const unsigned src = (unsigned)g * SP_GROUPB, dst = lds_base + (unsigned)(g & 3) * SP_GROUPB;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int piece = wave + 4 * j;
            if (piece < 8 || (piece == 8 && lane < 16)) sp_dma16(dst + piece * 1024, src + piece * 1024 + lane * 16, rsrc);
        }
    };
    // weights of the channel group in MFMA src0 layout: lane = channel (fr) x k-group (fg), 8 consecutive k per lane
    uint4 wf[7][4];
    {
        const char* wg = p.w + (size_t)cg * 64 * 224 * 2;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
                wf[ky][nf] = *reinterpret_cast<const uint4*>(wg + ((size_t)(nf * 16 + fr) * 224 + ky * 32 + fg * 8) * 2);
    }
    float bias4[4][4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + cg * 64 + nf * 16 + fg * 4);
        bias4[nf][0] = bv.x; bias4[nf][1] = bv.y; bias4[nf][2] = bv.z; bias4[nf][3] = bv.w;
    }
    if constexpr (RED) {
        const char* wg1 = p.w1 + (size_t)cg * 64 * 64 * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i, ks = idx >> 8, nf = (idx >> 6) & 3, l = idx & 63;
            *reinterpret_cast<uint4*>(w1s + idx * 16) = *reinterpret_cast<const uint4*>(wg1 + ((size_t)(nf * 16 + (l & 15)) * 64 + ks * 32 + (l >> 4) * 8) * 2);
        }
        if (tid < 64) b1s[tid] = p.b1[cg * 64 + tid];
    }
    request_group(rs); request_group(rs + 1); request_group(rs + 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint4 prevr[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};      // the previous odd conv row, horizontally pooled: this lane's two chunks
    const int crow = wave >> 1;                               // which of the step's two conv rows this wave computes
    const int px0 = (wave & 1) * 64;                          // ... and which half of it
    for (int r = rs; r < p1; ++r) {
        request_group(r + 3);                                 // next step's new rows, into the slot step r - 1 has released (barriers below)
        sp_f32x4 acc[4][4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) acc[nf][mf] = (sp_f32x4){0.f, 0.f, 0.f, 0.f};
        const int y = 2 * r + crow;
        // (the next kernel row's four fragments are read while this one's sixteen MFMAs issue: two register sets, by hand)
        uint4 xf[2][4];
        {
            const char* rowp = smem + ((2 * y) & 15) * SP_ROWB + (px0 + fr + fg) * 16;
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) xf[0][mf] = *reinterpret_cast<const uint4*>(rowp + mf * 256);
        }
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            if (ky + 1 < 7) {
                const char* rowp = smem + ((2 * y + ky + 1) & 15) * SP_ROWB + (px0 + fr + fg) * 16;
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) xf[(ky + 1) & 1][mf] = *reinterpret_cast<const uint4*>(rowp + mf * 256);
            }
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) SpMma<T>::run(acc[nf][mf], wf[ky][nf], xf[ky & 1][mf]);
        }
        // (every wave is done with the previous step's pool phase: the row images may be overwritten)
        __syncthreads();
        {
            char* irow = img + crow * SP_IMG;
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) {
                    T o4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) Tr<T>::st(&o4[e], relu_f(acc[nf][mf][e] + bias4[nf][e]));
                    *reinterpret_cast<uint2*>(irow + (px0 + mf * 16 + fr) * SP_PXB + nf * 32 + fg * 8) = *reinterpret_cast<const uint2*>(o4);
                }
        }
        // The group requested at the top of the step has had the MFMA phase and the epilogue to land; the barrier makes every wave's pieces
        // everybody's.  vmcnt counts stores as well, and this is the ONE place in a step where waiting for "everything" costs nothing: the only
        // other operations in flight are the previous step's pooled / reduced stores, a whole step old.  (First version: the wait stood in front of
        // the step's first barrier, one MFMA phase behind those stores -- every step stalled on their acknowledgement: 97 us per launch with the
        // reduction's second store stream against 60 without.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // pool: 64 pooled pixels x 8 sixteen-byte channel chunks = 512 items, two per thread: wave w owns pooled pixels 16 w + fr, a lane the chunks
        // fg and 4 + fg of its pixel -- i.e. the pooled row comes out in the layout of an MFMA src1 operand (pixel = lane & 15, k-group = lane >> 4,
        // 8 consecutive channels), which is what the reduction below multiplies
        const int q = wave * 16 + fr;
        const int xl = q ? 2 * q - 1 : 0;                     // left tap clamped onto the centre at the map's edge (vpool3s2 / hpool do the same)
        uint4 pooled[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = (ks * 4 + fg) * 16;
            const char* r0 = img, *r1 = img + SP_IMG;
            uint4 h0 = sp_max(sp_max(*reinterpret_cast<const uint4*>(r0 + xl * SP_PXB + ch), *reinterpret_cast<const uint4*>(r0 + 2 * q * SP_PXB + ch)),
                              *reinterpret_cast<const uint4*>(r0 + (2 * q + 1) * SP_PXB + ch));
            uint4 h1 = sp_max(sp_max(*reinterpret_cast<const uint4*>(r1 + xl * SP_PXB + ch), *reinterpret_cast<const uint4*>(r1 + 2 * q * SP_PXB + ch)),
                              *reinterpret_cast<const uint4*>(r1 + (2 * q + 1) * SP_PXB + ch));
            uint4 o = sp_max(h0, h1);
            if (r > 0) o = sp_max(o, prevr[ks]);              // conv row 2r - 1; above the map's first row the window is clamped
            prevr[ks] = h1;
            pooled[ks] = o;
            if (r >= p0)
                *reinterpret_cast<uint4*>(p.y + ((((size_t)b * p.Hp + r) * SP_WP + q) * p.C + cg * 64) * 2 + ch) = o;
        }
        if constexpr (RED) {
            if (r >= p0) {
                sp_f32x4 ra[4];
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) ra[nf] = (sp_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) SpMma<T>::run(ra[nf], *reinterpret_cast<const uint4*>(w1s + ((ks * 4 + nf) * 64 + lane) * 16), pooled[ks]);
                char* orow = p.o1 + ((((size_t)b * p.Hp + r) * SP_WP + q) * p.C + cg * 64) * 2;
#pragma unroll
                for (int np = 0; np < 2; ++np) {
                    // channel blocks 2 np and 2 np + 1 -> 8 consecutive channels per lane (dev.h: swap_pair), one 16-byte store
                    float v[8];
                    swap_pair(ra[2 * np], ra[2 * np + 1], v);
                    const int c0 = np * 32 + (fg & 1) * 16 + (fg >> 1) * 8;
                    const float4 ba = *reinterpret_cast<const float4*>(b1s + c0), bb = *reinterpret_cast<const float4*>(b1s + c0 + 4);
                    v[0] = relu_f(v[0] + ba.x); v[1] = relu_f(v[1] + ba.y); v[2] = relu_f(v[2] + ba.z); v[3] = relu_f(v[3] + ba.w);
                    v[4] = relu_f(v[4] + bb.x); v[5] = relu_f(v[5] + bb.y); v[6] = relu_f(v[6] + bb.z); v[7] = relu_f(v[7] + bb.w);
                    *reinterpret_cast<uint4*>(orow + c0 * 2) = pack_chunk<T>(v);
                }
            }
        }
    }
}
DeviceOnce g_stem_attr;
}  // namespace

bool rgb_stem_pool_ok(int dt, int H, int W, int C, int Kp) {
    return (dt == DT_F16 || dt == DT_BF16) && W == SP_W && H >= 8 && (H % 4) == 0 && (C % 64) == 0 && Kp == 224;
}

hipError_t launch_rgb_stem_pool(const void* pk, const void* w, const float* bias, void* y, int dt, int B, int H, int W, int C, hipStream_t s,
                                const void* w1, const float* b1, void* o1) {
    if (!rgb_stem_pool_ok(dt, H, W, C, 224)) return hipErrorInvalidValue;
    StemPoolDev p;
    p.pk = (const char*)pk; p.w = (const char*)w; p.bias = bias; p.y = (char*)y;
    p.B = B; p.H = H; p.C = C; p.Hp = H / 4; p.ngroups = C / 64;
    // band height: 16 pooled rows (1 / 16 of the conv recomputed at a band's top) when that already gives two workgroups per CU; small batches take
    // shorter bands down to 2 rows (B = 1: 64 workgroups of three steps instead of 8 of seventeen -- latency, not throughput, is what a one-environment step pays)
    p.band = SP_BAND;
    while (p.band > 2 && (long)B * p.ngroups * ((p.Hp + p.band - 1) / p.band) < 512) p.band >>= 1;
    p.nbands = (p.Hp + p.band - 1) / p.band;
    p.img_bytes = (unsigned)(H + 6) * SP_ROWB;
    p.w1 = (const char*)w1; p.b1 = b1; p.o1 = (char*)o1;
    const bool red = w1 && b1 && o1;
    if (g_stem_attr.need()) {
        const void* fns[4] = {reinterpret_cast<const void*>(rgb_stem_pool_kernel<f16, false>), reinterpret_cast<const void*>(rgb_stem_pool_kernel<bf16, false>),
                              reinterpret_cast<const void*>(rgb_stem_pool_kernel<f16, true>), reinterpret_cast<const void*>(rgb_stem_pool_kernel<bf16, true>)};
        for (const void* fn : fns) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS);
            if (e != hipSuccess) return e;
        }
        g_stem_attr.done();
    }
    const dim3 grid((unsigned)(B * p.nbands * p.ngroups));
    if (dt == DT_BF16) {
        if (red) hipLaunchKernelGGL((rgb_stem_pool_kernel<bf16, true>), grid, dim3(256), SP_LDS, s, p);
        else hipLaunchKernelGGL((rgb_stem_pool_kernel<bf16, false>), grid, dim3(256), SP_LDS, s, p);
    } else {
        if (red) hipLaunchKernelGGL((rgb_stem_pool_kernel<f16, true>), grid, dim3(256), SP_LDS, s, p);
        else hipLaunchKernelGGL((rgb_stem_pool_kernel<f16, false>), grid, dim3(256), SP_LDS, s, p);
    }
    return hipGetLastError();
}

}  // namespace hcm
