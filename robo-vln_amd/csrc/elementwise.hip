// Memory-bound kernels of the HCM step: frame packing, pooling, GroupNorm, LayerNorm, BERT
// embeddings, recurrent cells + heads, small glue.  All are HBM/L2-bound: 16-byte vector accesses,
// grid-stride loops, wave64 reductions.
#include <cstdlib>
#include "kernels.h"
#include "dev.h"

namespace hcm {

// run `...` with T bound to the storage type named by dt
#define HCM_DISPATCH_T(dt, ...)                                           \
    do {                                                                  \
        if ((dt) == DT_BF16) { using T = bf16; __VA_ARGS__; }             \
        else if ((dt) == DT_F16) { using T = f16; __VA_ARGS__; }          \
        else if ((dt) == DT_F32) { using T = float; __VA_ARGS__; }        \
        else return hipErrorInvalidValue;                                 \
    } while (0)

static inline int grid_for(size_t n, int block = 256, int cap = 256 * 16) {
    size_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (size_t)cap) g = cap;
    return (int)g;
}

// ------------------------------------------------------------------------------------------ packed RGB frame (stem)
// see kernels.h: launch_pack_frame.  One thread per packed pixel (8-byte store).
template <typename S, typename T>
__global__ void pack_frame_kernel(const S* __restrict__ x, T* __restrict__ y, int B, int H, int W, float scale, int bt, int Hp, int Wp) {
    const size_t total = (size_t)B * Hp * Wp;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int px = (int)(e % Wp);
        const int py = (int)((e / Wp) % Hp);
        const int b = (int)(e / ((size_t)Wp * Hp));
        const int ix = px - bt, iy = py - bt;            // bt: border on the top and left
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if ((unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H) {
            const S* p = x + ((size_t)(b * H + iy) * W + ix) * 3;
            v0 = (float)p[0] * scale; v1 = (float)p[1] * scale; v2 = (float)p[2] * scale;
        }
        T o[4];
        Tr<T>::st(&o[0], v0); Tr<T>::st(&o[1], v1); Tr<T>::st(&o[2], v2); Tr<T>::st(&o[3], 0.f);
        *reinterpret_cast<uint2*>(y + e * 4) = *reinterpret_cast<const uint2*>(o);
    }
}
// uint8 frames with the stem's border (round 5): FOUR packed pixels per thread.  Output pixels px0 .. px0 + 3 (px0 a multiple of 4) are input pixels
// px0 - 3 .. px0, i.e. bytes [3 px0 - 9, 3 px0 + 3) of the row: inside the 16-byte window that starts at the 4-byte-aligned byte 3 px0 - 12, read as four
// dwords (the row is 3 W bytes, W % 4 == 0: a dword is inside the row or outside it), written as two 16-byte stores.  The one-pixel-per-thread form
// issued three 1-byte loads and one 8-byte store per pixel and ran at 2.6 TB/s.  Same conversion ((float)byte * scale, one rounding): the same bits.
template <typename T>
__global__ __launch_bounds__(256) void pack_frame_u8x4_kernel(const uint8_t* __restrict__ x, T* __restrict__ y, int B, int H, int W, float scale, int Hp, int Wp) {
    const int Wq = Wp >> 2;
    const size_t total = (size_t)B * Hp * Wq;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(e % Wq);
        const int py = (int)((e / Wq) % Hp);
        const int b = (int)(e / ((size_t)Wq * Hp));
        const int px0 = q * 4, iy = py - 3;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        const bool row_ok = (unsigned)iy < (unsigned)H;
        if (row_ok) {
            const uint8_t* row = x + (size_t)(b * H + iy) * W * 3;
            const int a0 = 3 * px0 - 12;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int a = a0 + 4 * d;
                if (a >= 0 && a + 4 <= 3 * W) w[d] = *reinterpret_cast<const uint32_t*>(row + a);
            }
        }
        T o[16];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ix = px0 - 3 + t;
            const bool ok = row_ok && (unsigned)ix < (unsigned)W;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int byte = 3 * t + 3 + c;
                const float v = ok ? (float)((w[byte >> 2] >> ((byte & 3) * 8)) & 0xFFu) * scale : 0.f;
                Tr<T>::st(&o[t * 4 + c], v);
            }
            Tr<T>::st(&o[t * 4 + 3], 0.f);
        }
        T* yo = y + (((size_t)(b * Hp + py)) * Wp + px0) * 4;
        *reinterpret_cast<uint4*>(yo) = *reinterpret_cast<const uint4*>(o);
        *reinterpret_cast<uint4*>(yo + 8) = *reinterpret_cast<const uint4*>(o + 8);
    }
}
hipError_t launch_pack_frame(const void* x, int src_dt, void* y, int dt, int B, int H, int W, float scale, hipStream_t s, int border) {
    if ((W & 1) || (dt != DT_BF16 && dt != DT_F16)) return hipErrorInvalidValue;
    const int bt = border ? 3 : 0, Hp = border ? H + 6 : H, Wp = border ? W + 8 : W;
    const size_t total = (size_t)B * Hp * Wp;
    if (src_dt == DT_F32) {
        if (dt == DT_BF16) hipLaunchKernelGGL((pack_frame_kernel<float, bf16>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)x, (bf16*)y, B, H, W, scale, bt, Hp, Wp);
        else hipLaunchKernelGGL((pack_frame_kernel<float, f16>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)x, (f16*)y, B, H, W, scale, bt, Hp, Wp);
    } else if (src_dt == DT_U8 && border && (W & 3) == 0 && ((size_t)x & 3) == 0) {
        const size_t tot4 = (size_t)B * Hp * (Wp >> 2);
        if (dt == DT_BF16) hipLaunchKernelGGL((pack_frame_u8x4_kernel<bf16>), dim3(grid_for(tot4)), dim3(256), 0, s, (const uint8_t*)x, (bf16*)y, B, H, W, scale, Hp, Wp);
        else hipLaunchKernelGGL((pack_frame_u8x4_kernel<f16>), dim3(grid_for(tot4)), dim3(256), 0, s, (const uint8_t*)x, (f16*)y, B, H, W, scale, Hp, Wp);
    } else if (src_dt == DT_U8) {
        if (dt == DT_BF16) hipLaunchKernelGGL((pack_frame_kernel<uint8_t, bf16>), dim3(grid_for(total)), dim3(256), 0, s, (const uint8_t*)x, (bf16*)y, B, H, W, scale, bt, Hp, Wp);
        else hipLaunchKernelGGL((pack_frame_kernel<uint8_t, f16>), dim3(grid_for(total)), dim3(256), 0, s, (const uint8_t*)x, (f16*)y, B, H, W, scale, bt, Hp, Wp);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ depth avg_pool2d(2)
// habitat ResNetEncoder.forward: F.avg_pool2d(x, 2) on the (B,1,H,W) depth frame.
template <typename T>
__global__ void avgpool2_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)B * Ho * Wo;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(e % Wo);
        const int oy = (int)((e / Wo) % Ho);
        const int b = (int)(e / ((size_t)Wo * Ho));
        const float* p = x + ((size_t)(b * H + 2 * oy) * W + 2 * ox);
        const float2 r0 = *reinterpret_cast<const float2*>(p);
        const float2 r1 = *reinterpret_cast<const float2*>(p + W);
        Tr<T>::st(y + e, ((r0.x + r0.y) + (r1.x + r1.y)) * 0.25f);
    }
}
// same, written into a zero-bordered frame [B][H/2 + 6][W/2 + 8] (3 px border left / top, 5 right, 3 bottom): the packed input
// of the depth trunk's 7x7/2 stem (kernels.h: launch_pack_frame describes the RGB counterpart)
template <typename T>
__global__ void avgpool2_padded_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int H, int W) {
    const int Ho = H / 2, Wo = W / 2, Hp = Ho + 6, Wp = Wo + 8;
    const size_t total = (size_t)B * Hp * Wp;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int px = (int)(e % Wp) - 3;
        const int py = (int)((e / Wp) % Hp) - 3;
        const int b = (int)(e / ((size_t)Wp * Hp));
        float v = 0.f;
        if ((unsigned)px < (unsigned)Wo && (unsigned)py < (unsigned)Ho) {
            const float* p = x + ((size_t)(b * H + 2 * py) * W + 2 * px);
            const float2 r0 = *reinterpret_cast<const float2*>(p);
            const float2 r1 = *reinterpret_cast<const float2*>(p + W);
            v = ((r0.x + r0.y) + (r1.x + r1.y)) * 0.25f;
        }
        Tr<T>::st(y + e, v);
    }
}
hipError_t launch_avgpool2_f32_padded(const float* x, void* y, int dt, int B, int H, int W, hipStream_t s) {
    const size_t total = (size_t)B * (H / 2 + 6) * (W / 2 + 8);
    if ((W & 3) || (dt != DT_BF16 && dt != DT_F16)) return hipErrorInvalidValue;
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(avgpool2_padded_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, x, (T*)y, B, H, W));
    return hipGetLastError();
}
hipError_t launch_avgpool2_f32(const float* x, void* y, int dt, int B, int H, int W, hipStream_t s) {
    const size_t total = (size_t)B * (H / 2) * (W / 2);
    if (W & 1) return hipErrorInvalidValue;
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(avgpool2_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, x, (T*)y, B, H, W));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ MaxPool2d(3, 2, 1) NHWC
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo) {
    constexpr int CH = Tr<T>::CH;
    const int cv = C / CH;
    const size_t total = (size_t)B * Ho * Wo * cv;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % cv) * CH;
        const size_t pix = e / cv;
        const int ox = (int)(pix % Wo);
        const int oy = (int)((pix / Wo) % Ho);
        const int b = (int)(pix / ((size_t)Wo * Ho));
        // branch-free: an out-of-range tap is clamped onto the nearest valid one, which lies inside the same window (pad 1, stride 2),
        // so the maximum is unchanged -- and all nine 16-byte loads are in flight together
        int iy[3], ix[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            iy[d] = min(max(2 * oy - 1 + d, 0), H - 1);
            ix[d] = min(max(2 * ox - 1 + d, 0), W - 1);
        }
        float v[9][CH];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) ld_chunk(x + ((size_t)(b * H + iy[dy]) * W + ix[dx]) * C + c, v[dy * 3 + dx]);
        float m[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            m[j] = v[0][j];
#pragma unroll
            for (int t = 1; t < 9; ++t) m[j] = max_nan(m[j], v[t][j]);
        }
        st_chunk(y + pix * C + c, m);
    }
}
hipError_t launch_maxpool3x3s2(const void* x, void* y, int dt, int B, int H, int W, int C, int Ho, int Wo, hipStream_t s) {
    const int CH = dt_chunk(dt);
    if (C % CH) return hipErrorInvalidValue;
    const size_t total = (size_t)B * Ho * Wo * (C / CH);
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(maxpool_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)x, (T*)y, B, H, W, C, Ho, Wo));
    return hipGetLastError();
}

// MaxPool2d(3, 2, 1) over relu(GroupNorm(x)) with the normalisation applied ON LOAD (round 5: the depth trunk's stem -- conv -> GroupNorm -> ReLU ->
// max-pool, habitat's ResNet stem as used at resnet_encoders.py:27-33).  x is the UN-normalised conv output, `part` the (sum, sum of squares) partials its
// epilogue left per (sample, 64-pixel block, group) -- what gn_apply_kernel would have consumed: the stand-alone apply pass (one read and one write of the
// largest map of the trunk) and its launch disappear.  The per-channel scale / shift and the expression v * sc - sh are gn_apply_kernel's; rounding is
// monotone and ReLU commutes with it, so max(round(relu(.))) == round(max(relu(.))): BIT-IDENTICAL to the apply pass followed by maxpool_kernel.
// grid (blocks per sample, B): a workgroup rebuilds its sample's scale / shift table (G <= 256 groups, C <= 512 channels) and pools a slice of its pixels.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_gn_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ part, int PS, float eps, int G, int H, int W, int C, int Ho, int Wo) {
    constexpr int CH = Tr<T>::CH;
    __shared__ float s_scale[512], s_shift[512];
    __shared__ float s_mean[256], s_rstd[256];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int Cg = C / G, HW = H * W;
    for (int g = tid; g < G; g += 256) {
        // (the partials are added in index order -- gn_apply_kernel's order -- but REQUESTED 16 at a time: the stem map has 256 of them per group, and
        //  four per round trip made this prologue 64 dependent round trips, ~40 us of a 44 us launch)
        float a = 0.f, q = 0.f;
        const float2* pp = reinterpret_cast<const float2*>(part) + (size_t)b * PS * G + g;
        int i = 0;
        for (; i + 16 <= PS; i += 16) {
            float2 t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = pp[(size_t)(i + u) * G];
#pragma unroll
            for (int u = 0; u < 16; ++u) { a += t[u].x; q += t[u].y; }
        }
        for (; i < PS; ++i) { const float2 t = pp[(size_t)i * G]; a += t.x; q += t.y; }
        const float inv_n = 1.0f / ((float)HW * (float)Cg);
        const float mean = a * inv_n;
        const float var = relu_f(q * inv_n - mean * mean);
        s_mean[g] = mean;
        s_rstd[g] = rsqrtf(var + eps);
    }
    __syncthreads();
    for (int ch = tid; ch < C; ch += 256) {
        const int g = ch / Cg;
        const float sc = s_rstd[g] * gamma[ch];
        s_scale[ch] = sc;
        s_shift[ch] = s_mean[g] * sc - beta[ch];
    }
    __syncthreads();
    const int cv = C / CH;
    const int total = Ho * Wo * cv;
    const T* xb = x + (size_t)b * HW * C;
    T* yb = y + (size_t)b * Ho * Wo * C;
    for (int e = blockIdx.x * 256 + tid; e < total; e += gridDim.x * 256) {
        const int c = (e % cv) * CH;
        const int pix = e / cv;
        const int ox = pix % Wo, oy = pix / Wo;
        int iy[3], ix[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            iy[d] = min(max(2 * oy - 1 + d, 0), H - 1);
            ix[d] = min(max(2 * ox - 1 + d, 0), W - 1);
        }
        float v[9][CH];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) ld_chunk(xb + ((size_t)iy[dy] * W + ix[dx]) * C + c, v[dy * 3 + dx]);
        float m[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float sc = s_scale[c + j], sh = s_shift[c + j];
            m[j] = relu_f(v[0][j] * sc - sh);
#pragma unroll
            for (int t = 1; t < 9; ++t) m[j] = max_nan(m[j], relu_f(v[t][j] * sc - sh));
        }
        st_chunk(yb + (size_t)pix * C + c, m);
    }
}
bool maxpool_gn_ok(int dt, int C, int G) { return dt != DT_F32 && C % dt_chunk(dt) == 0 && C <= 512 && G >= 1 && G <= 256 && C % G == 0; }
hipError_t launch_maxpool3x3s2_gn(const void* x, void* y, const float* gamma, const float* beta, const float* part, int PS, float eps, int G, int dt, int B, int H,
                                  int W, int C, int Ho, int Wo, hipStream_t s) {
    if (!maxpool_gn_ok(dt, C, G) || PS < 1 || !part || !gamma || !beta) return hipErrorInvalidValue;
    const int total = Ho * Wo * (C / dt_chunk(dt));
    int bps = (total + 256 * 16 - 1) / (256 * 16);        // ~16 output chunks per thread: every workgroup rebuilds its sample's scale / shift table
    if (bps < 1) bps = 1;
    if (bps > 16) bps = 16;
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(maxpool_gn_kernel<T>, dim3(bps, B), dim3(256), 0, s, (const T*)x, (T*)y, gamma, beta, part, PS, eps, G, H, W, C, Ho, Wo));
    return hipGetLastError();
}

// vertical half of MaxPool2d(3, 2, 1) (the horizontal half is fused into the stem conv's epilogue, igemm.hip)
template <typename T>
__global__ void vpool3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int Ho) {
    constexpr int CH = Tr<T>::CH;
    const int cv = C / CH;
    const size_t rowv = (size_t)W * cv;                 // 16-byte vectors per row
    const size_t total = (size_t)B * Ho * rowv;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t in_row = e % rowv;
        const size_t rowi = e / rowv;
        const int oy = (int)(rowi % Ho);
        const int b = (int)(rowi / Ho);
        float v[3][CH];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int iy = min(max(2 * oy - 1 + d, 0), H - 1);      // clamped taps stay inside the window
            ld_chunk(x + ((size_t)(b * H + iy) * rowv + in_row) * CH, v[d]);
        }
        float m[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) m[j] = max_nan(max_nan(v[0][j], v[1][j]), v[2][j]);
        st_chunk(y + e * CH, m);
    }
}
hipError_t launch_vpool3s2(const void* x, void* y, int dt, int B, int H, int W, int C, hipStream_t s) {
    const int CH = dt_chunk(dt);
    if (C % CH) return hipErrorInvalidValue;
    const int Ho = (H + 2 - 3) / 2 + 1;
    const size_t total = (size_t)B * Ho * W * (C / CH);
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(vpool3s2_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)x, (T*)y, B, H, W, C, Ho));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ adaptive_avg_pool2d NHWC
// F.adaptive_avg_pool2d(x,(OH,OW)) (resnet_encoders.py:160-166) / AdaptiveAvgPool2d(1): window [floor(i*H/OH), ceil((i+1)*H/OH)).
template <typename T>
__global__ void adaptive_pool_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int OH, int OW, int ldy, int ldx) {
    constexpr int CH = Tr<T>::CH;
    const int cv = C / CH;
    const size_t total = (size_t)B * OH * OW * cv;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % cv) * CH;
        const size_t pix = e / cv;
        const int ox = (int)(pix % OW);
        const int oy = (int)((pix / OW) % OH);
        const int b = (int)(pix / ((size_t)OW * OH));
        const int y0 = (oy * H) / OH, y1 = ((oy + 1) * H + OH - 1) / OH;
        const int x0 = (ox * W) / OW, x1 = ((ox + 1) * W + OW - 1) / OW;
        float acc[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[j] = 0.f;
        for (int iy = y0; iy < y1; ++iy) {
#pragma unroll 4
            for (int ix = x0; ix < x1; ++ix) {
                float v[CH];
                ld_chunk(x + ((size_t)(b * H + iy) * W + ix) * ldx + c, v);
#pragma unroll
                for (int j = 0; j < CH; ++j) acc[j] += v[j];
            }
        }
        const float inv = 1.0f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[j] *= inv;
        st_chunk(y + pix * ldy + c, acc);
    }
}
hipError_t launch_adaptive_avgpool(const void* x, void* y, int dt, int B, int H, int W, int C, int OH, int OW, int ldy, hipStream_t s, int ldx) {
    const int CH = dt_chunk(dt);
    if (ldx <= 0) ldx = C;
    if (C % CH || ldy % CH || ldx % CH) return hipErrorInvalidValue;
    const size_t total = (size_t)B * OH * OW * (C / CH);
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(adaptive_pool_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)x, (T*)y, B, H, W, C, OH, OW, ldy, ldx));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ mean over rows
// AdaptiveAvgPool1d(1) over tokens: rgb_linear.0 (seq2seq_highlevel_cma.py:83-85), cross_pooler (:114-115,:209-210).
template <typename T>
__global__ void mean_rows_kernel(const T* __restrict__ x, void* __restrict__ y, int B, int S0, int C, int ldx, int ldy, int out_f32,
                                 const int* __restrict__ lens) {
    const size_t total = (size_t)B * C;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const int b = (int)(e / C);
        const T* p = x + (size_t)b * S0 * ldx + c;
        int S = S0;
        if (lens) { S = lens[b]; S = S < 1 ? 1 : S > S0 ? S0 : S; }
        // four independent partial sums and an unrolled body keep several loads in flight (a single dependent chain
        // of S strided 2-byte loads costs S x L2 latency: 24 us for the 80-token cross_pooler)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int s_ = 0;
#pragma unroll 2
        for (; s_ + 4 <= S; s_ += 4) {
            a0 += Tr<T>::ld(p + (size_t)s_ * ldx);
            a1 += Tr<T>::ld(p + (size_t)(s_ + 1) * ldx);
            a2 += Tr<T>::ld(p + (size_t)(s_ + 2) * ldx);
            a3 += Tr<T>::ld(p + (size_t)(s_ + 3) * ldx);
        }
        for (; s_ < S; ++s_) a0 += Tr<T>::ld(p + (size_t)s_ * ldx);
        float acc = ((a0 + a1) + (a2 + a3)) / (float)S;
        if (out_f32) reinterpret_cast<float*>(y)[(size_t)b * ldy + c] = acc;
        else Tr<T>::st(reinterpret_cast<T*>(y) + (size_t)b * ldy + c, acc);
    }
}
hipError_t launch_mean_rows(const void* x, void* y, int dt, int B, int S, int C, int ldx, int ldy, int out_f32, hipStream_t s, const int* lens) {
    const size_t total = (size_t)B * C;
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(mean_rows_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)x, y, B, S, C, ldx, ldy, out_f32, lens));
    return hipGetLastError();
}

template <typename T>
__global__ void fill_cols_kernel(const float* __restrict__ tab, T* __restrict__ y, int B, int S, int C, int ldy) {
    const size_t total = (size_t)B * S * C;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const size_t row = e / C;            // b*S + s
        const int s_ = (int)(row % S);
        Tr<T>::st(y + row * ldy + c, tab[s_ * C + c]);
    }
}
hipError_t launch_fill_cols(const float* tab, void* y, int dt, int B, int S, int C, int ldy, hipStream_t s) {
    const size_t total = (size_t)B * S * C;
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(fill_cols_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, tab, (T*)y, B, S, C, ldy));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ GroupNorm (NHWC, in place)
// nn.GroupNorm(G, C) after every conv of the habitat ResNet (+ residual + ReLU of the bottleneck), ONE launch:
// a workgroup owns (sample b, slab of CS channels made of whole groups): pass 1 accumulates per-channel sum / sum of
// squares in registers (a thread always sees the same 16-byte channel chunk), wave-shuffle + LDS reduce to per-group
// mean / rstd in f32; pass 2 re-reads the slab (L2-resident: HW*CS*2 B <= 64 KB for every layer of the trunk),
// normalises, adds the residual, applies ReLU and stores.  No atomics, no memset, no stats buffer.
template <typename T>
__global__ __launch_bounds__(256) void gn_fused_kernel(T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int HW, int C, int G, int CS, float eps, int relu, int cg_true) {
    constexpr int CH = Tr<T>::CH;
    __shared__ float s_p1[4][1024], s_p2[4][1024];                         // per-wave partials (fixed summation order)
    __shared__ float s_sum[1024], s_sq[1024], s_mean[1024], s_rstd[1024];  // per channel of the slab (CS <= 1024)
    const int slabs = C / CS;
    const int b = blockIdx.x / slabs;
    const int c_base = (blockIdx.x % slabs) * CS;
    const int Cg = C / G;
    const int cpr = CS / CH;                 // 16-byte chunks per pixel inside the slab (power of two, <= 32)
    const int tid = threadIdx.x;
    const int cc = tid % cpr;                // this thread's chunk column (fixed)
    const int prow = tid / cpr;              // first pixel
    const int pstep = 256 / cpr;
    T* xb = x + (size_t)b * HW * C + c_base + cc * CH;
    const T* rb = res ? res + (size_t)b * HW * C + c_base + cc * CH : nullptr;
    for (int i = tid; i < 4 * 1024; i += 256) { (&s_p1[0][0])[i] = 0.f; (&s_p2[0][0])[i] = 0.f; }
    __syncthreads();
    float s1[CH], s2[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    for (int p = prow; p < HW; p += pstep) {
        float v[CH];
        ld_chunk(xb + (size_t)p * C, v);
#pragma unroll
        for (int j = 0; j < CH; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
    }
    // lanes with equal (lane % cpr) hold the same channels: butterfly over the other lane bits
    for (int o = 32; o >= cpr; o >>= 1) {
#pragma unroll
        for (int j = 0; j < CH; ++j) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
    }
    if ((tid & 63) < cpr) {
        const int wv = tid >> 6;
#pragma unroll
        for (int j = 0; j < CH; ++j) { s_p1[wv][cc * CH + j] = s1[j]; s_p2[wv][cc * CH + j] = s2[j]; }
    }
    __syncthreads();
    for (int ch = tid; ch < CS; ch += 256) {
        s_sum[ch] = (s_p1[0][ch] + s_p1[1][ch]) + (s_p1[2][ch] + s_p1[3][ch]);
        s_sq[ch] = (s_p2[0][ch] + s_p2[1][ch]) + (s_p2[2][ch] + s_p2[3][ch]);
    }
    __syncthreads();
    for (int ch = tid; ch < CS; ch += 256) {
        const int g0 = (ch / Cg) * Cg;       // first channel of this channel's group (inside the slab)
        float a = 0.f, q = 0.f;
        for (int j = 0; j < Cg; ++j) { a += s_sum[g0 + j]; q += s_sq[g0 + j]; }
        const float inv_n = 1.0f / ((float)HW * (float)(cg_true > 0 ? cg_true : Cg));     // cg_true: the group's other channels are zero padding
        const float mean = a * inv_n;
        const float var = relu_f(q * inv_n - mean * mean);
        const float rstd = rsqrtf(var + eps);
        const float ga = gamma[c_base + ch];
        s_mean[ch] = mean * rstd * ga - beta[c_base + ch];     // y = x*scale - shift'
        s_rstd[ch] = rstd * ga;
    }
    __syncthreads();
    float sc[CH], sh[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { sc[j] = s_rstd[cc * CH + j]; sh[j] = s_mean[cc * CH + j]; }
    for (int p = prow; p < HW; p += pstep) {
        float v[CH], r[CH];
        ld_chunk(xb + (size_t)p * C, v);
        if (rb) ld_chunk(rb + (size_t)p * C, r);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            float o = v[j] * sc[j] - sh[j];
            if (rb) o += r[j];
            if (relu) o = relu_f(o);
            v[j] = o;
        }
        st_chunk(xb + (size_t)p * C, v);
    }
}

static long gn_slab_limit() {
    static const long v = dev_env("HCM_GN_SLAB") ? atol(dev_env("HCM_GN_SLAB")) : 32768;
    return v;
}

// Large feature maps (HW >= 256 pixels per sample): two launches over (sample, pixel chunk) workgroups that read whole
// pixel rows -- every access is a full 128-byte-or-wider run, where the slab kernel above would touch 16 bytes of each
// line (the stem's 64 x 64 x 64 map: 143 us -> ~30 us).  Launch 1 writes per-(sample, chunk, group) sum / sum of squares;
// launch 2 adds the <= 16 partials of its sample in a fixed order (deterministic, no atomics), normalises its chunk, adds
// the residual, applies ReLU.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ part, int HW, int C, int G, int P) {
    constexpr int CH = Tr<T>::CH;
    __shared__ float s_p1[4][512], s_p2[4][512];     // C <= 512 on this path
    const int b = blockIdx.y, pc = blockIdx.x;
    const int cpr = C / CH;                       // power of two, <= 64 (f16/bf16) or 128 (f32)
    const int tid = threadIdx.x;
    const int cc = tid % cpr, prow = tid / cpr, pstep = 256 / cpr;
    const int chunk = HW / P;
    const T* xb = x + ((size_t)b * HW + (size_t)pc * chunk) * C + cc * CH;
    float s1[CH], s2[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    for (int p = prow; p < chunk; p += pstep) {
        float v[CH];
        ld_chunk(xb + (size_t)p * C, v);
#pragma unroll
        for (int j = 0; j < CH; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
    }
    for (int o = 32; o >= cpr; o >>= 1) {
#pragma unroll
        for (int j = 0; j < CH; ++j) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
    }
    const int wv = tid >> 6;
    for (int i = tid; i < 4 * C; i += 256) { s_p1[i / C][i % C] = 0.f; s_p2[i / C][i % C] = 0.f; }
    __syncthreads();
    // cpr <= 64: after the butterfly lanes < cpr hold the wave's totals; cpr = 128 (f32): a wave covers half a pixel row and
    // every lane owns distinct channels (the other half-row waves leave zeros in this wave's slots)
    if ((tid & 63) < cpr || cpr > 64) {
#pragma unroll
        for (int j = 0; j < CH; ++j) { s_p1[wv][cc * CH + j] = s1[j]; s_p2[wv][cc * CH + j] = s2[j]; }
    }
    __syncthreads();
    const int Cg = C / G;
    for (int g = tid; g < G; g += 256) {
        float a = 0.f, q = 0.f;
        for (int j = 0; j < Cg; ++j) {
            const int ch = g * Cg + j;
            a += (s_p1[0][ch] + s_p1[1][ch]) + (s_p1[2][ch] + s_p1[3][ch]);
            q += (s_p2[0][ch] + s_p2[1][ch]) + (s_p2[2][ch] + s_p2[3][ch]);
        }
        float* o = part + (((size_t)b * P + pc) * G + g) * 2;
        o[0] = a; o[1] = q;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ part, int HW, int C, int G,
                                                        int P, float eps, int relu, int PS) {
    constexpr int CH = Tr<T>::CH;
    __shared__ float s_scale[512], s_shift[512];
    __shared__ float s_mean[256], s_rstd[256];
    const int b = blockIdx.y, pc = blockIdx.x;
    const int tid = threadIdx.x;
    const int Cg = C / G;
    // PS partial sums per (sample, group): P of them from gn_stats_kernel, or HW / 64 from the producing conv's epilogue
    for (int g = tid; g < G; g += 256) {
        float a = 0.f, q = 0.f;
        {   // index order, requested 16 at a time (see maxpool_gn_kernel)
            const float2* pp = reinterpret_cast<const float2*>(part) + (size_t)b * PS * G + g;
            int i = 0;
            for (; i + 16 <= PS; i += 16) {
                float2 t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = pp[(size_t)(i + u) * G];
#pragma unroll
                for (int u = 0; u < 16; ++u) { a += t[u].x; q += t[u].y; }
            }
            for (; i < PS; ++i) { const float2 t = pp[(size_t)i * G]; a += t.x; q += t.y; }
        }
        const float inv_n = 1.0f / ((float)HW * (float)Cg);
        const float mean = a * inv_n;
        const float var = relu_f(q * inv_n - mean * mean);
        s_mean[g] = mean;
        s_rstd[g] = rsqrtf(var + eps);
    }
    __syncthreads();
    for (int ch = tid; ch < C; ch += 256) {
        const int g = ch / Cg;
        const float sc = s_rstd[g] * gamma[ch];
        s_scale[ch] = sc;
        s_shift[ch] = s_mean[g] * sc - beta[ch];
    }
    __syncthreads();
    const int cpr = C / CH;
    const int cc = tid % cpr, prow = tid / cpr, pstep = 256 / cpr;
    const int chunk = HW / P;
    const size_t base = ((size_t)b * HW + (size_t)pc * chunk) * C + cc * CH;
    float sc[CH], sh[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { sc[j] = s_scale[cc * CH + j]; sh[j] = s_shift[cc * CH + j]; }
    for (int p = prow; p < chunk; p += pstep) {
        float v[CH], r[CH];
        ld_chunk(x + base + (size_t)p * C, v);
        if (res) ld_chunk(res + base + (size_t)p * C, r);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            float o = v[j] * sc[j] - sh[j];
            if (res) o += r[j];
            if (relu) o = relu_f(o);
            v[j] = o;
        }
        st_chunk(x + base + (size_t)p * C, v);
    }
}

// Fallback for shapes the slab kernel's layout does not cover (a group wider than 1024 channels: the 1 x 1 x 2048 compression map of
// 64-pixel depth frames): one workgroup per (sample, group), two passes over its HW x Cg elements, fixed summation order.
template <typename T>
__global__ __launch_bounds__(256) void gn_generic_kernel(T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int HW, int C, int G, float eps, int relu, int cg_true) {
    __shared__ float s_a[256], s_q[256];
    const int Cg = C / G, b = blockIdx.x / G, g = blockIdx.x - b * G, tid = threadIdx.x;
    T* xb = x + (size_t)b * HW * C + (size_t)g * Cg;
    const T* rb = res ? res + (size_t)b * HW * C + (size_t)g * Cg : nullptr;
    const int n = HW * Cg;
    float a = 0.f, q = 0.f;
    for (int e = tid; e < n; e += 256) {
        const int p = e / Cg, c = e - p * Cg;
        const float v = Tr<T>::ld(xb + (size_t)p * C + c);
        a += v; q += v * v;
    }
    s_a[tid] = a; s_q[tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { s_a[tid] += s_a[tid + o]; s_q[tid] += s_q[tid + o]; }
        __syncthreads();
    }
    const float inv_n = 1.0f / ((float)HW * (float)(cg_true > 0 ? cg_true : Cg));
    const float mean = s_a[0] * inv_n;
    const float rstd = rsqrtf(relu_f(s_q[0] * inv_n - mean * mean) + eps);
    for (int e = tid; e < n; e += 256) {
        const int p = e / Cg, c = e - p * Cg;
        const size_t off = (size_t)p * C + c;
        float o = (Tr<T>::ld(xb + off) - mean) * rstd * gamma[g * Cg + c] + beta[g * Cg + c];
        if (rb) o += Tr<T>::ld(rb + off);
        if (relu) o = relu_f(o);
        Tr<T>::st(xb + off, o);
    }
}

hipError_t launch_groupnorm(void* x, const void* res, const float* gamma, const float* beta, float* stats, int dt, int B,
                            int HW, int C, int G, float eps, int relu, hipStream_t s, int cg_true) {
    const int CH = dt_chunk(dt);
    if (C % CH || C % G) return hipErrorInvalidValue;
    const int Cg = C / G;
    if (cg_true < 0 || cg_true > Cg) return hipErrorInvalidValue;
    static const int two_pass = dev_env("HCM_GN_TWO") ? atoi(dev_env("HCM_GN_TWO")) : 1;
    const int P = gn_partials(HW);
    const int cprw = C / CH;
    if (two_pass && !cg_true && stats && P > 0 && !(cprw & (cprw - 1)) && cprw <= 128 && C <= 512 && G <= 256 && HW % P == 0 &&
        (HW >= 1024 || C <= 256)) {             // 16 x 16 maps with 512 channels: the slab kernel is faster (13 vs 21 us)
        HCM_DISPATCH_T(dt, {
            hipLaunchKernelGGL(gn_stats_kernel<T>, dim3(P, B), dim3(256), 0, s, (const T*)x, stats, HW, C, G, P);
            hipLaunchKernelGGL(gn_apply_kernel<T>, dim3(P, B), dim3(256), 0, s, (T*)x, (const T*)res, gamma, beta, stats, HW, C, G, P, eps, relu, P);
        });
        return hipGetLastError();
    }
    // slab: whole groups, multiple of the chunk, power-of-two chunk count <= 32, about 32 K elements per workgroup
    int unit = Cg > CH ? Cg : CH;
    if (unit % Cg || unit % CH) {
        HCM_DISPATCH_T(dt, hipLaunchKernelGGL(gn_generic_kernel<T>, dim3(B * G), dim3(256), 0, s, (T*)x, (const T*)res, gamma, beta, HW, C, G, eps, relu, cg_true));
        return hipGetLastError();
    }
    int CS = unit;
    while (CS * 2 <= C && CS * 2 <= 256 && (CS * 2) / CH <= 32 && (long)HW * CS * 2 <= gn_slab_limit() && C % (CS * 2) == 0) CS *= 2;
    const int cpr = CS / CH;
    if (cpr & (cpr - 1) || cpr > 256 || CS > 1024 || C % CS) {
        HCM_DISPATCH_T(dt, hipLaunchKernelGGL(gn_generic_kernel<T>, dim3(B * G), dim3(256), 0, s, (T*)x, (const T*)res, gamma, beta, HW, C, G, eps, relu, cg_true));
        return hipGetLastError();
    }
    const int grid = B * (C / CS);
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(gn_fused_kernel<T>, dim3(grid), dim3(256), 0, s, (T*)x, (const T*)res, gamma, beta, HW, C, G, CS, eps, relu, cg_true));
    return hipGetLastError();
}

bool groupnorm_apply_ok(int dt, int HW, int C, int G) {
    const int CH = dt_chunk(dt);
    if (dt == DT_F32 || C % CH || C % G) return false;
    const int P = gn_partials(HW), cprw = C / CH;
    // (no `HW >= 1024 || C <= 256` clause as in launch_groupnorm: without the statistics launch the apply pass alone beats the slab kernel
    // on the 16 x 16 x 512 maps too)
    return P > 0 && !(cprw & (cprw - 1)) && cprw <= 128 && C <= 512 && G <= 256 && HW % P == 0 && HW % 64 == 0 && 32 % (C / G) == 0;
}
hipError_t launch_groupnorm_apply(void* x, const void* res, const float* gamma, const float* beta, const float* part, int PS, int dt, int B,
                                  int HW, int C, int G, float eps, int relu, hipStream_t s) {
    if (!groupnorm_apply_ok(dt, HW, C, G) || PS < 1) return hipErrorInvalidValue;
    const int P = gn_partials(HW);
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(gn_apply_kernel<T>, dim3(P, B), dim3(256), 0, s, (T*)x, (const T*)res, gamma, beta, part, HW, C, G, P, eps,
                                          relu, PS));
    return hipGetLastError();
}

// gn_apply_kernel with an UN-normalised residual (round 5): x = relu?(GN(x) + round_T(GN2(res))) where res is the raw output of the block's down-sample
// conv and part2 / gamma2 / beta2 its GroupNorm -- the stage-first bottlenecks of the GroupNorm trunk (habitat's ResNet: `out = relu(bn3(conv3) +
// downsample(x))`, resnet_encoders.py:27-33).  The down-sample branch's own apply pass (a read and a write of its map, one launch) disappears; its
// normalised value is rounded to the storage type exactly as that pass stored it before it is added: BIT-IDENTICAL to the two apply passes.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply2_kernel(T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ part, const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                         const float* __restrict__ part2, int HW, int C, int G, int P, float eps, float eps2, int relu, int PS) {
    constexpr int CH = Tr<T>::CH;
    __shared__ float s_scale[2][512], s_shift[2][512];
    __shared__ float s_mean[2][256], s_rstd[2][256];
    const int b = blockIdx.y, pc = blockIdx.x;
    const int tid = threadIdx.x;
    const int Cg = C / G;
    for (int g2 = tid; g2 < 2 * G; g2 += 256) {
        const int which = g2 >= G, g = which ? g2 - G : g2;
        const float* pp = which ? part2 : part;
        float a = 0.f, q = 0.f;
        {   // index order, requested 16 at a time (see maxpool_gn_kernel)
            const float2* p2 = reinterpret_cast<const float2*>(pp) + (size_t)b * PS * G + g;
            int i = 0;
            for (; i + 16 <= PS; i += 16) {
                float2 t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = p2[(size_t)(i + u) * G];
#pragma unroll
                for (int u = 0; u < 16; ++u) { a += t[u].x; q += t[u].y; }
            }
            for (; i < PS; ++i) { const float2 t = p2[(size_t)i * G]; a += t.x; q += t.y; }
        }
        const float inv_n = 1.0f / ((float)HW * (float)Cg);
        const float mean = a * inv_n;
        const float var = relu_f(q * inv_n - mean * mean);
        s_mean[which][g] = mean;
        s_rstd[which][g] = rsqrtf(var + (which ? eps2 : eps));
    }
    __syncthreads();
    for (int c2 = tid; c2 < 2 * C; c2 += 256) {
        const int which = c2 >= C, ch = which ? c2 - C : c2;
        const int g = ch / Cg;
        const float sc = s_rstd[which][g] * (which ? gamma2 : gamma)[ch];
        s_scale[which][ch] = sc;
        s_shift[which][ch] = s_mean[which][g] * sc - (which ? beta2 : beta)[ch];
    }
    __syncthreads();
    const int cpr = C / CH;
    const int cc = tid % cpr, prow = tid / cpr, pstep = 256 / cpr;
    const int chunk = HW / P;
    const size_t base = ((size_t)b * HW + (size_t)pc * chunk) * C + cc * CH;
    float sc[CH], sh[CH], sc2[CH], sh2[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { sc[j] = s_scale[0][cc * CH + j]; sh[j] = s_shift[0][cc * CH + j]; sc2[j] = s_scale[1][cc * CH + j]; sh2[j] = s_shift[1][cc * CH + j]; }
    for (int p = prow; p < chunk; p += pstep) {
        float v[CH], r[CH];
        ld_chunk(x + base + (size_t)p * C, v);
        ld_chunk(res + base + (size_t)p * C, r);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            T rt;
            Tr<T>::st(&rt, r[j] * sc2[j] - sh2[j]);          // what the down-sample branch's apply pass stored (no ReLU on that branch)
            float o = v[j] * sc[j] - sh[j];
            o += Tr<T>::ld(&rt);
            if (relu) o = relu_f(o);
            v[j] = o;
        }
        st_chunk(x + base + (size_t)p * C, v);
    }
}
hipError_t launch_groupnorm_apply2(void* x, const void* res, const float* gamma, const float* beta, const float* part, const float* gamma2, const float* beta2,
                                   const float* part2, int PS, int dt, int B, int HW, int C, int G, float eps, float eps2, int relu, hipStream_t s) {
    if (!groupnorm_apply_ok(dt, HW, C, G) || PS < 1 || !res || !part2) return hipErrorInvalidValue;
    const int P = gn_partials(HW);
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(gn_apply2_kernel<T>, dim3(P, B), dim3(256), 0, s, (T*)x, (const T*)res, gamma, beta, part, gamma2, beta2, part2, HW, C, G, P,
                                          eps, eps2, relu, PS));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ LayerNorm (one wave per row)
// nn.LayerNorm / BertLayerNorm: biased variance, eps inside the sqrt; two-pass in registers.
template <typename T, int D>
__global__ void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ post, int post_rows,
                                 T* __restrict__ y, int rows, float eps) {
    constexpr int PER = D / 64;              // elements per lane, strided by 64 lanes in groups of VEC
    constexpr int VEC = (PER % 4 == 0) ? 4 : 1;
    constexpr int NV = PER / VEC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + (size_t)row * D;
    const T* rr = res ? res + (size_t)row * D : nullptr;
    float v[PER];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = (i * 64 + lane) * VEC + j;
            float a = Tr<T>::ld(xr + c);
            if (rr) a += Tr<T>::ld(rr + c);
            v[i * VEC + j] = a;
            sum += a;
        }
    const float mean = wave_sum(sum) * (1.0f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const float d = v[i] - mean; sq += d * d; }
    const float rstd = rsqrtf(wave_sum(sq) * (1.0f / D) + eps);
    const float* pp = post ? post + (size_t)(row % post_rows) * D : nullptr;
    T* yr = y + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = (i * 64 + lane) * VEC + j;
            float o = (v[i * VEC + j] - mean) * rstd * gamma[c] + beta[c];
            if (pp) o += pp[c];
            Tr<T>::st(yr + c, o);
        }
}

// Vectorised variant: 32 lanes per row, 16-byte chunks (D/CH chunks per row, D/CH/32 per lane), two rows per wave.
template <typename T, int D, bool HAS_RES, bool HAS_POST>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ post, int post_rows,
                                                             T* __restrict__ y, int rows, float eps) {
    constexpr int CH = Tr<T>::CH;
    constexpr int CPL = D / CH / 32;         // chunks per lane
    static_assert(CPL >= 1 && D % (CH * 32) == 0, "row must split into 32 x 16-byte chunks");
    const int sub = threadIdx.x & 31;
    const int row = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (row >= rows) return;
    const T* xr = x + (size_t)row * D;
    const T* rr = res ? res + (size_t)row * D : nullptr;
    float v[CPL][CH];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = (i * 32 + sub) * CH;
        ld_chunk(xr + c, v[i]);
        if constexpr (HAS_RES) {
            float r[CH];
            ld_chunk(rr + c, r);
#pragma unroll
            for (int j = 0; j < CH; ++j) v[i][j] += r[j];
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) sum += v[i][j];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum * (1.0f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int j = 0; j < CH; ++j) { const float d = v[i][j] - mean; sq += d * d; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = rsqrtf(sq * (1.0f / D) + eps);
    const float* pp = post ? post + (size_t)(row % post_rows) * D : nullptr;
    T* yr = y + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = (i * 32 + sub) * CH;
        float o[CH], gm[CH], bt[CH];
        // gamma / beta / post as 16-byte vector loads (c is a multiple of CH), no per-element branches
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
            const float4 g4 = *reinterpret_cast<const float4*>(gamma + c + 4 * q);
            const float4 b4 = *reinterpret_cast<const float4*>(beta + c + 4 * q);
            gm[4 * q] = g4.x; gm[4 * q + 1] = g4.y; gm[4 * q + 2] = g4.z; gm[4 * q + 3] = g4.w;
            bt[4 * q] = b4.x; bt[4 * q + 1] = b4.y; bt[4 * q + 2] = b4.z; bt[4 * q + 3] = b4.w;
            if constexpr (HAS_POST) {
                const float4 p4 = *reinterpret_cast<const float4*>(pp + c + 4 * q);
                bt[4 * q] += p4.x; bt[4 * q + 1] += p4.y; bt[4 * q + 2] += p4.z; bt[4 * q + 3] += p4.w;
            }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) o[j] = (v[i][j] - mean) * rstd * gm[j] + bt[j];
        st_chunk(yr + c, o);
    }
}

hipError_t launch_layernorm(const void* x, const void* res, const float* gamma, const float* beta, const float* post,
                            int post_rows, void* y, int dt, int rows, int D, float eps, hipStream_t s) {
    const int pr = post_rows > 0 ? post_rows : 1;
    if (D == 768 || D == 256 || D == 512) {
        constexpr int rpb = 8;                   // rows per workgroup (32 lanes each); 2..8 measured equal: 4.2 us, launch floor
        const dim3 grid((rows + rpb - 1) / rpb), block(32 * rpb);
#define LV2(DD, R_, P_) hipLaunchKernelGGL((layernorm_vec_kernel<T, DD, R_, P_>), grid, block, 0, s, (const T*)x, (const T*)res, gamma, beta, post, pr, (T*)y, rows, eps)
#define LV(DD) do { if (res) { if (post) LV2(DD, true, true); else LV2(DD, true, false); } else { if (post) LV2(DD, false, true); else LV2(DD, false, false); } } while (0)
        HCM_DISPATCH_T(dt, { if (D == 768) LV(768); else if (D == 256) LV(256); else LV(512); });
#undef LV
#undef LV2
        return hipGetLastError();
    }
    const int wpb = 4;
    const dim3 grid((rows + wpb - 1) / wpb), block(64 * wpb);
#define L(DD) hipLaunchKernelGGL((layernorm_kernel<T, DD>), grid, block, 0, s, (const T*)x, (const T*)res, gamma, beta, post, pr, (T*)y, rows, eps)
    HCM_DISPATCH_T(dt, { if (D == 128) L(128); else return hipErrorInvalidValue; });
#undef L
    return hipGetLastError();
}

// LayerNorm of an f32 row with two outputs (the bf16 BERT keeps its residual stream in f32: forward.cpp bert()): y16 = the next GEMM's
// 16-bit operand, y32 = the stream itself.  32 lanes per row, float4 chunks.
template <typename T, int D>
__global__ __launch_bounds__(256) void layernorm_f32in_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               T* __restrict__ y16, float* __restrict__ y32, int rows, float eps) {
    constexpr int CPL = D / 4 / 32;
    static_assert(CPL >= 1 && D % 128 == 0, "row must split into 32 x 16-byte chunks");
    const int sub = threadIdx.x & 31;
    const int row = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    float v[CPL][4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        ld_chunk(xr + (i * 32 + sub) * 4, v[i]);
        sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum * (1.0f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; sq += d * d; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = rsqrtf(sq * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = (i * 32 + sub) * 4;
        const float4 g4 = *reinterpret_cast<const float4*>(gamma + c), b4 = *reinterpret_cast<const float4*>(beta + c);
        float o[4] = {(v[i][0] - mean) * rstd * g4.x + b4.x, (v[i][1] - mean) * rstd * g4.y + b4.y,
                      (v[i][2] - mean) * rstd * g4.z + b4.z, (v[i][3] - mean) * rstd * g4.w + b4.w};
        st_chunk(y32 + (size_t)row * D + c, o);
        T o4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) Tr<T>::st(&o4[j], o[j]);
        *reinterpret_cast<uint2*>(y16 + (size_t)row * D + c) = *reinterpret_cast<const uint2*>(o4);
    }
}
// ... and its round-5 sibling for the implicit stream: the 16-bit operand + (mean, rstd) per row, NO f32 output (forward.cpp bert(): the next projection's
// epilogue rebuilds the stream value from the sum and these statistics).  A kernel of its own: layernorm_f32in_kernel's instruction stream is what the
// goldens of the "bf16" mode and the fused-block experiment are pinned to, and adding two conditional stores to it changed its floating-point
// contraction (1.6e-5 on the record).
template <typename T, int D>
__global__ __launch_bounds__(256) void layernorm_f32in_stats_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     T* __restrict__ y16, float* __restrict__ stats, int rows, float eps) {
    constexpr int CPL = D / 4 / 32;
    const int sub = threadIdx.x & 31;
    const int row = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    float v[CPL][4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        ld_chunk(xr + (i * 32 + sub) * 4, v[i]);
        sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum * (1.0f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; sq += d * d; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = rsqrtf(sq * (1.0f / D) + eps);
    if (sub == 0) *reinterpret_cast<float2*>(stats + 2 * (size_t)row) = make_float2(mean, rstd);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = (i * 32 + sub) * 4;
        const float4 g4 = *reinterpret_cast<const float4*>(gamma + c), b4 = *reinterpret_cast<const float4*>(beta + c);
        T o4[4];
        Tr<T>::st(&o4[0], (v[i][0] - mean) * rstd * g4.x + b4.x); Tr<T>::st(&o4[1], (v[i][1] - mean) * rstd * g4.y + b4.y);
        Tr<T>::st(&o4[2], (v[i][2] - mean) * rstd * g4.z + b4.z); Tr<T>::st(&o4[3], (v[i][3] - mean) * rstd * g4.w + b4.w);
        *reinterpret_cast<uint2*>(y16 + (size_t)row * D + c) = *reinterpret_cast<const uint2*>(o4);
    }
}
hipError_t launch_layernorm_f32in(const float* x, const float* gamma, const float* beta, void* y16, float* y32, int dt, int rows, int D, float eps,
                                  hipStream_t s, float* stats) {
    if (dt != DT_BF16 && dt != DT_F16) return hipErrorInvalidValue;
    if (stats) {                                   // the statistics form: no f32 output
        if (y32) return hipErrorInvalidValue;
        constexpr int rpb2 = 8;
        const dim3 grid2((rows + rpb2 - 1) / rpb2), block2(32 * rpb2);
#define LS(T, DD) hipLaunchKernelGGL((layernorm_f32in_stats_kernel<T, DD>), grid2, block2, 0, s, x, gamma, beta, (T*)y16, stats, rows, eps)
        if (D == 768) { if (dt == DT_BF16) LS(bf16, 768); else LS(f16, 768); }
        else if (D == 256) { if (dt == DT_BF16) LS(bf16, 256); else LS(f16, 256); }
        else if (D == 512) { if (dt == DT_BF16) LS(bf16, 512); else LS(f16, 512); }
        else return hipErrorInvalidValue;
#undef LS
        return hipGetLastError();
    }
    constexpr int rpb = 8;
    const dim3 grid((rows + rpb - 1) / rpb), block(32 * rpb);
#define LF(T, DD) hipLaunchKernelGGL((layernorm_f32in_kernel<T, DD>), grid, block, 0, s, x, gamma, beta, (T*)y16, y32, rows, eps)
    if (D == 768) { if (dt == DT_BF16) LF(bf16, 768); else LF(f16, 768); }
    else if (D == 256) { if (dt == DT_BF16) LF(bf16, 256); else LF(f16, 256); }
    else if (D == 512) { if (dt == DT_BF16) LF(bf16, 512); else LF(f16, 512); }
    else return hipErrorInvalidValue;
#undef LF
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ BERT embeddings + LN
// BertEmbeddings.forward: word[id] + position[l] + token_type[0] -> LayerNorm(eps 1e-12)  (call site seq2seq_highlevel_cma.py:192-195)
template <typename T, typename I>
__global__ void bert_embed_kernel(const I* __restrict__ ids, const float* __restrict__ word, const float* __restrict__ pos,
                                  const float* __restrict__ type0, const float* __restrict__ gamma, const float* __restrict__ beta,
                                  T* __restrict__ y, int rows, int L, int D, int vocab, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    long id = (long)ids[row];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    const int l = row % L;
    const float* wr = word + (size_t)id * D;
    const float* pr = pos + (size_t)l * D;
    float v[16];                              // D <= 1024
    const int per = D / 64;
    float sum = 0.f;
    for (int i = 0; i < per; ++i) {
        const int c = i * 64 + lane;
        const float a = wr[c] + pr[c] + type0[c];
        v[i] = a;
        sum += a;
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
    for (int i = 0; i < per; ++i) { const float d = v[i] - mean; sq += d * d; }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    for (int i = 0; i < per; ++i) {
        const int c = i * 64 + lane;
        Tr<T>::st(y + (size_t)row * D + c, (v[i] - mean) * rstd * gamma[c] + beta[c]);
    }
}

hipError_t launch_bert_embed(const void* ids, int ids_dt, const float* word, const float* pos, const float* type0,
                             const float* gamma, const float* beta, void* y, int dt, int B, int L, int D, int vocab,
                             float eps, hipStream_t s) {
    const int rows = B * L;
    if (D % 64 || D > 1024) return hipErrorInvalidValue;
    const dim3 grid((rows + 3) / 4), block(256);
#define L_(I) hipLaunchKernelGGL((bert_embed_kernel<T, I>), grid, block, 0, s, (const I*)ids, word, pos, type0, gamma, beta, (T*)y, rows, L, D, vocab, eps)
    HCM_DISPATCH_T(dt, {
        if (ids_dt == DT_I64) L_(int64_t); else if (ids_dt == DT_I32) L_(int32_t); else if (ids_dt == DT_F32) L_(float); else return hipErrorInvalidValue;
    });
#undef L_
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ recurrent cell + heads
// RNNStateEncoder.single_forward (models/decoder/state_encoder.py:72-81): hidden * mask, one step, repack.
__global__ void rnn_prep_kernel(const float* __restrict__ h_in, const float* __restrict__ mask, float* __restrict__ xh,
                                int B, int Hd, int ld, int col0) {
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < Hd; j += blockDim.x) xh[(size_t)b * ld + col0 + j] = h_in[(size_t)b * Hd + j] * mask[b];
}
hipError_t launch_rnn_prep(const float* h_in, const float* mask, float* xh, int B, int Hd, int ld, int col0, hipStream_t s) {
    hipLaunchKernelGGL(rnn_prep_kernel, dim3(B), dim3(256), 0, s, h_in, mask, xh, B, Hd, ld, col0);
    return hipGetLastError();
}

__device__ __forceinline__ void heads_eval(const float* hs, int Hd, const Heads& hd, int b) {
    // hs: new hidden state of sample b in LDS; one wave per output row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __shared__ float lg[64];
    __shared__ int best_s;
    for (int r = wave; r < hd.r0 + hd.r1; r += nw) {
        const bool first = r < hd.r0;
        const int rr = first ? r : r - hd.r0;
        const float* w = (first ? hd.w0 : hd.w1) + (size_t)rr * Hd;
        float acc = 0.f;
        for (int j = lane; j < Hd; j += 64) acc += w[j] * hs[j];
        acc = wave_sum(acc);
        if (lane == 0) {
            if (first) { const float v = acc + hd.b0[rr]; hd.out0[(size_t)b * hd.ld0 + rr] = v; if (rr < 64) lg[rr] = v; }
            else hd.out1[(size_t)b * hd.ld1 + rr] = acc + hd.b1[rr];
        }
    }
    if (hd.pred) {                        // (kernel-uniform) argmax of the first head + embedding row of the winner
        __syncthreads();
        if (threadIdx.x == 0) {
            int best = 0;
            float bv = lg[0];
            for (int j = 1; j < hd.r0; ++j)
                if (lg[j] > bv) { bv = lg[j]; best = j; }
            hd.pred[b] = best;
            best_s = best < hd.emb_rows ? best : hd.emb_rows - 1;
        }
        __syncthreads();
        const float* e = hd.emb + (size_t)best_s * hd.emb_dim;
        for (int j = threadIdx.x; j < hd.emb_dim; j += blockDim.x) hd.emb_out[(size_t)b * hd.emb_ld + j] = e[j];
    }
}

__global__ void lstm_cell_kernel(const float* __restrict__ gates, const float* __restrict__ h_in, const float* __restrict__ mask,
                                 float* __restrict__ h_out, int B, int Hd, Heads hd) {
    extern __shared__ float hs[];
    const int b = blockIdx.x;
    const float mk = mask[b];
    const float* g = gates + (size_t)b * 4 * Hd;
    bool bad = false;
    for (int j = threadIdx.x; j < Hd; j += blockDim.x) {
        const float c = h_in[(size_t)(B + b) * Hd + j] * mk;          // hidden[1] = c
        bad |= !isfinite(g[j] + g[Hd + j] + g[2 * Hd + j] + g[3 * Hd + j]);       // (a finite sum has finite terms)
        const float gi = sigmoidf_(g[j]), gf = sigmoidf_(g[Hd + j]), gg = tanhf(g[2 * Hd + j]), go = sigmoidf_(g[3 * Hd + j]);
        const float c2 = gf * c + gi * gg;
        const float h2 = go * tanhf(c2);
        hs[j] = h2;
        h_out[(size_t)b * Hd + j] = h2;
        h_out[(size_t)(B + b) * Hd + j] = c2;
    }
    if (hd.bad && __syncthreads_or(bad) && threadIdx.x == 0) atomicAdd(hd.bad, 1u);
    __syncthreads();
    heads_eval(hs, Hd, hd, b);
}
hipError_t launch_lstm_cell(const float* gates, const float* h_in, const float* mask, float* h_out, int B, int Hd,
                            const Heads& heads, hipStream_t s) {
    hipLaunchKernelGGL(lstm_cell_kernel, dim3(B), dim3(256), Hd * sizeof(float), s, gates, h_in, mask, h_out, B, Hd, heads);
    return hipGetLastError();
}

__global__ void gru_cell_kernel(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ h_in,
                                const float* __restrict__ mask, float* __restrict__ h_out, int B, int Hd, Heads hd) {
    extern __shared__ float hs[];
    const int b = blockIdx.x;
    const float mk = mask[b];
    const float* a = gi + (size_t)b * 3 * Hd;
    const float* c = gh + (size_t)b * 3 * Hd;
    bool bad = false;
    for (int j = threadIdx.x; j < Hd; j += blockDim.x) {
        const float h = h_in[(size_t)b * Hd + j] * mk;
        bad |= !isfinite(a[j] + a[Hd + j] + a[2 * Hd + j]) || !isfinite(c[j] + c[Hd + j] + c[2 * Hd + j]);
        const float r = sigmoidf_(a[j] + c[j]);
        const float z = sigmoidf_(a[Hd + j] + c[Hd + j]);
        const float n = tanhf(a[2 * Hd + j] + r * c[2 * Hd + j]);
        const float h2 = (1.f - z) * n + z * h;
        hs[j] = h2;
        h_out[(size_t)b * Hd + j] = h2;
    }
    if (hd.bad && __syncthreads_or(bad) && threadIdx.x == 0) atomicAdd(hd.bad, 1u);
    __syncthreads();
    heads_eval(hs, Hd, hd, b);
}
hipError_t launch_gru_cell(const float* gi, const float* gh, const float* h_in, const float* mask, float* h_out, int B,
                           int Hd, const Heads& heads, hipStream_t s) {
    hipLaunchKernelGGL(gru_cell_kernel, dim3(B), dim3(256), Hd * sizeof(float), s, gi, gh, h_in, mask, h_out, B, Hd, heads);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ range check (fp16 calibration)
// slot[0] = max |x| as float bits (non-negative floats order like unsigned ints), slot[1] = count of non-finite elements
template <typename T>
__global__ void absmax_kernel(const T* __restrict__ x, int rows, int cols, int ld, unsigned* __restrict__ slot) {
    const size_t total = (size_t)rows * cols;
    float mx = 0.f;
    unsigned bad = 0;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const float v = Tr<T>::ld(x + (e / cols) * (size_t)ld + e % cols);
        if (!(fabsf(v) <= 3.0e38f)) ++bad; else mx = fmaxf(mx, fabsf(v));
    }
    mx = wave_max(mx);
    bad += __shfl_xor(bad, 32, 64); bad += __shfl_xor(bad, 16, 64); bad += __shfl_xor(bad, 8, 64);
    bad += __shfl_xor(bad, 4, 64); bad += __shfl_xor(bad, 2, 64); bad += __shfl_xor(bad, 1, 64);
    if ((threadIdx.x & 63) == 0) {
        atomicMax(slot, __float_as_uint(mx));
        if (bad) atomicAdd(slot + 1, bad);
    }
}
hipError_t launch_absmax(const void* x, int dt, int rows, int cols, int ld, unsigned* slot, hipStream_t s) {
    const size_t total = (size_t)rows * cols;
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(absmax_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)x, rows, cols, ld, slot));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ split-K reduction
// y[m][n] = act(sum_s part[s][m][n] + bias[n]) in a fixed order (deterministic); part is f32 [S][M][N], y is T or f32 at ldy.
template <typename T>
__global__ void splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, void* __restrict__ y, int S, int M,
                                     int N, int ldy, int act, int out_f32) {
    const size_t total = (size_t)M * N;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(e % N);
        const size_t m = e / N;
        float a = bias ? bias[n] : 0.f;
        // the same left-to-right sum, eight independent requests at a time (as a plain loop the S loads were one latency chain: 25 us at 56 slices)
        int s_ = 0;
        for (; s_ + 8 <= S; s_ += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(s_ + j) * total + e];
#pragma unroll
            for (int j = 0; j < 8; ++j) a += v[j];
        }
        for (; s_ < S; ++s_) a += part[(size_t)s_ * total + e];
        if (act == ACT_RELU) a = relu_f(a);
        else if (act == ACT_GELU) a = gelu_erf(a);
        if (out_f32) reinterpret_cast<float*>(y)[m * ldy + n] = a;
        else Tr<T>::st(reinterpret_cast<T*>(y) + m * ldy + n, a);
    }
}
hipError_t launch_splitk_reduce(const float* part, const float* bias, void* y, int dt, int S, int M, int N, int ldy, int act, int out_f32,
                                hipStream_t s) {
    const size_t total = (size_t)M * N;
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3(grid_for(total)), dim3(256), 0, s, part, bias, y, S, M, N, ldy, act, out_f32));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ CMANet instruction encoder
// InstructionEncoder.forward (models/encoders/instruction_encoder.py:70-92): lengths = #non-zero ids, embedding lookup.
// x[row][0..E) = table[id], zero-padded to ldx columns; one block per sample also counts its length.
template <typename I>
__global__ void instr_embed_kernel(const I* __restrict__ ids, const float* __restrict__ table, float* __restrict__ x,
                                   int* __restrict__ lengths, int L, int E, int ldx, int vocab) {
    const int b = blockIdx.x;
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int t = threadIdx.x; t < L; t += blockDim.x) mine += ((long)ids[(size_t)b * L + t] != 0) ? 1 : 0;
    if (mine) atomicAdd(&cnt, mine);
    for (int e = threadIdx.x; e < L * ldx; e += blockDim.x) {
        const int t = e / ldx, j = e - t * ldx;
        long id = (long)ids[(size_t)b * L + t];
        if (id < 0) id = 0;
        if (id >= vocab) id = vocab - 1;
        x[((size_t)b * L + t) * ldx + j] = j < E ? table[(size_t)id * E + j] : 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) lengths[b] = cnt;
}
hipError_t launch_instr_embed(const void* ids, int ids_dt, const float* table, float* x, int* lengths, int B, int L, int E, int ldx,
                              int vocab, hipStream_t s) {
    if (ids_dt == DT_I64) hipLaunchKernelGGL(instr_embed_kernel<int64_t>, dim3(B), dim3(256), 0, s, (const int64_t*)ids, table, x, lengths, L, E, ldx, vocab);
    else if (ids_dt == DT_I32) hipLaunchKernelGGL(instr_embed_kernel<int32_t>, dim3(B), dim3(256), 0, s, (const int32_t*)ids, table, x, lengths, L, E, ldx, vocab);
    else if (ids_dt == DT_F32) hipLaunchKernelGGL(instr_embed_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)ids, table, x, lengths, L, E, ldx, vocab);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// One time step of the PACKED LSTM (nn.LSTM over pack_padded_sequence): sample b is active at step t iff t < len_b; an
// inactive sample keeps (h, c) and emits zeros -- which makes the reverse direction start at the sample's own last
// token.  pre = x_t W_ih^T + b_ih + b_hh for every token [B*L][4H]; gh = h W_hh^T [B][4H]; gate order i,f,g,o.
__global__ void instr_lstm_cell_kernel(const float* __restrict__ pre, const float* __restrict__ gh, float* __restrict__ h,
                                       float* __restrict__ c, const int* __restrict__ lengths, float* __restrict__ out, int t, int L,
                                       int Hd, int ld_out, int col0) {
    const int b = blockIdx.x;
    const bool act = t < lengths[b];
    const float* p = pre + ((size_t)b * L + t) * 4 * Hd;
    const float* g = gh + (size_t)b * 4 * Hd;
    float* o = out + ((size_t)b * L + t) * ld_out + col0;
    for (int j = threadIdx.x; j < Hd; j += blockDim.x) {
        if (!act) { o[j] = 0.f; continue; }
        const float gi = sigmoidf_(p[j] + g[j]), gf = sigmoidf_(p[Hd + j] + g[Hd + j]);
        const float gg = tanhf(p[2 * Hd + j] + g[2 * Hd + j]), go = sigmoidf_(p[3 * Hd + j] + g[3 * Hd + j]);
        const float c2 = gf * c[(size_t)b * Hd + j] + gi * gg;
        const float h2 = go * tanhf(c2);
        c[(size_t)b * Hd + j] = c2;
        h[(size_t)b * Hd + j] = h2;
        o[j] = h2;
    }
}
// The same step for nn.GRU (INSTRUCTION_ENCODER.rnn_type = "GRU", instruction_encoder.py:42): pre = x_t W_ih^T + b_ih [B*L][3H], gh = h W_hh^T + b_hh
// [B][3H], gate order r, z, n; n = tanh(pre_n + r * gh_n), h' = (1 - z) n + z h.
__global__ void instr_gru_cell_kernel(const float* __restrict__ pre, const float* __restrict__ gh, float* __restrict__ h,
                                      const int* __restrict__ lengths, float* __restrict__ out, int t, int L, int Hd, int ld_out, int col0) {
    const int b = blockIdx.x;
    const bool act = t < lengths[b];
    const float* p = pre + ((size_t)b * L + t) * 3 * Hd;
    const float* g = gh + (size_t)b * 3 * Hd;
    float* o = out + ((size_t)b * L + t) * ld_out + col0;
    for (int j = threadIdx.x; j < Hd; j += blockDim.x) {
        if (!act) { o[j] = 0.f; continue; }
        const float r = sigmoidf_(p[j] + g[j]), z = sigmoidf_(p[Hd + j] + g[Hd + j]);
        const float n = tanhf(p[2 * Hd + j] + r * g[2 * Hd + j]);
        const float h2 = (1.0f - z) * n + z * h[(size_t)b * Hd + j];
        h[(size_t)b * Hd + j] = h2;
        o[j] = h2;
    }
}
hipError_t launch_instr_gru_cell(const float* pre, const float* gh, float* h, const int* lengths, float* out, int t, int B, int L, int Hd, int ld_out,
                                 int col0, hipStream_t s) {
    hipLaunchKernelGGL(instr_gru_cell_kernel, dim3(B), dim3(256), 0, s, pre, gh, h, lengths, out, t, L, Hd, ld_out, col0);
    return hipGetLastError();
}
hipError_t launch_instr_lstm_cell(const float* pre, const float* gh, float* h, float* c, const int* lengths, float* out, int t, int B,
                                  int L, int Hd, int ld_out, int col0, hipStream_t s) {
    hipLaunchKernelGGL(instr_lstm_cell_kernel, dim3(B), dim3(256), 0, s, pre, gh, h, c, lengths, out, t, L, Hd, ld_out, col0);
    return hipGetLastError();
}

// The whole packed (bi)LSTM scan in ONE launch: a workgroup owns SB samples of one direction for all L steps.  h lives in LDS
// (double-buffered, broadcast reads), c in registers; W_hh is read from L2 every step, transposed and gate-interleaved ([k][unit][4
// gates]: the four gate weights of (k, unit j) are one 16-byte load, coalesced across the units) -- 4*H*H*4 bytes per step and
// workgroup.  SB is chosen small (2 for B = 64, both directions: 64 workgroups): per step a thread then does SB * 256 FMAs behind 64
// 16-byte loads, and the step is bounded by that weight stream (1 MB from L2) instead of by 8 samples' worth of FMAs and LDS
// broadcasts on 16 CUs (CMANet step at B = 64, L = 80: 3.15 -> 2.56 ms; the trunks alone take 2.4).
struct LstmScanArgs {
    const float* pre[2];       // per direction: x_t W_ih^T + b_ih + b_hh for every token, [B*L][4H]
    const float* wt[2];        // per direction: W_hh as [H (k)][H (unit)][4 (gate)]
};
// 1024 threads = 256 hidden units x 4 quarters of the k range: each thread streams a quarter of its unit's recurrent weights
// (64 k x 4 gates, several loads in flight) for SB samples; the four partial sums per (gate, sample, unit) meet in LDS.
template <int SB>
__global__ __launch_bounds__(1024) void instr_lstm_scan_kernel(LstmScanArgs a, const int* __restrict__ lengths, float* __restrict__ out,
                                                                int B, int L, int H, int ld_out) {
    extern __shared__ float sm[];          // h [2][SB][H], partials [3][4][SB][H]
    float* hs = sm;
    float* ps = sm + 2 * SB * H;
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * SB;
    const int j = threadIdx.x & 255, kq = threadIdx.x >> 8;
    const int KQ = H / 4;
    const float* __restrict__ pre = a.pre[dir];
    const float* __restrict__ wt = a.wt[dir] + ((size_t)kq * KQ * H + j) * 4;
    float c[SB];
#pragma unroll
    for (int s_ = 0; s_ < SB; ++s_) c[s_] = 0.f;
    for (int i = threadIdx.x; i < 2 * SB * H; i += 1024) hs[i] = 0.f;
    __syncthreads();
    int cur = 0;
    for (int step = 0; step < L; ++step) {
        const int t = dir ? L - 1 - step : step;
        float acc[4][SB];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s_ = 0; s_ < SB; ++s_) acc[g][s_] = 0.f;
        const float* hc = hs + cur * SB * H + kq * KQ;
#pragma unroll 8
        for (int k = 0; k < KQ; ++k) {
            const float4 w4 = *reinterpret_cast<const float4*>(wt + (size_t)k * 4 * H);
            const float w0 = w4.x, w1 = w4.y, w2 = w4.z, w3 = w4.w;
#pragma unroll
            for (int s_ = 0; s_ < SB; ++s_) {
                const float hk = hc[s_ * H + k];
                acc[0][s_] += w0 * hk; acc[1][s_] += w1 * hk; acc[2][s_] += w2 * hk; acc[3][s_] += w3 * hk;
            }
        }
        if (kq > 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int s_ = 0; s_ < SB; ++s_) ps[(((kq - 1) * 4 + g) * SB + s_) * H + j] = acc[g][s_];
        }
        __syncthreads();
        float* hn = hs + (cur ^ 1) * SB * H;
        if (kq == 0) {
#pragma unroll
            for (int s_ = 0; s_ < SB; ++s_) {
                const int b = b0 + s_;
                const bool valid = b < B;
                const bool act = valid && t < lengths[valid ? b : 0];
                float hv = hs[cur * SB * H + s_ * H + j], o = 0.f;
                if (act) {
                    const float* p = pre + ((size_t)b * L + t) * 4 * H + j;
                    float gsum[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        gsum[g] = p[g * H] + ((acc[g][s_] + ps[((0 * 4 + g) * SB + s_) * H + j]) + (ps[((1 * 4 + g) * SB + s_) * H + j] + ps[((2 * 4 + g) * SB + s_) * H + j]));
                    const float gi = sigmoidf_(gsum[0]), gf = sigmoidf_(gsum[1]), gg = tanhf(gsum[2]), go = sigmoidf_(gsum[3]);
                    c[s_] = gf * c[s_] + gi * gg;
                    hv = go * tanhf(c[s_]);
                    o = hv;
                }
                hn[s_ * H + j] = hv;
                if (valid) out[((size_t)b * L + t) * ld_out + dir * H + j] = o;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}
hipError_t launch_instr_lstm_scan(const float* pre0, const float* pre1, const float* wt0, const float* wt1, const int* lengths, float* out,
                                  int B, int L, int H, int dirs, int ld_out, hipStream_t s) {
    if (H != 256) return hipErrorInvalidValue;          // 256 hidden units x 4 k-quarters = 1024 threads
    LstmScanArgs a;
    a.pre[0] = pre0; a.pre[1] = pre1; a.wt[0] = wt0; a.wt[1] = wt1;
    // samples per workgroup: the fewest that keep the launch within one workgroup per CU
    static const char* fsb = dev_env("HCM_LSTM_SB");
    // (SB = 1 is not built: slower at B = 64 -- 128 workgroups take the CUs from the trunks running beside the scan -- and its
    // instantiation showed run-to-run differences of ~3e-5 whose cause was not found; SB = 2 / 4 / 8 are bitwise reproducible, tested)
    int SB = 2;
    while (SB < 8 && ((B + SB - 1) / SB) * dirs > 128) SB *= 2;
    if (fsb) SB = atoi(fsb);
    const size_t lds = (size_t)(2 * SB * H + 3 * 4 * SB * H) * sizeof(float);
    const void* fn = SB == 2 ? reinterpret_cast<const void*>(instr_lstm_scan_kernel<2>)
                   : SB == 4 ? reinterpret_cast<const void*>(instr_lstm_scan_kernel<4>) : SB == 8 ? reinterpret_cast<const void*>(instr_lstm_scan_kernel<8>) : nullptr;
    if (!fn) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    void* args[] = {&a, &lengths, &out, &B, &L, &H, &ld_out};
    return hipLaunchKernel(fn, dim3((B + SB - 1) / SB, dirs), dim3(1024), args, lds, s);
}

// CMANet._attn (models/cma.py:201-209): ONE query per sample over S positions:
//   logits[s] = q . k[s];  logits -= 1e8 where masked (s >= len_b when lengths != null);  p = softmax(logits * scale);
//   out = sum_s p[s] v[s].   q [B][D] (ldq), k [B][S][.] (ldk), v [B][S][.] (ldv), out [B][.] (ldo); all f32.
__global__ void attn1q_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk, const float* __restrict__ v,
                              int ldv, const int* __restrict__ lengths, float* __restrict__ out, int ldo, int S, int D, int Dv,
                              float scale) {
    extern __shared__ float sh[];            // [S] probabilities
    __shared__ float red[2];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const float* qb = q + (size_t)b * ldq;
    const int len = lengths ? lengths[b] : S;
    for (int sidx = wave; sidx < S; sidx += nw) {
        const float* kr = k + ((size_t)b * S + sidx) * ldk;
        float acc = 0.f;
        for (int j = lane; j < D; j += 64) acc += qb[j] * kr[j];
        acc = wave_sum(acc);
        if (lane == 0) sh[sidx] = (acc - (sidx >= len ? 1e8f : 0.f)) * scale;
    }
    __syncthreads();
    if (wave == 0) {
        float m = -3.0e38f;
        for (int i = lane; i < S; i += 64) m = fmaxf(m, sh[i]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float sum = 0.f;
        for (int i = lane; i < S; i += 64) sum += expf(sh[i] - m);
        sum = wave_sum(sum);
        if (lane == 0) { red[0] = m; red[1] = 1.f / sum; }
    }
    __syncthreads();
    const float m = red[0], inv = red[1];
    for (int i = threadIdx.x; i < S; i += blockDim.x) sh[i] = expf(sh[i] - m) * inv;
    __syncthreads();
    for (int cidx = threadIdx.x; cidx < Dv; cidx += blockDim.x) {
        float acc = 0.f;
        const float* vb = v + (size_t)b * S * ldv + cidx;
        for (int i = 0; i < S; ++i) acc += sh[i] * vb[(size_t)i * ldv];
        out[(size_t)b * ldo + cidx] = acc;
    }
}
hipError_t launch_attn1q(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const int* lengths, float* out,
                         int ldo, int B, int S, int D, int Dv, float scale, hipStream_t s) {
    hipLaunchKernelGGL(attn1q_kernel, dim3(B), dim3(256), S * sizeof(float), s, q, ldq, k, ldk, v, ldv, lengths, out, ldo, S, D, Dv, scale);
    return hipGetLastError();
}

// torch.argmax(output, dim=1) (hierarchical_trainer.py:1098): first maximal index
__global__ void argmax_kernel(const float* __restrict__ logits, int64_t* __restrict__ pred, int B, int n, int ld) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* p = logits + (size_t)b * ld;
    int best = 0;
    float bv = p[0];
    for (int j = 1; j < n; ++j)
        if (p[j] > bv) { bv = p[j]; best = j; }
    pred[b] = best;
}
hipError_t launch_argmax(const float* logits, int64_t* pred, int B, int n, int ld, hipStream_t s) {
    hipLaunchKernelGGL(argmax_kernel, dim3((B + 63) / 64), dim3(64), 0, s, logits, pred, B, n, ld);
    return hipGetLastError();
}

// sub_task_embedding lookup (seq2seq_lowlevel.py:141)
__global__ void embed_rows_kernel(const float* __restrict__ emb, const int64_t* __restrict__ idx, float* __restrict__ y,
                                  int B, int D, int ld, int col0, int nrows) {
    const int b = blockIdx.x;
    long i = idx[b];
    if (i < 0) i = 0;
    if (i >= nrows) i = nrows - 1;
    for (int j = threadIdx.x; j < D; j += blockDim.x) y[(size_t)b * ld + col0 + j] = emb[(size_t)i * D + j];
}
hipError_t launch_embed_rows(const float* emb, const int64_t* idx, float* y, int B, int D, int ld, int col0, int nrows, hipStream_t s) {
    hipLaunchKernelGGL(embed_rows_kernel, dim3(B), dim3(64), 0, s, emb, idx, y, B, D, ld, col0, nrows);
    return hipGetLastError();
}

template <typename T>
__global__ void to_f32_kernel(const T* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) y[e] = Tr<T>::ld(x + e);
}
template <typename T>
__global__ void from_f32_kernel(const float* __restrict__ x, T* __restrict__ y, size_t n) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) Tr<T>::st(y + e, x[e]);
}
hipError_t launch_convert_to_f32(const void* x, int dt, float* y, size_t n, hipStream_t s) {
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(to_f32_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, (const T*)x, y, n));
    return hipGetLastError();
}
hipError_t launch_convert_from_f32(const float* x, void* y, int dt, size_t n, hipStream_t s) {
    HCM_DISPATCH_T(dt, hipLaunchKernelGGL(from_f32_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, x, (T*)y, n));
    return hipGetLastError();
}

template <typename A, typename B_>
__global__ void convert_kernel(const A* __restrict__ x, B_* __restrict__ y, size_t n) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) Tr<B_>::st(y + e, Tr<A>::ld(x + e));
}
hipError_t launch_convert(const void* x, int dt_in, void* y, int dt_out, size_t n, hipStream_t s) {
    if (dt_in == dt_out) return hipMemcpyAsync(y, x, n * dt_size(dt_in), hipMemcpyDeviceToDevice, s);
    HCM_DISPATCH_T(dt_in, {
        using A = T;
        if (dt_out == DT_BF16) hipLaunchKernelGGL((convert_kernel<A, bf16>), dim3(grid_for(n)), dim3(256), 0, s, (const A*)x, (bf16*)y, n);
        else if (dt_out == DT_F16) hipLaunchKernelGGL((convert_kernel<A, f16>), dim3(grid_for(n)), dim3(256), 0, s, (const A*)x, (f16*)y, n);
        else if (dt_out == DT_F32) hipLaunchKernelGGL((convert_kernel<A, float>), dim3(grid_for(n)), dim3(256), 0, s, (const A*)x, (float*)y, n);
        else return hipErrorInvalidValue;
    });
    return hipGetLastError();
}


// Development aid (make DEV=1, HCM_MARKS=1): a one-lane kernel that stores the constant 100 MHz wall clock -- placed by forward.cpp between the
// launches of a chain, it records WHEN that chain reached that point of the (hipGraph-replayed, multi-stream) step.  rocprofv3's kernel trace
// serialises the streams (kernels in flight: 1 for 91 % of a traced step), so it cannot show how the three chains really interleave.
__global__ void mark_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
hipError_t launch_mark(unsigned long long* slot, hipStream_t s) {
    hipLaunchKernelGGL(mark_kernel, dim3(1), dim3(1), 0, s, slot);
    return hipGetLastError();
}

// Busy-wait for `ticks` of the 100 MHz wall clock on one wave: the stream-overlap probe of api.cpp (two streams that share a hardware queue run two of these
// back to back, two that do not run them side by side).
__global__ void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
hipError_t launch_spin(unsigned long long ticks, hipStream_t s) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks);
    return hipGetLastError();
}

}  // namespace hcm
