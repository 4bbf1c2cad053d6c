// One launch per cross-modal layer for BOTH Visual_Ling_Attn calls (rgb, depth): everything of InterModuleAttnLayer.forward
// (models/transformer/transformer.py:209-221) that follows the query / key / value projections --
//
//   att  = softmax(Q K^T / sqrt(64)) V                      ScaledDotProductAttention  (:81-109)   [in-kernel when Lk <= 32: layer 0,
//                                                                                                   whose keys are the 16 visual tokens]
//   x1   = LayerNorm(I + att Wo^T + bo)                     MultiHeadAttention         (:111-126)  (the residual is the query stream I)
//   out  = LayerNorm(x1 + relu(x1 W1^T + b1) W2^T + b2)     PositionWiseFeedForward    (:25-43)
//   pool = mean over the instruction's tokens of out        cross_pooler, seq2seq_highlevel_cma.py:209-210 (last layer only)
//
// for a block of up to 80 instruction tokens of one (environment, visual stream) per workgroup.  The activations of the block never
// leave the CU between these steps: attention output, x1 and the 256-wide slices of the 1024-wide FFN intermediate live in LDS as
// 16-bit MFMA operands (rows padded to 528 B: conflict-free fragment reads), the two LayerNorms reduce over the 8 waves through a
// small LDS table, the FFN runs in four 256-column slices (slice c of fc1 feeds k-range c of fc2, accumulated in registers).  Weights
// are the MFMA A operand and stream straight from L2 into fragment registers (1 MB per layer, shared by all workgroups); a lane's
// accumulator holds 4 consecutive channels of one token, as everywhere in this library.
//
// Against the seven launches it replaces (attention, fc_o, LayerNorm, fc1, fc2, LayerNorm, mean) the GEMMs use the same MFMA
// instruction over the same k order on the same rounded operands; the LayerNorm / mean reductions have a different (fixed) order, so
// results agree with the unfused path to fp32 round-off of those reductions (tests/test_fusion_toggles_gpu.py), not bit for bit.
#include <cstdlib>
#include "kernels.h"
#include "dev.h"

namespace hcm {

typedef float v_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 v_bf16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct VMma;
template <> struct VMma<bf16> {
    static __device__ __forceinline__ void run(v_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v_bf16x8, a), __builtin_bit_cast(v_bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct VMma<f16> {
    static __device__ __forceinline__ void run(v_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), acc, 0, 0, 0);
    }
};

constexpr int V_D = 256, V_RB = 80, V_MF = 5, V_LDA = 528;          // model width, rows per workgroup, 16-row fragments, LDS row bytes
constexpr int V_KVMAX = 32;                                         // keys of the in-kernel attention
constexpr size_t V_LDS = (size_t)2 * V_RB * V_LDA + (size_t)V_KVMAX * 1024 + (size_t)8 * V_RB * 2 * sizeof(float);

template <typename T> __device__ __forceinline__ void v_st4(char* p, const float (&v)[4]) {
    T o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) Tr<T>::st(&o[e], v[e]);
    *reinterpret_cast<uint2*>(p) = *reinterpret_cast<const uint2*>(o);
}
template <typename T> __device__ __forceinline__ void v_ld4(const char* p, float (&v)[4]) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    T o[4];
    *reinterpret_cast<uint2*>(o) = u;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = Tr<T>::ld(&o[e]);
}

// acc[i][j] += W[n = nbase + i*16 + fr][k0 + ks*32 + fg*8 ..] x A[row j*16 + fr][ks*32 + fg*8 ..] for ks = 0..7 (256 of K).
// All 16 weight fragments of the call are requested up front (64 VGPRs): the loads are L2 hits of ~500 cycles each, and issued one k-step
// at a time they were the kernel's critical path (82 us per launch; MFMA work is 11 us).
template <typename T>
__device__ __forceinline__ void v_wload(uint4 (&wf)[8][2], const T* __restrict__ W, int ldw, int nbase, int k0, int fr, int fg) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[ks][i] = *reinterpret_cast<const uint4*>(W + (size_t)(nbase + i * 16 + fr) * ldw + k0 + ks * 32 + fg * 8);
}
template <typename T>
__device__ __forceinline__ void v_mma256(v_f32x4 (&acc)[2][V_MF], const uint4 (&wf)[8][2], const char* sA, int fr, int fg) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        uint4 xf[V_MF];
#pragma unroll
        for (int j = 0; j < V_MF; ++j) xf[j] = *reinterpret_cast<const uint4*>(sA + (j * 16 + fr) * V_LDA + (ks * 32 + fg * 8) * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < V_MF; ++j) VMma<T>::run(acc[i][j], wf[ks][i], xf[j]);
    }
}

// LayerNorm over the 256 channels of every row of the block: a lane holds v[i][j][e] = channel wave*32 + i*16 + fg*4 + e of row j*16 + fr.
// Per row: sums over the lane's 8 values -> the 4 lane groups (xor 16, 32) -> the 8 waves through sRed; fixed order, no atomics.
template <typename T>
__device__ __forceinline__ void v_layernorm(float (&v)[2][V_MF][4], const float* __restrict__ gamma, const float* __restrict__ beta, float* sRed, int wave,
                                            int fr, int fg) {
#pragma unroll
    for (int j = 0; j < V_MF; ++j) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { a += v[i][j][e]; q += v[i][j][e] * v[i][j][e]; }
        a += __shfl_xor(a, 16, 64); q += __shfl_xor(q, 16, 64);
        a += __shfl_xor(a, 32, 64); q += __shfl_xor(q, 32, 64);
        if (fg == 0) { sRed[(wave * V_RB + j * 16 + fr) * 2] = a; sRed[(wave * V_RB + j * 16 + fr) * 2 + 1] = q; }
    }
    __syncthreads();
    float g4[2][4], b4[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + wave * 32 + i * 16 + fg * 4), b = *reinterpret_cast<const float4*>(beta + wave * 32 + i * 16 + fg * 4);
        g4[i][0] = g.x; g4[i][1] = g.y; g4[i][2] = g.z; g4[i][3] = g.w;
        b4[i][0] = b.x; b4[i][1] = b.y; b4[i][2] = b.z; b4[i][3] = b.w;
    }
#pragma unroll
    for (int j = 0; j < V_MF; ++j) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { a += sRed[(w * V_RB + j * 16 + fr) * 2]; q += sRed[(w * V_RB + j * 16 + fr) * 2 + 1]; }
        const float mean = a * (1.0f / V_D);
        const float rstd = rsqrtf(fmaxf(q * (1.0f / V_D) - mean * mean, 0.f) + 1e-5f);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][j][e] = (v[i][j][e] - mean) * rstd * g4[i][e] + b4[i][e];
    }
    __syncthreads();                 // sRed may be written again
}

template <typename T>
__global__ __launch_bounds__(512) void vla_post_kernel(VlaPost p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                   // attention output, later the FFN intermediate slice, last the output image
    char* sX = smem + V_RB * V_LDA;                    // x1 = LayerNorm(I + att Wo^T)
    char* sKV = sX + V_RB * V_LDA;                     // K | V of the in-kernel attention: [Lk][512] T
    float* sRed = reinterpret_cast<float*>(sKV + V_KVMAX * 1024);
    const int st = blockIdx.y;
    const int nblk = (p.L + V_RB - 1) / V_RB;
    const int b = blockIdx.x / nblk, r0 = (blockIdx.x - b * nblk) * V_RB;
    const int nrow = p.L - r0 < V_RB ? p.L - r0 : V_RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const T* q = reinterpret_cast<const T*>(p.q) + ((size_t)b * p.L + r0) * V_D;
    const T* I = reinterpret_cast<const T*>(p.I) + ((size_t)b * p.L + r0) * V_D;
    T* out = reinterpret_cast<T*>(p.out[st]) + ((size_t)b * p.L + r0) * V_D;

    const int nb = wave * 32;                          // this wave's 32 output channels of every 256-wide GEMM
    uint4 wf[8][2];
    v_wload<T>(wf, reinterpret_cast<const T*>(p.wo), V_D, nb, 0, fr, fg);      // fc_o's fragments travel while the attention runs
    // ---- attention output of the block into sA (rows >= nrow: zeros)
    if (p.fuse_att) {
        const int Lk = p.Lk[st];
        const T* kv = reinterpret_cast<const T*>(p.kv[st]) + (size_t)b * Lk * 512;
        for (int e = tid; e < Lk * 64; e += 512)
            *reinterpret_cast<uint4*>(sKV + e * 16) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(kv) + (size_t)e * 16);
        __syncthreads();
        for (int task = tid; task < V_RB * 4; task += 512) {
            const int row = task >> 2, head = task & 3;
            float o[64];
#pragma unroll
            for (int d = 0; d < 64; ++d) o[d] = 0.f;
            if (row < nrow) {
                float qv[64];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float t8[8];
                    ld_chunk(q + (size_t)row * V_D + head * 64 + c * 8, t8);
#pragma unroll
                    for (int d = 0; d < 8; ++d) qv[c * 8 + d] = t8[d] * 0.125f;               // 1 / sqrt(64)
                }
                // online softmax over the (few) keys: running maximum m, normaliser l, un-normalised output o
                float m = -3.0e38f, l = 0.f;
                for (int k = 0; k < Lk; ++k) {
                    const T* kp = reinterpret_cast<const T*>(sKV + k * 1024) + head * 64;
                    float s = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        float t8[8];
                        ld_chunk(kp + c * 8, t8);
#pragma unroll
                        for (int d = 0; d < 8; ++d) s += qv[c * 8 + d] * t8[d];
                    }
                    const float mn = fmaxf(m, s);
                    const float alpha = __expf(m - mn), pk = __expf(s - mn);
                    l = l * alpha + pk;
                    const T* vp = kp + 256;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        float t8[8];
                        ld_chunk(vp + c * 8, t8);
#pragma unroll
                        for (int d = 0; d < 8; ++d) o[c * 8 + d] = o[c * 8 + d] * alpha + pk * t8[d];
                    }
                    m = mn;
                }
                const float inv = 1.0f / l;
#pragma unroll
                for (int d = 0; d < 64; ++d) o[d] *= inv;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float t8[8];
#pragma unroll
                for (int d = 0; d < 8; ++d) t8[d] = o[c * 8 + d];
                st_chunk(reinterpret_cast<T*>(sA + row * V_LDA) + head * 64 + c * 8, t8);
            }
        }
    } else {
        const T* att = reinterpret_cast<const T*>(p.att[st]) + ((size_t)b * p.L + r0) * V_D;
        for (int e = tid; e < V_RB * 32; e += 512) {
            const int row = e >> 5, c = e & 31;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (row < nrow) v = *reinterpret_cast<const uint4*>(att + (size_t)row * V_D + c * 8);
            *reinterpret_cast<uint4*>(sA + row * V_LDA + c * 16) = v;
        }
    }
    __syncthreads();

    float v[2][V_MF][4];
    // ---- x1 = LayerNorm(I + att Wo^T + bo)
    {
        v_f32x4 acc[2][V_MF];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < V_MF; ++j) acc[i][j] = (v_f32x4){0.f, 0.f, 0.f, 0.f};
        v_mma256<T>(acc, wf, sA, fr, fg);
        v_wload<T>(wf, reinterpret_cast<const T*>(p.w1), V_D, nb, 0, fr, fg);   // fc1 slice 0: in flight during the LayerNorm
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = nb + i * 16 + fg * 4;
            const float4 bb = *reinterpret_cast<const float4*>(p.bo + n);
#pragma unroll
            for (int j = 0; j < V_MF; ++j) {
                const int row = j * 16 + fr;
                float r4[4] = {0.f, 0.f, 0.f, 0.f};
                if (row < nrow) v_ld4<T>(reinterpret_cast<const char*>(I + (size_t)row * V_D + n), r4);
                v[i][j][0] = acc[i][j][0] + bb.x + r4[0]; v[i][j][1] = acc[i][j][1] + bb.y + r4[1];
                v[i][j][2] = acc[i][j][2] + bb.z + r4[2]; v[i][j][3] = acc[i][j][3] + bb.w + r4[3];
            }
        }
    }
    v_layernorm<T>(v, p.g1, p.be1, sRed, wave, fr, fg);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < V_MF; ++j) v_st4<T>(sX + (j * 16 + fr) * V_LDA + (nb + i * 16 + fg * 4) * 2, v[i][j]);
    __syncthreads();

    // ---- FFN in 256-column slices of the intermediate: H_c = relu(x1 W1[c]^T + b1[c]) (-> sA), acc2 += H_c W2[:, c]^T
    v_f32x4 acc2[2][V_MF];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < V_MF; ++j) acc2[i][j] = (v_f32x4){0.f, 0.f, 0.f, 0.f};
    const int nslice = p.d_ff / 256;
    for (int c = 0; c < nslice; ++c) {
        v_f32x4 acc1[2][V_MF];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < V_MF; ++j) acc1[i][j] = (v_f32x4){0.f, 0.f, 0.f, 0.f};
        v_mma256<T>(acc1, wf, sX, fr, fg);
        v_wload<T>(wf, reinterpret_cast<const T*>(p.w2), p.d_ff, nb, c * 256, fr, fg);   // fc2 k-range c: in flight during the ReLU epilogue + barrier
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 bb = *reinterpret_cast<const float4*>(p.b1 + c * 256 + nb + i * 16 + fg * 4);
#pragma unroll
            for (int j = 0; j < V_MF; ++j) {
                const float h[4] = {fmaxf(acc1[i][j][0] + bb.x, 0.f), fmaxf(acc1[i][j][1] + bb.y, 0.f), fmaxf(acc1[i][j][2] + bb.z, 0.f),
                                    fmaxf(acc1[i][j][3] + bb.w, 0.f)};
                v_st4<T>(sA + (j * 16 + fr) * V_LDA + (nb + i * 16 + fg * 4) * 2, h);
            }
        }
        __syncthreads();
        v_mma256<T>(acc2, wf, sA, fr, fg);
        if (c + 1 < nslice) v_wload<T>(wf, reinterpret_cast<const T*>(p.w1), V_D, (c + 1) * 256 + nb, 0, fr, fg);
        __syncthreads();
    }
    // ---- out = LayerNorm(x1 + ffn + b2)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = nb + i * 16 + fg * 4;
        const float4 bb = *reinterpret_cast<const float4*>(p.b2 + n);
#pragma unroll
        for (int j = 0; j < V_MF; ++j) {
            float r4[4];
            v_ld4<T>(sX + (j * 16 + fr) * V_LDA + n * 2, r4);
            v[i][j][0] = acc2[i][j][0] + bb.x + r4[0]; v[i][j][1] = acc2[i][j][1] + bb.y + r4[1];
            v[i][j][2] = acc2[i][j][2] + bb.z + r4[2]; v[i][j][3] = acc2[i][j][3] + bb.w + r4[3];
        }
    }
    v_layernorm<T>(v, p.g2, p.be2, sRed, wave, fr, fg);
    // rounded output -> LDS image -> 16-byte row stores; pooled mean over the instruction's own tokens from the rounded values
    int len = p.L;
    if (p.lens) { len = p.lens[b]; len = len < 1 ? 1 : len > p.L ? p.L : len; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < V_MF; ++j) {
            char* dst = sA + (j * 16 + fr) * V_LDA + (nb + i * 16 + fg * 4) * 2;
            v_st4<T>(dst, v[i][j]);
            if (p.pooled[st] && r0 + j * 16 + fr < len) {
                float r4[4];
                v_ld4<T>(dst, r4);
#pragma unroll
                for (int e = 0; e < 4; ++e) cs[e] += r4[e];
            }
        }
        if (p.pooled[st]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = cs[e];
                t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
                cs[e] = t;
            }
            if (fr == 0) {
                float* dst = p.pooled[st] + (size_t)b * p.ld_pool + nb + i * 16 + fg * 4;
                const float invl = 1.0f / (float)len;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[e] = cs[e] * invl;
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < nrow * 32; e += 512) {
        const int row = e >> 5, c = e & 31;
        *reinterpret_cast<uint4*>(out + (size_t)row * V_D + c * 8) = *reinterpret_cast<const uint4*>(sA + row * V_LDA + c * 16);
    }
}

bool vla_post_ok(int dt, int d_model, int heads, int d_ff) {
    return (dt == DT_BF16 || dt == DT_F16) && d_model == V_D && heads == 4 && d_ff >= 256 && d_ff % 256 == 0;
}

hipError_t launch_vla_post(const VlaPost& p, int dt, hipStream_t s) {
    if (dt != DT_BF16 && dt != DT_F16) return hipErrorInvalidValue;
    if (p.B < 1 || p.L < 1 || p.d_ff % 256) return hipErrorInvalidValue;
    for (int st = 0; st < p.streams; ++st)
        if (p.fuse_att && (p.Lk[st] < 1 || p.Lk[st] > V_KVMAX)) return hipErrorInvalidValue;
    // the pooled mean is complete inside one workgroup only when the block covers the whole instruction
    if ((p.pooled[0] || p.pooled[1]) && p.L > V_RB) return hipErrorInvalidValue;
    const void* fn = dt == DT_BF16 ? reinterpret_cast<const void*>(vla_post_kernel<bf16>) : reinterpret_cast<const void*>(vla_post_kernel<f16>);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vla_post_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vla_post_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    VlaPost q = p;
    void* args[] = {&q};
    const int nblk = (p.L + V_RB - 1) / V_RB;
    return hipLaunchKernel(fn, dim3(p.B * nblk, p.streams), dim3(512), args, V_LDS, s);
}

}  // namespace hcm
