// softmax(Q K^T / sqrt(d)) V with head dim d = 64: BERT self-attention (12 heads, L x L, no mask -- the reference
// passes no attention_mask, seq2seq_highlevel_cma.py:194) and ScaledDotProductAttention of the cross-modal block
// (4 heads, L x 16 or L x L; models/transformer/transformer.py:81-109, masks None).
//
// v1 (fp32 path): one workgroup per (batch, head, block of 64 queries); K and V staged in LDS as f32 in chunks of up to
// KC_MAX keys (any Lk: the online softmax state carries across chunks, keys are visited in order whatever the chunking);
// four lanes share a query row (16 of the 64 dims each), online softmax in registers.  LDS-bandwidth bound, not MFMA bound.
#include <cstdlib>
#include "kernels.h"
#include "dev.h"

namespace hcm {

template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                        T* __restrict__ out, int heads, int Lq, int Lk0, int ldq, int ldk,
                                                        int ldv, int ldo, int q_batch_mod, int KC, const int* __restrict__ klens) {
    constexpr int CH = Tr<T>::CH;
    constexpr int D = 64;
    extern __shared__ __attribute__((aligned(16))) char smem_att[];
    float* Ks = reinterpret_cast<float*>(smem_att);
    float* Vs = Ks + (size_t)KC * D;

    const int b = blockIdx.x / heads;
    const int h = blockIdx.x % heads;
    const int bq = b % q_batch_mod;
    const int tid = threadIdx.x;
    int Lk = Lk0;                           // keys this sample attends over (uniform per workgroup); rows keep the stride Lk0
    if (klens) { Lk = klens[b]; Lk = Lk < 1 ? 1 : Lk > Lk0 ? Lk0 : Lk; }

    const int part = tid & 3;               // which 16-dim slice of the head
    {
        const int q0 = blockIdx.y * 64;
        const int qi = q0 + (tid >> 2);
        const bool active = qi < Lq;
        float qr[16], o[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { qr[j] = 0.f; o[j] = 0.f; }
        if (active) {
            const T* qp = q + ((size_t)bq * Lq + qi) * ldq + h * D + part * 16;
#pragma unroll
            for (int c = 0; c < 16; c += CH) {
                float t[CH];
                ld_chunk(qp + c, t);
#pragma unroll
                for (int j = 0; j < CH; ++j) qr[c + j] = t[j] * 0.125f;     // 1/sqrt(64)
            }
        }
        float m = -3.0e38f, l = 0.f;
        for (int k0 = 0; k0 < Lk; k0 += KC) {
        const int kn = Lk - k0 < KC ? Lk - k0 : KC;
        if (k0) __syncthreads();
        // stage this chunk of K, V (f32) -- pieces of CH elements
        for (int e = tid; e < kn * (D / CH); e += 256) {
            const int row = e / (D / CH);
            const int c = (e % (D / CH)) * CH;
            float kv[CH], vv[CH];
            ld_chunk(k + ((size_t)b * Lk0 + k0 + row) * ldk + h * D + c, kv);
            ld_chunk(v + ((size_t)b * Lk0 + k0 + row) * ldv + h * D + c, vv);
#pragma unroll
            for (int j = 0; j < CH; ++j) { Ks[row * D + c + j] = kv[j]; Vs[row * D + c + j] = vv[j]; }
        }
        __syncthreads();
        for (int key = 0; key < kn; ++key) {
            const float4* kp = reinterpret_cast<const float4*>(Ks + key * D + part * 16);
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 kk = kp[c];
                s += qr[4 * c] * kk.x + qr[4 * c + 1] * kk.y + qr[4 * c + 2] * kk.z + qr[4 * c + 3] * kk.w;
            }
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            const float mn = fmaxf(m, s);
            const float alpha = __expf(m - mn);
            const float pr = __expf(s - mn);
            l = l * alpha + pr;
            const float4* vp = reinterpret_cast<const float4*>(Vs + key * D + part * 16);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 vv = vp[c];
                o[4 * c] = o[4 * c] * alpha + pr * vv.x;
                o[4 * c + 1] = o[4 * c + 1] * alpha + pr * vv.y;
                o[4 * c + 2] = o[4 * c + 2] * alpha + pr * vv.z;
                o[4 * c + 3] = o[4 * c + 3] * alpha + pr * vv.w;
            }
            m = mn;
        }
        }
        if (active) {
            const float inv = 1.0f / l;
            T* op = out + ((size_t)b * Lq + qi) * ldo + h * D + part * 16;
#pragma unroll
            for (int c = 0; c < 16; c += CH) {
                float t[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) t[j] = o[c + j] * inv;
                st_chunk(op + c, t);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// MFMA variant for the 16-bit storage types (Lk <= 512; the score registers of a query tile are sized by the MAXKT template
// parameter, so short instructions keep the 160-key build's occupancy).  One workgroup per (batch, head), 4 waves, each wave owns
// 16-query tiles.  Scores are computed TRANSPOSED, S^T = K Q^T (A operand = K rows from LDS, B operand = Q rows
// straight from global), so that in the accumulator layout a lane holds 4 keys of ONE query per key tile: the
// softmax row reduction is in-register plus two cross-lane steps (xor 16, 32), and the exponentiated scores are
// already in A-operand position for P.V -- no LDS round trip for P.  The key order inside a 32-key MFMA step is
// permuted (slots 0-3: keys 4g..4g+3 of the even tile, 4-7: the same of the odd tile); V is staged transposed
// ([d][key], padded rows) and read with the same permutation, which a dot product is invariant to.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <typename T> struct AttMma;
template <> struct AttMma<bf16> {
    static __device__ __forceinline__ f32x4_t run(const uint4& a, const uint4& b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct AttMma<f16> {
    static __device__ __forceinline__ f32x4_t run(const uint4& a, const uint4& b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    }
};
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
    T t[2];
    Tr<T>::st(&t[0], a);
    Tr<T>::st(&t[1], b);
    return (uint32_t)t[0].v | ((uint32_t)t[1].v << 16);
}

template <typename T, int MAXKT>
__global__ __launch_bounds__(512) void attention_mfma_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                             T* __restrict__ out, int heads, int Lq, int Lk0, int ldq, int ldk,
                                                             int ldv, int ldo, int q_batch_mod, const int* __restrict__ klens) {
    int Lk = Lk0;                             // keys this sample attends over (uniform per workgroup); rows keep the stride Lk0
    if (klens) { Lk = klens[blockIdx.x / heads]; Lk = Lk < 1 ? 1 : Lk > Lk0 ? Lk0 : Lk; }
    constexpr int D = 64;                     // MAXKT 16-key tiles: Lk <= 16 * MAXKT (10 -> 160, 16 -> 256, 32 -> 512)
    extern __shared__ __attribute__((aligned(16))) char smem_att[];
    const int KT2 = (Lk + 31) / 32;           // 32-key MFMA steps
    const int Lkp = KT2 * 32;
    const int vstride = Lkp + 4;              // elements per V^T row (pad keeps the 8-byte reads spread over banks)
    char* Ks = smem_att;                      // [Lkp][64] T, 128-B rows, chunk index ^ (row & 7)
    T* Vt = reinterpret_cast<T*>(smem_att + (size_t)Lkp * 128);   // [64][vstride] T

    const int b = blockIdx.x / heads;
    const int h = blockIdx.x % heads;
    const int bq = b % q_batch_mod;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // stage K (swizzled rows) and V^T; rows >= Lk are zero
    // the first query tile's fragments are requested BEFORE the K / V staging: their round trip runs beside it instead of behind the barrier
    const int fr = lane & 15, fg = lane >> 4;
    const int QT = (Lq + 15) / 16;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    if (wave < QT) {
        int qrow = wave * 16 + fr;
        if (qrow >= Lq) qrow = Lq - 1;
        const T* qp = q + ((size_t)bq * Lq + qrow) * ldq + h * D;
        q0 = *reinterpret_cast<const uint4*>(qp + fg * 8);
        q1 = *reinterpret_cast<const uint4*>(qp + 32 + fg * 8);
    }
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;   // one wave per 16-query tile up to eight (round 4: L = 80 is five tiles -- four waves left three idle for half the kernel)
    for (int e = tid; e < Lkp * 8; e += nthreads) {
        const int row = e >> 3, c = e & 7;
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (row < Lk) {
            kv = *reinterpret_cast<const uint4*>(k + ((size_t)b * Lk0 + row) * ldk + h * D + c * 8);
            vv = *reinterpret_cast<const uint4*>(v + ((size_t)b * Lk0 + row) * ldv + h * D + c * 8);
        }
        *reinterpret_cast<uint4*>(Ks + row * 128 + ((c ^ (row & 7)) << 4)) = kv;
        const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            T t;
            t.v = (uint16_t)(w[j >> 1] >> ((j & 1) * 16));
            Vt[(size_t)(c * 8 + j) * vstride + row] = t;
        }
    }
    __syncthreads();

    for (int qt = wave; qt < QT; qt += nwaves) {
        // B operand: Q rows of this tile (clamped: rows past Lq compute garbage that is never stored)
        if (qt != wave) {
            int qrow = qt * 16 + fr;
            if (qrow >= Lq) qrow = Lq - 1;
            const T* qp = q + ((size_t)bq * Lq + qrow) * ldq + h * D;
            q0 = *reinterpret_cast<const uint4*>(qp + fg * 8);
            q1 = *reinterpret_cast<const uint4*>(qp + 32 + fg * 8);
        }

        f32x4_t sc[MAXKT];
        float mx = -3.0e38f;
#pragma unroll
        for (int t = 0; t < MAXKT; ++t) {
            sc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            if (t < 2 * KT2) {
                const int r = t * 16 + fr;
                const uint4 k0 = *reinterpret_cast<const uint4*>(Ks + r * 128 + (((0 + fg) ^ (r & 7)) << 4));
                const uint4 k1 = *reinterpret_cast<const uint4*>(Ks + r * 128 + (((4 + fg) ^ (r & 7)) << 4));
                sc[t] = AttMma<T>::run(k0, q0, sc[t]);
                sc[t] = AttMma<T>::run(k1, q1, sc[t]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = t * 16 + fg * 4 + e;
                    const float sv = key < Lk ? sc[t][e] * 0.125f : -3.0e38f;     // 1/sqrt(64); padded keys masked
                    sc[t][e] = sv;
                    mx = fmaxf(mx, sv);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < MAXKT; ++t)
            if (t < 2 * KT2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __expf(sc[t][e] - mx);
                    sc[t][e] = pv;
                    sum += pv;
                }
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);

        f32x4_t o[4];
#pragma unroll
        for (int dtile = 0; dtile < 4; ++dtile) o[dtile] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c2 = 0; c2 < MAXKT / 2; ++c2)
            if (c2 < KT2) {
                uint4 pa;
                pa.x = pack2<T>(sc[2 * c2][0], sc[2 * c2][1]);
                pa.y = pack2<T>(sc[2 * c2][2], sc[2 * c2][3]);
                pa.z = pack2<T>(sc[2 * c2 + 1][0], sc[2 * c2 + 1][1]);
                pa.w = pack2<T>(sc[2 * c2 + 1][2], sc[2 * c2 + 1][3]);
#pragma unroll
                for (int dtile = 0; dtile < 4; ++dtile) {
                    const T* vr = Vt + (size_t)(dtile * 16 + fr) * vstride + c2 * 32 + fg * 4;
                    const uint2 lo = *reinterpret_cast<const uint2*>(vr);
                    const uint2 hi = *reinterpret_cast<const uint2*>(vr + 16);
                    const uint4 vb = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    o[dtile] = AttMma<T>::run(pa, vb, o[dtile]);
                }
            }
        const float inv = 1.0f / sum;           // per query = per (lane & 15); rows of o are queries (fg*4 + e)
        // o[dtile][e] belongs to query qt*16 + fg*4 + e, column dtile*16 + fr; its normaliser lives in lane (fg*4+e)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float inv_q = __shfl(inv, fg * 4 + e, 64);
            const int qi = qt * 16 + fg * 4 + e;
            if (qi < Lq) {
                T* op = out + ((size_t)b * Lq + qi) * ldo + h * D + fr;
#pragma unroll
                for (int dtile = 0; dtile < 4; ++dtile) Tr<T>::st(op + dtile * 16, o[dtile][e] * inv_q);
            }
        }
    }
}

hipError_t launch_attention(const void* q, const void* k, const void* v, void* out, int dt, int B, int heads, int Lq,
                            int Lk, int ldq, int ldk, int ldv, int ldo, int q_batch_mod, hipStream_t s, const int* klens) {
    constexpr int KC_MAX = 256;                                   // keys per LDS chunk of the VALU kernel (128 KB of f32 K + V)
    const int KC = Lk < KC_MAX ? Lk : KC_MAX;
    const size_t lds = (size_t)KC * 64 * sizeof(float) * 2;
    if (Lk < 1 || Lq < 1) return hipErrorInvalidValue;
    const int CH = dt_chunk(dt);
    if ((ldq % CH) || (ldk % CH) || (ldv % CH) || (ldo % CH)) return hipErrorInvalidValue;
    if (q_batch_mod <= 0) q_batch_mod = B;
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_mfma_kernel<bf16, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_mfma_kernel<f16, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_mfma_kernel<bf16, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_mfma_kernel<f16, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_once.done();
    }
    static const char* valu = dev_env("HCM_ATT_VALU");
    if ((dt == DT_BF16 || dt == DT_F16) && Lk <= 512 && !(valu && atoi(valu))) {
        const int Lkp = (Lk + 31) / 32 * 32;
        const size_t lds2 = (size_t)Lkp * 128 + (size_t)64 * (Lkp + 4) * 2;          // 131.6 KB at Lk = 512
        static const bool w4 = dev_env("HCM_ATT_W4") != nullptr;          // (development build: the four-wave launch of rounds 1-3, for the A/B)
        const int qtiles = (Lq + 15) / 16;
        const int threads = w4 ? 256 : 64 * (qtiles < 4 ? 4 : qtiles > 8 ? 8 : qtiles);
#define LM(T, KT) hipLaunchKernelGGL((attention_mfma_kernel<T, KT>), dim3(B * heads), dim3(threads), lds2, s, (const T*)q, (const T*)k, (const T*)v, (T*)out, heads, Lq, Lk, ldq, ldk, ldv, ldo, q_batch_mod, klens)
        if (Lk <= 160) { if (dt == DT_BF16) LM(bf16, 10); else LM(f16, 10); }
        else if (Lk <= 256) { if (dt == DT_BF16) LM(bf16, 16); else LM(f16, 16); }
        else { if (dt == DT_BF16) LM(bf16, 32); else LM(f16, 32); }
#undef LM
        return hipGetLastError();
    }
#define LA(T) hipLaunchKernelGGL(attention_kernel<T>, dim3(B * heads, (Lq + 63) / 64), dim3(256), lds, s, (const T*)q, (const T*)k, (const T*)v, (T*)out, heads, Lq, Lk, ldq, ldk, ldv, ldo, q_batch_mod, KC, klens)
    if (dt == DT_BF16) LA(bf16); else if (dt == DT_F16) LA(f16); else if (dt == DT_F32) LA(float); else return hipErrorInvalidValue;
#undef LA
    return hipGetLastError();
}

}  // namespace hcm
