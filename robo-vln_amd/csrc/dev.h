// Device-side helpers shared by the HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace hcm {

struct bf16 { uint16_t v; };

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round to nearest even: v_cvt_pk_bf16_f32 (gfx950).  Round 5: rounds 1-4 did this in integer arithmetic (u + 0x7FFF + lsb, >> 16: three VALU
// operations per element, exact for finite inputs) -- the same bits for every finite value, a quiet NaN for a NaN (the integer form could carry
// into the sign), and every bf16 epilogue of the library one conversion per element PAIR instead of six operations.
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
typedef __bf16 dev_bf16x8 __attribute__((ext_vector_type(8)));

struct f16 { uint16_t v; };
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float8_t __attribute__((ext_vector_type(8)));

template <typename T> struct Tr;
template <> struct Tr<float> {
    static constexpr int CH = 4;     // elements per 16-byte chunk
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Tr<bf16> {
    static constexpr int CH = 8;
    static __device__ __forceinline__ float ld(const bf16* p) { return bf2f(p->v); }
    static __device__ __forceinline__ void st(bf16* p, float v) { p->v = f2bf(v); }
};

template <> struct Tr<f16> {
    static constexpr int CH = 8;
    static __device__ __forceinline__ float ld(const f16* p) { return (float)__builtin_bit_cast(_Float16, p->v); }
    static __device__ __forceinline__ void st(f16* p, float v) { p->v = __builtin_bit_cast(uint16_t, (_Float16)v); }
};

// load / store CH consecutive elements (16 bytes) as floats
__device__ __forceinline__ void ld_chunk(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void st_chunk(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ void ld_chunk(const bf16* p, float (&o)[8]) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[2 * i] = __uint_as_float(w[i] << 16);
        o[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ void st_chunk(bf16* p, const float (&o)[8]) {
    float8_t f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = o[i];
    *reinterpret_cast<dev_bf16x8*>(p) = __builtin_convertvector(f, dev_bf16x8);
}

__device__ __forceinline__ void ld_chunk(const f16* p, float (&o)[8]) {
    const half8_t h = *reinterpret_cast<const half8_t*>(p);
    const float8_t f = __builtin_convertvector(h, float8_t);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f[i];
}
// 16 bytes already in registers -> 8 floats (T selects the decoding)
template <typename T> __device__ __forceinline__ void cvt_chunk(const uint4& v, float (&o)[8]);
template <> __device__ __forceinline__ void cvt_chunk<bf16>(const uint4& v, float (&o)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[2 * i] = __uint_as_float(w[i] << 16);
        o[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
    }
}
template <> __device__ __forceinline__ void cvt_chunk<f16>(const uint4& v, float (&o)[8]) {
    const half8_t h = __builtin_bit_cast(half8_t, v);
    const float8_t f = __builtin_convertvector(h, float8_t);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f[i];
}
__device__ __forceinline__ void st_chunk(f16* p, const float (&o)[8]) {
    float8_t f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = o[i];
    *reinterpret_cast<half8_t*>(p) = __builtin_convertvector(f, half8_t);
}

// 8 floats -> 16 bytes of T in registers (one rounding each, as st_chunk)
template <typename T> __device__ __forceinline__ uint4 pack_chunk(const float (&o)[8]);
template <> __device__ __forceinline__ uint4 pack_chunk<bf16>(const float (&o)[8]) {
    float8_t f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = o[i];
    return __builtin_bit_cast(uint4, __builtin_convertvector(f, dev_bf16x8));
}
template <> __device__ __forceinline__ uint4 pack_chunk<f16>(const float (&o)[8]) {
    float8_t f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = o[i];
    return __builtin_bit_cast(uint4, __builtin_convertvector(f, half8_t));
}
// Accumulator tiles of two ADJACENT 16-channel blocks (a: channels c0 + fg*4 + r, b: channels c0 + 16 + fg*4 + r of pixel `lane & 15`) ->
// 8 CONSECUTIVE channels of that pixel per lane, v[0..7] = channels c0 + (fg & 1) * 16 + (fg >> 1) * 8 + 0..7: v_permlane16_swap_b32 exchanges the
// odd 16-lane rows of its first operand with the even rows of its second, which is exactly this regrouping.  No LDS, no barrier: the
// epilogue that follows stores 16 bytes per lane (64 contiguous bytes per pixel and wave instruction).
template <typename V4> __device__ __forceinline__ void swap_pair(const V4& a, const V4& b, float (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[r]), __float_as_uint(b[r]), false, false);
        v[r] = __uint_as_float(s[0]);
        v[4 + r] = __uint_as_float(s[1]);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// IEEE-754-2019 maximum (v_maximum3_f32 on gfx950): NaN-propagating, so that relu(NaN) = NaN and max-pool / variance clamps keep a NaN
// like torch's do -- with fmaxf (maxNum: "the other operand") a NaN or an inf - inf born upstream would be washed into zeros by the next
// ReLU and come out of the policy as a finite, wrong action; this way it reaches the recurrent cells' overflow guard (HCM_STEP_NONFINITE).
__device__ __forceinline__ float max_nan(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ float relu_f(float v) { return __builtin_elementwise_maximum(v, 0.f); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf-GELU for the 16-bit paths: erf by Abramowitz-Stegun 7.1.28, erf(z) = 1 - (1 + a1 z + ... + a6 z^6)^-16 for z >= 0 (|error| <= 3e-7
// absolute, i.e. three orders of magnitude below the rounding of the fp16 / bf16 value it is stored as), with 1/sqrt(2) folded into the
// coefficients (the polynomial is in |x|) and the sign handled by the identity
//     gelu(x) = 0.5 (x + |x| erf(|x| / sqrt 2)) = 0.5 (x + |x| (1 - r)),      r = p(|x|)^-16
// -- per element one v_and (|x|), one quarter-rate v_rcp, and 6 FMAs + 4 squarings + 3 more operations that all pack two elements per
// instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): 8.5 issue slots + the rcp, against the ~50-instruction branchy libm erff (a
// 256 x 256 output tile is 128 GELUs per thread, which with erff cost as much as the whole K = 768 main loop of BERT's FFN1: 44.8 -> 35.0 us
// per launch) and round 2's 7.1.26 form (v_rcp + v_exp, 11 slots + two quarter-rate instructions; stand-alone FFN1 35.5 -> 34.4 us).
// NaN -> NaN, -inf -> NaN as torch's formula gives them; +inf -> NaN where torch keeps +inf (non-finite either way: the next LayerNorm
// turns both into NaN, and the recurrent cells' guard counts it).  The fp32 path keeps the exact gelu_erf.
constexpr float kGeluA1 = 0.04986734688282013f, kGeluA2 = 0.02114100567996502f, kGeluA3 = 0.0032776263542473316f,
                kGeluA4 = 3.8003574445610866e-05f, kGeluA5 = 4.889063711743802e-05f, kGeluA6 = 5.38297490493278e-06f;
__device__ __forceinline__ float gelu_fast(float x) {
    const float ax = fabsf(x);
    float p = fmaf(ax, fmaf(ax, fmaf(ax, fmaf(ax, fmaf(ax, fmaf(ax, kGeluA6, kGeluA5), kGeluA4), kGeluA3), kGeluA2), kGeluA1), 1.0f);
    p *= p; p *= p; p *= p; p *= p;                                   // p^16 (overflows to inf from |x| ~ 16: r = 0, erf = 1)
    const float r = __builtin_amdgcn_rcpf(p);                         // 1 - erf(|x| / sqrt 2)
    return 0.5f * (x + fmaf(-ax, r, ax));
}
template <typename T> __device__ __forceinline__ float gelu_t(float x) { return gelu_fast(x); }
template <> __device__ __forceinline__ float gelu_t<float>(float x) { return gelu_erf(x); }
// GELU of N (even) values in place: gelu_fast's operations on 2-vectors so that they come out as v_pk_*_f32 (with scalar code the compiler
// prefers v_fmaak_f32 with the coefficient as a literal, one element per instruction), and stage by stage over all N / 2 pairs (not pair by
// pair: a pair's chain is ~14 DEPENDENT operations and the epilogues run with two waves per SIMD).  Bit-identical to gelu_fast.
typedef float dev_f2 __attribute__((ext_vector_type(2)));
template <typename T, int N> __device__ __forceinline__ void gelu_vec(float (&v)[N]) {
    if constexpr (std::is_same<T, float>::value) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = gelu_erf(v[e]);
    } else {
        constexpr int H = N / 2;
        dev_f2 x[H], ax[H], p[H];
#pragma unroll
        for (int h = 0; h < H; ++h) { x[h] = (dev_f2){v[2 * h], v[2 * h + 1]}; ax[h] = __builtin_elementwise_abs(x[h]); }
#pragma unroll
        for (int h = 0; h < H; ++h) p[h] = __builtin_elementwise_fma(ax[h], (dev_f2)(kGeluA6), (dev_f2)(kGeluA5));
#pragma unroll
        for (int h = 0; h < H; ++h) p[h] = __builtin_elementwise_fma(ax[h], p[h], (dev_f2)(kGeluA4));
#pragma unroll
        for (int h = 0; h < H; ++h) p[h] = __builtin_elementwise_fma(ax[h], p[h], (dev_f2)(kGeluA3));
#pragma unroll
        for (int h = 0; h < H; ++h) p[h] = __builtin_elementwise_fma(ax[h], p[h], (dev_f2)(kGeluA2));
#pragma unroll
        for (int h = 0; h < H; ++h) p[h] = __builtin_elementwise_fma(ax[h], p[h], (dev_f2)(kGeluA1));
#pragma unroll
        for (int h = 0; h < H; ++h) p[h] = __builtin_elementwise_fma(ax[h], p[h], (dev_f2)(1.0f));
#pragma unroll
        for (int sq = 0; sq < 4; ++sq)
#pragma unroll
            for (int h = 0; h < H; ++h) p[h] *= p[h];
#pragma unroll
        for (int h = 0; h < H; ++h) { p[h].x = __builtin_amdgcn_rcpf(p[h].x); p[h].y = __builtin_amdgcn_rcpf(p[h].y); }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const dev_f2 g = (dev_f2)(0.5f) * (x[h] + __builtin_elementwise_fma(-ax[h], p[h], ax[h]));
            v[2 * h] = g.x; v[2 * h + 1] = g.y;
        }
    }
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace hcm
