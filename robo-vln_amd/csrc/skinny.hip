// Few-row GEMM (round 5): the linear layers of a step whose row count is small -- every BERT / cross-modal / recurrent projection of a
// one-to-four environment call (the reference's own operating point: hierarchical_trainer.py:1088-1107 steps ONE environment), and the
// recurrent-state GEMMs (M = batch rows) of any batch.
//
//   Y[m][n] = act( sum_k X[m][k] * W[n][k] + bias[n] (+ res[m][n]) )        X [M][ldx], W [N][ldw] (K contiguous), fp32 accumulate
//
// Why a third GEMM kernel.  With M <= a few hundred rows the implicit-GEMM kernels are a LATENCY chain, not a throughput problem: a 64 x 32
// tile of igemm_dma_kernel walks K through a 6-deep LDS ring behind workgroup barriers, 48-144 workgroups on a 256-CU chip, 9.6-9.8 us per
// launch at K = 768 and 21 us at K = 3072 (profiles/r5_igemm_shapes.md, the M = 64 rows) -- a B = 1 step is ~340 such launches.  Here a WAVE owns
// one 16-channel x (16 NJ)-row output tile for the whole K: its operand fragments come straight from L2 / HBM into registers in MFMA operand
// layout (a lane reads 16 contiguous bytes of one row: W rows as the A operand, X rows as the B operand, exactly the roles igemm_dma_kernel
// gives them), UNR k steps per request round, the next round requested before the current one is consumed (two register sets; hipcc counts
// the vmcnt).  No LDS, no barrier, N / 16 x M / 16 independent waves: a K = 768 layer is one or two memory latencies long.
//
// Bit-identical to igemm_dma_kernel / gemm256f_kernel: the same MFMA instruction (v_mfma_f32_16x16x32 for 16-bit T, four v_mfma_f32_16x16x4
// per 16-byte chunk for fp32 with igemm.hip's chunk order), the same k order (one accumulator chain per output element, k ascending), the same
// epilogue operations in the same order (acc + bias, + residual, activation, one rounding).  So a row's value does not depend on which of the
// kernels computed it, i.e. on the batch it was in (tests/test_ops_gpu.py, test_ragged_batch_equals_per_environment_unpadded_calls,
// hcm_refresh_instruction's bitwise contract).
//
// Reference ops replaced: nn.Linear of BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput (seq2seq_highlevel_cma.py:192-195),
// the cross-modal projections (transformer.py), the LSTM / GRU weight products of RNNStateEncoder (state_encoder.py:52-81).
#include <cstdlib>
#include <type_traits>
#include "kernels.h"
#include "dev.h"

namespace hcm {

namespace {

typedef float sk_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sk_bf16x8 __attribute__((ext_vector_type(8)));

struct SkinnyDev {
    const char* x; const char* w; const float* bias; const char* res; char* y;
    int M, N, K, ldx, ldw, ldy, ldr, act, out_f32, res_f32;
    long long g_x, g_w, g_b, g_y;      // grouped launch (blockIdx.z): element offsets per group
    unsigned x_bytes, w_bytes;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sk_make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 sk_ld16(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
}

template <typename T> struct SkMma;
template <> struct SkMma<bf16> {
    static __device__ __forceinline__ void run(sk_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sk_bf16x8, a), __builtin_bit_cast(sk_bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct SkMma<f16> {
    static __device__ __forceinline__ void run(sk_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), acc, 0, 0, 0);
    }
};
template <> struct SkMma<float> {      // igemm.hip's Mma<float>: the lane's four k values one after the other
    static __device__ __forceinline__ void run(sk_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

constexpr unsigned kOob = 0x80000000u;      // an offset beyond num_records: the buffer load returns zeros

// grid (ceil(N / 16), ceil(row fragments / (waves NJ)), groups), 64 x waves threads; wave w of block (bx, by) owns channels [16 bx, +16) x
// row fragments (by * waves + w) * NJ .. + NJ
template <typename T, int NJ, int UNR>
__global__ __launch_bounds__(512) void skinny_kernel(SkinnyDev p) {
    constexpr int ESZ = (int)sizeof(T);
    constexpr int CH = 16 / ESZ;               // elements per 16-byte chunk
    constexpr int KS = 4 * CH;                 // k values per step: 32 (16-bit), 16 (fp32)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwv = blockDim.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int j0 = (blockIdx.y * nwv + wave) * NJ;           // first row fragment of this wave
    if (j0 * 16 >= p.M) return;
    const int grp = blockIdx.z;
    const char* xb = p.x + (size_t)grp * p.g_x * ESZ;
    const char* wb = p.w + (size_t)grp * p.g_w * ESZ;
    const __amdgpu_buffer_rsrc_t rx = sk_make_rsrc(xb, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rw = sk_make_rsrc(wb, p.w_bytes);

    // the lane's rows: W row n0 + fr, X rows (j0 + j) * 16 + fr; its 16 bytes of a k step start at k = step * KS + fg * CH
    const int nrow = n0 + fr;
    const unsigned wrow = nrow < p.N ? (unsigned)((size_t)nrow * p.ldw * ESZ) : kOob;
    unsigned xrow[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int m = (j0 + j) * 16 + fr;
        xrow[j] = m < p.M ? (unsigned)((size_t)m * p.ldx * ESZ) : kOob;
    }
    const int nks = (p.K + KS - 1) / KS;
    const int nr = (nks + UNR - 1) / UNR;

    sk_f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = (sk_f32x4){0.f, 0.f, 0.f, 0.f};

    uint4 wa[UNR], xa[UNR][NJ], wq[UNR], xq[UNR][NJ];
    auto request = [&](uint4 (&wf)[UNR], uint4 (&xf)[UNR][NJ], int r) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int k = (r * UNR + u) * KS + fg * CH;
            const bool ok = k < p.K;                         // K % CH == 0: a chunk is all-valid or all-invalid
            const unsigned kb = (unsigned)(k * ESZ);
            wf[u] = sk_ld16(rw, (ok && wrow != kOob) ? wrow + kb : kOob);
#pragma unroll
            for (int j = 0; j < NJ; ++j) xf[u][j] = sk_ld16(rx, (ok && xrow[j] != kOob) ? xrow[j] + kb : kOob);
        }
    };
    // (k steps past K in the last round multiply zero-filled fragments: acc + 0, exact -- an accumulator that starts at +0 is never -0)
    auto consume = [&](const uint4 (&wf)[UNR], const uint4 (&xf)[UNR][NJ]) {
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int j = 0; j < NJ; ++j) SkMma<T>::run(acc[j], wf[u], xf[u][j]);
    };
    // straight-line pairs (request the next round, consume the current one): a branch between a request and the consume behind it would make
    // hipcc wait for vmcnt(0) at the join, i.e. for the round it has just requested
    request(wa, xa, 0);
    int r = 0;
    for (; r + 2 < nr; r += 2) {
        request(wq, xq, r + 1);
        __builtin_amdgcn_sched_barrier(0);      // (all of a round's requests go out before the MFMAs in front of them: hipcc would sink them)
        consume(wa, xa);
        __builtin_amdgcn_sched_barrier(0);
        request(wa, xa, r + 2);
        __builtin_amdgcn_sched_barrier(0);
        consume(wq, xq);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (r + 1 < nr) {
        request(wq, xq, r + 1);
        __builtin_amdgcn_sched_barrier(0);
        consume(wa, xa);
        consume(wq, xq);
    } else {
        consume(wa, xa);
    }

    // epilogue: the lane holds channels n .. n + 3 of row m (fragment j): acc + bias, + residual, activation, one rounding
    const int n = n0 + fg * 4;
    if (n + 4 > p.N) return;                                  // N % 4 == 0 (launcher)
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + (size_t)grp * p.g_b + n);
        b4[0] = b.x; b4[1] = b.y; b4[2] = b.z; b4[3] = b.w;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int m = (j0 + j) * 16 + fr;
        if (m >= p.M) continue;
        float v[4] = {acc[j][0] + b4[0], acc[j][1] + b4[1], acc[j][2] + b4[2], acc[j][3] + b4[3]};
        if (p.res) {
            float r4[4];
            if (p.res_f32 || std::is_same<T, float>::value) {
                const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + (size_t)grp * p.g_y + (size_t)m * p.ldr + n);
                r4[0] = r.x; r4[1] = r.y; r4[2] = r.z; r4[3] = r.w;
            } else {
                const T* rp = reinterpret_cast<const T*>(p.res) + (size_t)grp * p.g_y + (size_t)m * p.ldr + n;
                const uint2 rr = *reinterpret_cast<const uint2*>(rp);
                T t4[4];
                *reinterpret_cast<uint2*>(t4) = rr;
#pragma unroll
                for (int e = 0; e < 4; ++e) r4[e] = Tr<T>::ld(&t4[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r4[e];
        }
        if (p.act == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = relu_f(v[e]);
        } else if (p.act == ACT_GELU) {
            gelu_vec<T, 4>(v);
        }
        if (p.out_f32 || std::is_same<T, float>::value) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (size_t)grp * p.g_y + (size_t)m * p.ldy + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            T o4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) Tr<T>::st(&o4[e], v[e]);
            *reinterpret_cast<uint2*>(reinterpret_cast<T*>(p.y) + (size_t)grp * p.g_y + (size_t)m * p.ldy + n) = *reinterpret_cast<const uint2*>(o4);
        }
    }
}

// Cost model (measured: tools/skinny_bench.py, profiles/r5_skinny_shapes.md).  What bounds the kernel is the CUs' vector-memory path -- a 16-row x
// 64-byte fragment load costs it ~58 cycles -- and the dispatcher spreads one-wave workgroups evenly over the 256 CUs:
//     t ~= 2.6 us + 0.0275 us x (loads per wave) x ceil(waves / 256)
// (BERT's FFN2 at 80 rows: 240 waves x 192 loads -> 7.5 us measured, 7.9 modelled; the 64 x 32 implicit-GEMM tile it replaces requests
// (64 + 32) rows x K bytes / 1 KB = 576 LDS-DMA pieces from ONE workgroup at the same ~58 cycles each: 18.3 us measured, 18.6 modelled).
// NJ row fragments per wave share the wave's W fragments ((1 + NJ) loads per k step for NJ tiles) but make fewer, longer waves: measured, one
// fragment per wave is the fastest up to 80 rows and two up to 160; beyond that the implicit-GEMM tiles win except for K >= 2048.  fp32 (four
// quarter-rate MFMAs per fragment pair: the accumulator chain is as long as the loads) stays at one fragment and 80 rows.
struct SkPlan { int nj; double loads; };
static SkPlan sk_plan(int M, int cols16, int K, int esz) {
    const int frags = (M + 15) / 16, steps = (K * esz + 63) / 64;
    const int nj = frags <= 5 ? 1 : 2, unr = nj == 1 ? 12 : 8;
    const long waves = (long)((frags + nj - 1) / nj) * cols16;
    return SkPlan{nj, (double)(1 + nj) * ((steps + unr - 1) / unr * unr) * (double)((waves + 255) / 256)};
}

template <typename T>
hipError_t launch_t(const SkinnyDev& d, int groups, hipStream_t s) {
    const int frags = (d.M + 15) / 16, cols16 = (d.N + 15) / 16;
    static const int nj_env = dev_env("HCM_SKINNY_NJ") ? atoi(dev_env("HCM_SKINNY_NJ")) : 0;       // development build: the A/B
    const int nj = nj_env >= 1 && nj_env <= 5 ? nj_env : sk_plan(d.M, cols16 * groups, d.K, (int)sizeof(T)).nj;
    const int wtiles = (frags + nj - 1) / nj;
    // ONE wave per workgroup: the dispatcher places workgroups, not waves, and the waves must spread over all 256 CUs (five waves of a workgroup
    // on one CU: 8.8 us for BERT's QKV at 80 rows against 6.2 with one-wave workgroups, 6.9 on the implicit-GEMM tiles; HCM_SKINNY_WAVES,
    // development build: the A/B)
    static const int waves_env = dev_env("HCM_SKINNY_WAVES") ? atoi(dev_env("HCM_SKINNY_WAVES")) : 1;
    const int waves = wtiles < waves_env ? wtiles : (waves_env < 1 ? 1 : waves_env > 8 ? 8 : waves_env);
    const dim3 grid(cols16, (wtiles + waves - 1) / waves, groups), block(64 * waves);
    const void* fn = nj == 1 ? reinterpret_cast<const void*>(skinny_kernel<T, 1, 12>)
                   : nj == 2 ? reinterpret_cast<const void*>(skinny_kernel<T, 2, 8>)
                   : nj == 3 ? reinterpret_cast<const void*>(skinny_kernel<T, 3, 6>)
                   : nj == 4 ? reinterpret_cast<const void*>(skinny_kernel<T, 4, 4>) : reinterpret_cast<const void*>(skinny_kernel<T, 5, 4>);
    SkinnyDev dd = d;
    void* args[] = {&dd};
    return hipLaunchKernel(fn, grid, block, args, 0, s);
}

}  // namespace

// layout constraints (an `impl = 3` launch takes any row count); skinny_applicable adds the row-count rule of launch_igemm's own choice
static bool skinny_valid(const IGemm& g, int dt) {
    if (dt != DT_BF16 && dt != DT_F16 && dt != DT_F32) return false;
    if (g.KH != 1 || g.KW != 1 || g.stride != 1 || g.pad != 0 || (g.stride_w > 0 && g.stride_w != 1)) return false;     // rows at a fixed pixel stride
    if (g.gn_gamma || g.cs_part || g.hpool || g.x_src_dt >= 0 || g.gi_stats || g.ln_s || g.ln_part_out || g.ln_part_in || g.rln_stats) return false;
    if (dt == DT_F32 && g.res_f32) return false;
    const int CH = dt == DT_F32 ? 4 : 8;
    const int Kp = g.Kp ? g.Kp : g.K, ldx = g.xC ? g.xC : g.Cin, ldy = g.ldy ? g.ldy : g.N, ldr = g.ldr ? g.ldr : g.N;
    if (g.M < 1 || g.N < 4 || g.K < CH) return false;
    if ((g.K % CH) || (Kp % CH) || (ldx % CH) || (g.N % 4) || (ldy % 4) || (g.res && (ldr % 4))) return false;
    const size_t esz = dt == DT_F32 ? 4 : 2;
    if (((size_t)(g.M - 1) * ldx + g.K) * esz >= 0x7FFFFFF0ull || (size_t)g.N * Kp * esz >= 0x7FFFFFF0ull) return false;      // 32-bit offsets + the sentinel
    if (g.groups > 1 && ((g.g_x % CH) || (g.g_w % CH) || (g.g_b % 4) || (g.g_y % 4))) return false;
    return true;
}
bool skinny_applicable(const IGemm& g, int dt) {
    if (!skinny_valid(g, dt)) return false;
    static const int max_rows = dev_env("HCM_SKINNY_MAX_ROWS") ? atoi(dev_env("HCM_SKINNY_MAX_ROWS")) : 160;
    if ((long)g.M > (dt == DT_F32 && max_rows > 80 ? 80 : max_rows)) return false;
    // against the tile launch_igemm would take for so few rows (64 x 32, LDS-DMA ring): its one workgroup per tile requests (64 + 32) rows x K
    // bytes in 1 KB pieces at the same ~58 cycles each, whatever the row count, and carries ~0.2 us more of fixed cost (prologue, barriers)
    const int esz = dt == DT_F32 ? 4 : 2;
    const int groups = g.groups > 1 ? g.groups : 1;
    const double tiles = 96.0 * g.K * esz / 1024.0 * (double)(((long)(g.M + 63) / 64 * ((g.N + 31) / 32) * groups + 255) / 256) + 7.0;
    return sk_plan(g.M, (g.N + 15) / 16 * groups, g.K, esz).loads <= 0.97 * tiles;
}

hipError_t launch_skinny(const IGemm& g, int dt, hipStream_t s) {
    if (!skinny_valid(g, dt)) return hipErrorInvalidValue;
    SkinnyDev d;
    d.x = (const char*)g.x; d.w = (const char*)g.w; d.bias = g.bias; d.res = (const char*)g.res; d.y = (char*)g.y;
    d.M = g.M; d.N = g.N; d.K = g.K; d.ldx = g.xC ? g.xC : g.Cin; d.ldw = g.Kp ? g.Kp : g.K; d.ldy = g.ldy ? g.ldy : g.N; d.ldr = g.ldr ? g.ldr : g.N;
    d.act = g.act; d.out_f32 = g.out_f32; d.res_f32 = g.res_f32;
    const int groups = g.groups > 1 ? g.groups : 1;
    d.g_x = groups > 1 ? g.g_x : 0; d.g_w = groups > 1 ? g.g_w : 0; d.g_b = groups > 1 ? g.g_b : 0; d.g_y = groups > 1 ? g.g_y : 0;
    const size_t esz = dt == DT_F32 ? 4 : 2;
    d.x_bytes = (unsigned)(((size_t)(g.M - 1) * d.ldx + g.K) * esz);
    d.w_bytes = (unsigned)((size_t)g.N * d.ldw * esz);
    if (dt == DT_BF16) return launch_t<bf16>(d, groups, s);
    if (dt == DT_F16) return launch_t<f16>(d, groups, s);
    return launch_t<float>(d, groups, s);
}

}  // namespace hcm
