// BERT self-attention block tail in ONE launch (round 5): softmax(Q K^T / 8) V over the 12 heads of a sample, the attention-output projection
// (768 -> 768), bias + residual, and the post-attention LayerNorm -- what bert() otherwise runs as attention_mfma_kernel + igemm_dma_kernel +
// layernorm_vec_kernel (BertSelfAttention / BertSelfOutput of the `BertModel` call, seq2seq_highlevel_cma.py:192-195; no attention mask).
//
// Shape of the launch: a workgroup owns up to NRT 16-row tiles of ONE sample (L = 80: two workgroups per sample, 48 + 32 rows), 8 waves.
//   phase A  attention.  (head, query tile) tasks are dealt to the waves and run WITHOUT workgroup barriers: Q and K fragments come straight
//            from global memory in MFMA operand layout (a K row is an A-operand row of S^T = K Q^T), V goes through a wave-private 4.6 KB LDS
//            transposition area, 32 keys at a time.  The context rows land in LDS as T, in the 128-byte-row swizzled layout the implicit-GEMM
//            kernels use for their activation tiles (one 64-column head = one K segment).
//   phase B  projection.  A wave owns 96 output columns of all rows: its weight fragments are shared with NO other wave, so they are loaded
//            straight from L2 into registers (three K steps in flight) and the context fragments come from LDS; 24 K steps of
//            v_mfma_f32_16x16x32, weights as the A operand exactly as igemm_dma_kernel has them.
//   phase C  bias + residual (+ one rounding to T) into an LDS image of the rows, then LayerNorm with layernorm_vec_kernel's lane mapping and
//            reduction order, 16-byte stores of the normalised rows.
// Every f32 operation is the one the three launches perform, in the same order: BIT-IDENTICAL to them (tests/test_fusion_toggles_gpu.py), and a
// row's value does not depend on the batch it is in.  F32S = true: the bf16 mode's f32 residual stream (sum kept in f32, LayerNorm as
// layernorm_f32in_kernel with its two outputs).
// What bounds it: a workgroup ingests all 1.18 MB of W_o through its CU's one texture path (~9 us at 64 B / clk); the MFMA work of 48 rows is
// ~7 us beside it.  64-128 workgroups: the launch leaves half the chip to the other encoder chains of the step.
#include <cstdlib>
#include "kernels.h"
#include "dev.h"

namespace hcm {

namespace {

typedef float bb_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bb_bf16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct BbMma;
template <> struct BbMma<bf16> {
    static __device__ __forceinline__ bb_f32x4 run(const uint4& a, const uint4& b, bb_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bb_bf16x8, a), __builtin_bit_cast(bb_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct BbMma<f16> {
    static __device__ __forceinline__ bb_f32x4 run(const uint4& a, const uint4& b, bb_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    }
};
template <typename T> __device__ __forceinline__ uint32_t bb_pack2(float a, float b) {
    T t[2];
    Tr<T>::st(&t[0], a);
    Tr<T>::st(&t[1], b);
    return (uint32_t)t[0].v | ((uint32_t)t[1].v << 16);
}

constexpr int kD = 768, kHeads = 12, kDh = 64, kWaves = 8, kColsPerWave = kD / kWaves, kTN = kColsPerWave / 16;   // 96 columns = 6 tiles per wave
constexpr int kMaxKT = 6;                   // 16-key tiles of the score registers: L <= 96
constexpr int kVtStride = 36;               // elements per V^T row of a 32-key block (+4: the 8-byte reads stay spread over the banks)
constexpr int kVtBytes = kDh * kVtStride * 2;

struct BertBlockDev {
    const void* qkv;            // [B * L][ldq]: Q | K | V, 768 each
    const void* wo;             // W_o in fragment order (launch_pack_frag of the [768][768] K-contiguous weight)
    const float* bo;
    const void* res;            // residual stream, T (F32S: unused)
    const float* res32;         // F32S: the f32 stream
    const float* gamma;
    const float* beta;
    void* y;                    // LayerNorm output, T
    float* y32;                 // F32S: the f32 stream out
    const int* klens;
    int B, L, ldq, wps;
    float eps;
    unsigned long long* prof;   // development build, HCM_BB_PROF_PTR: [workgroup][wave][4] s_memtime stamps (start, end of A, end of B, end)
};
__device__ __forceinline__ unsigned long long bb_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <typename T, int NRT, bool F32S>
__global__ __launch_bounds__(64 * kWaves) void bert_attn_block_kernel(BertBlockDev p) {
    constexpr int ROWS = NRT * 16;
    constexpr int SEG = ROWS * 128;                       // bytes of one 64-column K segment of the context tile
    constexpr int CTX_BYTES = kHeads * SEG;
    extern __shared__ __attribute__((aligned(16))) char smem_bb[];
    char* ctx = smem_bb;
    // XCD-friendly placement (speed only): the workgroups of one sample share its K / V rows -- keep them on one XCD's L2 (block b runs on XCD b % 8)
    const int per = 8 * p.wps;
    const int grp = blockIdx.x / per, within = blockIdx.x % per;
    const int b = grp * 8 + (within & 7);
    const int wsub = within >> 3;
    if (b >= p.B) return;
    const int L = p.L;
    const int QT = (L + 15) >> 4;
    const int rt0 = wsub * NRT;
    if (rt0 >= QT) return;
    const int nrt = QT - rt0 < NRT ? QT - rt0 : NRT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const T* qkv = reinterpret_cast<const T*>(p.qkv);
    const size_t row0 = (size_t)b * L;

    int Lk = L;
    if (p.klens) { Lk = p.klens[b]; Lk = Lk < 1 ? 1 : Lk > L ? L : Lk; }
    const int KT2 = (Lk + 31) >> 5;                       // 32-key MFMA steps

    unsigned long long pt0 = 0, pt1 = 0, pt2 = 0;
    if (p.prof) pt0 = bb_now();
    // L2 warm-up of W_o (speed only).  Phase B streams 1.18 MB of weights that nobody has touched for a whole step: every K step would be an HBM
    // round trip (~2 us) with three K steps of prefetch in flight, i.e. the projection would run at the miss latency.  The workgroups that share
    // an XCD (block b runs on XCD b % 8) split the matrix between them and touch one dword per 128-byte line NOW, ~10 us before phase B needs it.
    unsigned warm = 0;
    {
        const int nx = (gridDim.x - (blockIdx.x & 7) + 7) >> 3;              // workgroups on this XCD, this one is number blockIdx.x >> 3
        constexpr int LINES = kD * kD * 2 / 128;
        const char* wbytes = reinterpret_cast<const char*>(p.wo);
        for (int ln = ((blockIdx.x >> 3) * (64 * kWaves) + tid); ln < LINES; ln += nx * 64 * kWaves)
            warm ^= __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(wbytes + (size_t)ln * 128));
    }
    // ---------------------------------------------------------------- phase A: attention, one HEAD per wave and round, no workgroup barrier
    // Every global load is a row-contiguous one (8 rows x 128 B per wave instruction: 8 cache-line look-ups instead of the 64 a fragment-shaped
    // load costs the CU's one texture path) and reaches MFMA operand layout through the wave's own LDS area: Q tiles and 32-key K blocks as
    // swizzled 128-byte rows (read back as fragments), V blocks transposed ([d][key]).  LDS operations of one wave execute in order; the waits
    // only keep the compiler from moving reads over writes.
    char* wl = smem_bb + CTX_BYTES + wave * kVtBytes;
    const int lrow = lane >> 3, lch = lane & 7;
    for (int h = wave; h < kHeads; h += kWaves) {
        uint4 qreg[NRT][2], kreg[kMaxKT / 2][4];
#pragma unroll
        for (int jt = 0; jt < NRT; ++jt)
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                int qrow = (rt0 + jt) * 16 + ps * 8 + lrow;
                if (qrow >= L) qrow = L - 1;                  // rows past L compute garbage that is never stored
                qreg[jt][ps] = make_uint4(0u, 0u, 0u, 0u);
                if (jt < nrt) qreg[jt][ps] = *reinterpret_cast<const uint4*>(qkv + (row0 + qrow) * p.ldq + h * kDh + lch * 8);
            }
#pragma unroll
        for (int c2 = 0; c2 < kMaxKT / 2; ++c2)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                kreg[c2][ps] = make_uint4(0u, 0u, 0u, 0u);
                const int key = c2 * 32 + ps * 8 + lrow;     // rows >= Lk are zero
                if (c2 < KT2 && key < Lk) kreg[c2][ps] = *reinterpret_cast<const uint4*>(qkv + (row0 + key) * p.ldq + kD + h * kDh + lch * 8);
            }
        // Q tiles -> B-operand fragments
        uint4 qf[NRT][2];
#pragma unroll
        for (int jt = 0; jt < NRT; ++jt) {
            qf[jt][0] = qf[jt][1] = make_uint4(0u, 0u, 0u, 0u);
            if (jt < nrt) {
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int r = ps * 8 + lrow;
                    *reinterpret_cast<uint4*>(wl + r * 128 + ((lch ^ (r & 7)) << 4)) = qreg[jt][ps];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                qf[jt][0] = *reinterpret_cast<const uint4*>(wl + fr * 128 + ((fg ^ (fr & 7)) << 4));
                qf[jt][1] = *reinterpret_cast<const uint4*>(wl + fr * 128 + (((4 + fg) ^ (fr & 7)) << 4));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        // K blocks -> A-operand fragments of the 16-key tiles
        uint4 kf[kMaxKT][2];
#pragma unroll
        for (int c2 = 0; c2 < kMaxKT / 2; ++c2) {
            kf[2 * c2][0] = kf[2 * c2][1] = kf[2 * c2 + 1][0] = kf[2 * c2 + 1][1] = make_uint4(0u, 0u, 0u, 0u);
            if (c2 < KT2) {
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int r = ps * 8 + lrow;
                    *reinterpret_cast<uint4*>(wl + r * 128 + ((lch ^ (r & 7)) << 4)) = kreg[c2][ps];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const int r = tl * 16 + fr;
                    kf[2 * c2 + tl][0] = *reinterpret_cast<const uint4*>(wl + r * 128 + ((fg ^ (r & 7)) << 4));
                    kf[2 * c2 + tl][1] = *reinterpret_cast<const uint4*>(wl + r * 128 + (((4 + fg) ^ (r & 7)) << 4));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        // V rows of every 32-key block, requested now (the K registers are free again): their round trip runs beside Q K^T and the softmax
        uint4 vreg[kMaxKT / 2][4];
#pragma unroll
        for (int c2 = 0; c2 < kMaxKT / 2; ++c2)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                vreg[c2][ps] = make_uint4(0u, 0u, 0u, 0u);
                const int key = c2 * 32 + ps * 8 + lrow;
                if (c2 < KT2 && key < Lk) vreg[c2][ps] = *reinterpret_cast<const uint4*>(qkv + (row0 + key) * p.ldq + 2 * kD + h * kDh + lch * 8);
            }
        // scores and softmax per query tile (the arithmetic of attention_mfma_kernel, operation for operation)
        uint4 pa[NRT][kMaxKT / 2];
        float inv[NRT];
#pragma unroll
        for (int jt = 0; jt < NRT; ++jt) {
            inv[jt] = 0.f;
#pragma unroll
            for (int c2 = 0; c2 < kMaxKT / 2; ++c2) pa[jt][c2] = make_uint4(0u, 0u, 0u, 0u);
            if (jt >= nrt) continue;
            bb_f32x4 sc[kMaxKT];
            float mx = -3.0e38f;
#pragma unroll
            for (int t = 0; t < kMaxKT; ++t) {
                sc[t] = (bb_f32x4){0.f, 0.f, 0.f, 0.f};
                if (t < 2 * KT2) {
                    sc[t] = BbMma<T>::run(kf[t][0], qf[jt][0], sc[t]);
                    sc[t] = BbMma<T>::run(kf[t][1], qf[jt][1], sc[t]);
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
            for (int t = 0; t < kMaxKT; ++t)
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
            inv[jt] = 1.0f / sum;               // per query = per (lane & 15)
#pragma unroll
            for (int c2 = 0; c2 < kMaxKT / 2; ++c2)
                if (c2 < KT2) {
                    pa[jt][c2].x = bb_pack2<T>(sc[2 * c2][0], sc[2 * c2][1]);
                    pa[jt][c2].y = bb_pack2<T>(sc[2 * c2][2], sc[2 * c2][3]);
                    pa[jt][c2].z = bb_pack2<T>(sc[2 * c2 + 1][0], sc[2 * c2 + 1][1]);
                    pa[jt][c2].w = bb_pack2<T>(sc[2 * c2 + 1][2], sc[2 * c2 + 1][3]);
                }
        }
        // P V, one 32-key block of V^T at a time, shared by the query tiles
        bb_f32x4 o[NRT][4];
#pragma unroll
        for (int jt = 0; jt < NRT; ++jt)
#pragma unroll
            for (int dtile = 0; dtile < 4; ++dtile) o[jt][dtile] = (bb_f32x4){0.f, 0.f, 0.f, 0.f};
        T* Vt = reinterpret_cast<T*>(wl);
#pragma unroll
        for (int c2 = 0; c2 < kMaxKT / 2; ++c2)
            if (c2 < KT2) {
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const uint32_t w[4] = {vreg[c2][ps].x, vreg[c2][ps].y, vreg[c2][ps].z, vreg[c2][ps].w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        T t;
                        t.v = (uint16_t)(w[j >> 1] >> ((j & 1) * 16));
                        Vt[(lch * 8 + j) * kVtStride + ps * 8 + lrow] = t;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                uint4 vb[4];
#pragma unroll
                for (int dtile = 0; dtile < 4; ++dtile) {
                    const T* vr = Vt + (dtile * 16 + fr) * kVtStride + fg * 4;
                    const uint2 lo = *reinterpret_cast<const uint2*>(vr);
                    const uint2 hi = *reinterpret_cast<const uint2*>(vr + 16);
                    vb[dtile] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int jt = 0; jt < NRT; ++jt)
                    if (jt < nrt) {
#pragma unroll
                        for (int dtile = 0; dtile < 4; ++dtile) o[jt][dtile] = BbMma<T>::run(pa[jt][c2], vb[dtile], o[jt][dtile]);
                    }
            }
        // context rows into the swizzled K segment of this head: row r, 16-byte chunk ch -> ch ^ (r & 7); rows of o are queries (fg*4 + e)
        char* seg = ctx + h * SEG;
#pragma unroll
        for (int jt = 0; jt < NRT; ++jt)
            if (jt < nrt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float inv_q = __shfl(inv[jt], fg * 4 + e, 64);
                    const int r = jt * 16 + fg * 4 + e;
#pragma unroll
                    for (int dtile = 0; dtile < 4; ++dtile) {
                        T t;
                        Tr<T>::st(&t, o[jt][dtile][e] * inv_q);
                        const int ch = dtile * 2 + (fr >> 3);
                        *reinterpret_cast<T*>(seg + r * 128 + ((ch ^ (r & 7)) << 4) + (fr & 7) * 2) = t;
                    }
                }
            }
    }

    // ---------------------------------------------------------------- phase B: projection, 96 columns per wave
    // A wave's weight fragments are shared with no other wave: they come straight from L2 into registers, from the FRAGMENT-ORDER copy of W_o
    // (launch_pack_frag: one fragment = 1 KB contiguous = 8 cache-line look-ups; the [N][K] layout would cost 64 per fragment and the CU's texture
    // path, one look-up per clock, would bound the whole kernel), three K steps in flight.  Context fragments from LDS.
    const T* wf = reinterpret_cast<const T*>(p.wo) + ((size_t)wave * kTN * 64 + lane) * 8;
    constexpr int KS = kD / 32;                            // 24 K steps
    constexpr int PF = 3;                                  // weight fragment sets in flight
    uint4 wa[PF][kTN];
    auto load_w = [&](int ks, uint4 (&dst)[kTN]) {
#pragma unroll
        for (int i = 0; i < kTN; ++i) dst[i] = *reinterpret_cast<const uint4*>(wf + ((size_t)ks * (kD / 16) + i) * 64 * 8);
    };
    load_w(0, wa[0]);
    load_w(1, wa[1]);
    bb_f32x4 acc[kTN][NRT];
#pragma unroll
    for (int i = 0; i < kTN; ++i)
#pragma unroll
        for (int j = 0; j < NRT; ++j) acc[i][j] = (bb_f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.prof) pt1 = bb_now();
    __syncthreads();                                       // the context tile is complete
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + 2 < KS) load_w(ks + 2, wa[(ks + 2) % PF]);
        uint4 xb[NRT];
        const char* seg = ctx + (ks >> 1) * SEG;
        const int chunk = (ks & 1) * 4 + fg;
#pragma unroll
        for (int j = 0; j < NRT; ++j) {
            const int r = j * 16 + fr;
            xb[j] = *reinterpret_cast<const uint4*>(seg + r * 128 + ((chunk ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < kTN; ++i)
#pragma unroll
            for (int j = 0; j < NRT; ++j)
                if (j < nrt) acc[i][j] = BbMma<T>::run(wa[ks % PF][i], xb[j], acc[i][j]);
    }
    if (p.prof) pt2 = bb_now();
    __syncthreads();                                       // every wave is done reading the context tile: the row image takes its place

    // ---------------------------------------------------------------- phase C: acc as an f32 row image -> + bias + residual -> LayerNorm
    // Bias and residual are added in the ROW pass (row-contiguous 16-byte loads, requested here, before the image is even written: they are cold
    // lines and their round trip is the longest thing left) -- (acc + bias) + residual, rounded once to T, exactly the projection launch's epilogue --
    // and the LayerNorm that follows is layernorm_vec_kernel's (layernorm_f32in_kernel's for F32S) lane mapping and reduction order.
    constexpr int IMG_STRIDE = kD * 4 + 16;                // bytes per image row
    constexpr int CW = F32S ? 4 : 8;                       // elements per lane chunk of the row pass
    constexpr int CPL = kD / CW / 32;                      // chunks per lane: 3 of 8 (T) / 6 of 4 (F32S)
    constexpr int NPASS = ROWS / (2 * kWaves);             // 16 rows per pass
    char* img = smem_bb;
    const int sub = tid & 31;
    float gm[CPL][CW], bt[CPL][CW], bs[CPL][CW];
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int q = 0; q < CW / 4; ++q) {
            const int c = (i * 32 + sub) * CW + 4 * q;
            const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + c), b4 = *reinterpret_cast<const float4*>(p.beta + c);
            const float4 s4 = *reinterpret_cast<const float4*>(p.bo + c);
            gm[i][4 * q] = g4.x; gm[i][4 * q + 1] = g4.y; gm[i][4 * q + 2] = g4.z; gm[i][4 * q + 3] = g4.w;
            bt[i][4 * q] = b4.x; bt[i][4 * q + 1] = b4.y; bt[i][4 * q + 2] = b4.z; bt[i][4 * q + 3] = b4.w;
            bs[i][4 * q] = s4.x; bs[i][4 * q + 1] = s4.y; bs[i][4 * q + 2] = s4.z; bs[i][4 * q + 3] = s4.w;
        }
    uint4 rres[NPASS][CPL];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int rl = rt0 * 16 + ps * 2 * kWaves + (tid >> 5);
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            rres[ps][i] = make_uint4(0u, 0u, 0u, 0u);
            if (ps * 2 * kWaves + (tid >> 5) < nrt * 16 && rl < L) {
                if constexpr (F32S) rres[ps][i] = *reinterpret_cast<const uint4*>(p.res32 + (row0 + rl) * kD + (i * 32 + sub) * 4);
                else rres[ps][i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.res) + (row0 + rl) * kD + (i * 32 + sub) * 8);
            }
        }
    }
    const int ncol0 = wave * kColsPerWave;
#pragma unroll
    for (int i = 0; i < kTN; ++i) {
        const int n = ncol0 + i * 16 + fg * 4;
#pragma unroll
        for (int j = 0; j < NRT; ++j)
            if (j < nrt)
                *reinterpret_cast<float4*>(img + (j * 16 + fr) * IMG_STRIDE + n * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int ml = ps * 2 * kWaves + (tid >> 5);
        const int rl = rt0 * 16 + ml;
        if (ml >= nrt * 16 || rl >= L) continue;
        const char* irow = img + ml * IMG_STRIDE;
        float v[CPL][CW];
        float sum = 0.f;
        if constexpr (!F32S) {
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int c = (i * 32 + sub) * 8;
                const float4 a0 = *reinterpret_cast<const float4*>(irow + c * 4), a1 = *reinterpret_cast<const float4*>(irow + c * 4 + 16);
                float s8[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, rr[8];
                cvt_chunk<T>(rres[ps][i], rr);
#pragma unroll
                for (int j = 0; j < 8; ++j) s8[j] = (s8[j] + bs[i][j]) + rr[j];
                cvt_chunk<T>(pack_chunk<T>(s8), v[i]);     // the projection's one rounding to T, then the LayerNorm reads that value
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += v[i][j];
            }
        } else {
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const float4 a = *reinterpret_cast<const float4*>(irow + (i * 32 + sub) * 16);
                const float4 r4 = __builtin_bit_cast(float4, rres[ps][i]);
                v[i][0] = (a.x + bs[i][0]) + r4.x; v[i][1] = (a.y + bs[i][1]) + r4.y; v[i][2] = (a.z + bs[i][2]) + r4.z; v[i][3] = (a.w + bs[i][3]) + r4.w;
                sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum * (1.0f / kD);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i)
#pragma unroll
            for (int j = 0; j < CW; ++j) { const float d = v[i][j] - mean; sq += d * d; }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        const float rstd = rsqrtf(sq * (1.0f / kD) + p.eps);
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = (i * 32 + sub) * CW;
            float o[CW];
#pragma unroll
            for (int j = 0; j < CW; ++j) o[j] = (v[i][j] - mean) * rstd * gm[i][j] + bt[i][j];
            if constexpr (!F32S) {
                st_chunk(reinterpret_cast<T*>(p.y) + (row0 + rl) * kD + c, o);
            } else {
                st_chunk(p.y32 + (row0 + rl) * kD + c, o);
                T o4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) Tr<T>::st(&o4[j], o[j]);
                *reinterpret_cast<uint2*>(reinterpret_cast<T*>(p.y) + (row0 + rl) * kD + c) = *reinterpret_cast<const uint2*>(o4);
            }
        }
    }
    asm volatile("" ::"v"(warm));                          // (the warm-up loads have to be issued; their values are not used)
    if (p.prof && lane == 0) {
        unsigned long long* o = p.prof + ((size_t)blockIdx.x * kWaves + wave) * 4;
        o[0] = pt0; o[1] = pt1; o[2] = pt2; o[3] = bb_now();
    }
}

template <typename T, int NRT, bool F32S> constexpr size_t bb_lds_bytes() {
    constexpr size_t a = (size_t)kHeads * NRT * 16 * 128 + (size_t)kWaves * kVtBytes;
    constexpr size_t c = (size_t)NRT * 16 * (kD * 4 + 16);
    return a > c ? a : c;
}

// W [N][K] (K-contiguous) -> fragment order: the 1 KB a wave loads for (K step ks, 16-column tile ct) is contiguous, lane (fg = l >> 4, fr = l & 15)
// holding W[ct * 16 + fr][ks * 32 + fg * 8 .. + 8): out[((ks * (N / 16) + ct) * 64 + l) * 8 + e]
template <typename T>
__global__ void pack_frag_kernel(const T* __restrict__ w, T* __restrict__ out, int N, int K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte chunk each
    if (idx >= N * K / 8) return;
    const int l = idx & 63, frag = idx >> 6;
    const int ct = frag % (N / 16), ks = frag / (N / 16);
    const int fr = l & 15, fg = l >> 4;
    reinterpret_cast<uint4*>(out)[idx] = *reinterpret_cast<const uint4*>(w + (size_t)(ct * 16 + fr) * K + ks * 32 + fg * 8);
}

}  // namespace

hipError_t launch_pack_frag(const void* w, void* out, int dt, int N, int K, hipStream_t s) {
    if ((dt != DT_F16 && dt != DT_BF16) || N % 16 || K % 32 || N < 16 || K < 32) return hipErrorInvalidValue;
    const int chunks = N * K / 8;
    if (dt == DT_F16) hipLaunchKernelGGL(pack_frag_kernel<f16>, dim3((chunks + 255) / 256), dim3(256), 0, s, (const f16*)w, (f16*)out, N, K);
    else hipLaunchKernelGGL(pack_frag_kernel<bf16>, dim3((chunks + 255) / 256), dim3(256), 0, s, (const bf16*)w, (bf16*)out, N, K);
    return hipGetLastError();
}

bool bert_attn_block_ok(int dt, int D, int heads, int L, int Kp, int ldq) {
    return (dt == DT_F16 || dt == DT_BF16) && D == kD && heads == kHeads && L >= 1 && L <= 16 * kMaxKT && Kp == kD && ldq % 8 == 0;
}

// y = LayerNorm(attention(qkv) Wo^T + bo + res) for B samples of L rows; res32 / y32 non-null: the f32 residual stream form (sum and LayerNorm in
// f32, y = the 16-bit operand copy, y32 = the stream).  y may alias res (rows are owned by one workgroup), y32 may alias res32.
hipError_t launch_bert_attn_block(const void* qkv, int ldq, const void* wo_frag, const float* bo, const void* res, const float* res32, const float* gamma,
                                  const float* beta, void* y, float* y32, int dt, int B, int L, const int* klens, float eps, hipStream_t s) {
    if (!bert_attn_block_ok(dt, kD, kHeads, L, kD, ldq) || B < 1) return hipErrorInvalidValue;
    if ((res32 != nullptr) != (y32 != nullptr) || (!res && !res32)) return hipErrorInvalidValue;
    constexpr int NRT = 3;
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bert_attn_block_kernel<f16, NRT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bert_attn_block_kernel<bf16, NRT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bert_attn_block_kernel<f16, NRT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bert_attn_block_kernel<bf16, NRT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_once.done();
    }
    BertBlockDev p;
    p.qkv = qkv; p.wo = wo_frag; p.bo = bo; p.res = res; p.res32 = res32; p.gamma = gamma; p.beta = beta; p.y = y; p.y32 = y32; p.klens = klens;
    p.B = B; p.L = L; p.ldq = ldq; p.eps = eps;
    static const char* prof_env = dev_env("HCM_BB_PROF_PTR");
    p.prof = prof_env ? reinterpret_cast<unsigned long long*>(strtoull(prof_env, nullptr, 0)) : nullptr;
    const int QT = (L + 15) / 16;
    p.wps = (QT + NRT - 1) / NRT;
    const int grid = (B + 7) / 8 * 8 * p.wps;
#define LB(T, F) hipLaunchKernelGGL((bert_attn_block_kernel<T, NRT, F>), dim3(grid), dim3(64 * kWaves), (bb_lds_bytes<T, NRT, F>()), s, p)
    if (res32) { if (dt == DT_F16) LB(f16, true); else LB(bf16, true); }
    else { if (dt == DT_F16) LB(f16, false); else LB(bf16, false); }
#undef LB
    return hipGetLastError();
}

}  // namespace hcm
