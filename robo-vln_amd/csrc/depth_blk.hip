// depth_blk_kernel (round 4): identity bottlenecks of the depth GroupNorm trunk's layer1 (32 x 32 maps, 128 channels, 32 mid) and layer2
// (16 x 16 maps, 256 channels, 64 mid) -- habitat's GroupNorm ResNet-50 as used at resnet_encoders.py:27-62 -- with a WHOLE SAMPLE of one
// trunk per workgroup.  As launches every conv of these blocks was a streaming kernel of 10-30 us with a GroupNorm whose statistics span
// workgroups (epilogue sums + a normalise-on-load consumer, or an apply pass): three launches and ~75 us per block.  Here the sample never
// leaves the CU between the three convs:
//   * a wave owns HW / 8 pixels for the whole block.  conv1's operand (the block input) and every weight fragment come straight from global /
//     L2 into registers in MFMA fragment layout (16 bytes per lane); the weights are tiny (34-136 KB per block) and L2-resident, the eight waves
//     read them redundantly -- no LDS staging, no barrier for them;
//   * the accumulators of a conv hold the wave's pixels x ALL its output channels, so GroupNorm is: per-lane sums -> 16-lane butterflies ->
//     one 8-wave exchange through a small LDS table (fixed order: deterministic) -> normalise in registers.  No second pass over memory;
//   * the two mid tensors live in LDS in the MFMA operand layout (pixel rows of CM channels, swizzled); the 3x3 conv reads shifted pixel rows;
//   * conv3 runs in 32- / 64-channel slabs (accumulator registers), adds the identity (re-read from L2, 16 bytes per lane after swap_pair) and stores
//     the block output; the next block of the run reads it back with L1-bypassing loads (same wave, same pixels).
// Same MFMA products as the launch-per-conv path on the same rounded operands; GroupNorm sums in a different (fixed) order: equal to that path
// to f32 round-off of the statistics, not bit for bit (tests/test_fusion_toggles_gpu.py, HCM_NO_DEPTH_BLK=1 in the development build).
#include "kernels.h"
#include "dev.h"

namespace hcm {

typedef float db_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 db_bf16x8 __attribute__((ext_vector_type(8)));
typedef int db_v4i __attribute__((ext_vector_type(4)));

template <typename T> struct DbMma;
template <> struct DbMma<bf16> {
    static __device__ __forceinline__ void run(db_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(db_bf16x8, a), __builtin_bit_cast(db_bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct DbMma<f16> {
    static __device__ __forceinline__ void run(db_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), acc, 0, 0, 0);
    }
};

struct DepthBlkDev {
    const char* x; char* y; int ld, nblocks;
    const char* w1[4]; const char* w2[4]; const char* w3[4];
    const float* g1[4]; const float* b1[4]; const float* g2[4]; const float* b2[4]; const float* g3[4]; const float* b3[4];
    float eps1[4], eps2[4], eps3[4];
};

// 16-byte load that bypasses the CU's vector L1 (sc1): the block input of blocks > 0 was stored by this very wave a moment ago
__device__ __forceinline__ uint4 db_load_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16));
}

__device__ __forceinline__ void db_dma16(unsigned lds_addr, unsigned voff, db_v4i rsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ db_v4i db_make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    db_v4i r;
    r[0] = (int)(unsigned)a; r[1] = (int)((unsigned)(a >> 32) & 0xFFFFu); r[2] = (int)bytes; r[3] = 0x00020000;
    return r;
}

// LOGW: log2 of the map side (5: 32 x 32, 4: 16 x 16); C: block channels per trunk; CM: mid channels.  16 GroupNorm groups per trunk.
template <typename T, int LOGW, int C, int CM>
__global__ __launch_bounds__(512) void depth_blk_kernel(DepthBlkDev p) {
    constexpr int WD = 1 << LOGW, HW = WD * WD;
    constexpr int PF = HW / 8 / 16;                 // 16-pixel fragments per wave: 8 / 2
    constexpr int CF = CM / 16;                     // channel fragments of the mid tensors: 2 / 4
    constexpr int KS1 = C / 32, KS3 = CM / 32;      // K steps of conv1 / conv3
    constexpr int TK = CM / 32;                     // K steps per tap of conv2
    constexpr int SF = PF >= 8 ? 2 : 4;             // channel fragments per output slab of conv3 (accumulator registers: SF x PF tiles)
    constexpr int NSLAB = C / (16 * SF);            // 32- / 64-channel output slabs
    constexpr int ROWB = CM * 2;                    // bytes per pixel row of a mid tensor: 64 / 128
    // LDS: two mid tensors, conv2's weights, the 8-wave reduction table.  conv1's weights travel in O2's bytes (dead until conv2's GroupNorm writes
    // it), conv3's in O1's (dead once conv2 has read it) -- all three matrices in FRAGMENT order: the 1 KB block of fragment (i, ks) holds lane l's
    // 16 bytes at l * 16, so that an A fragment is one conflict-free ds_read_b128 and one wave-level LDS-DMA request fills it.  (Round 4, first
    // form: every wave read its weight fragments from L2 itself -- 1.1 MB of 64-byte-segment loads per workgroup and block, 8 x redundant: 55 us
    // per block against 4.6 us of MFMAs.)
    constexpr int O1 = 0, O2 = HW * ROWB, W2R = 2 * HW * ROWB, W2_BYTES = CM * 9 * CM * 2, RED = W2R + W2_BYTES;
    constexpr int NF1 = (CM / 16) * (C / 32), NF2 = (CM / 16) * (9 * CM / 32), NF3 = (C / 16) * (CM / 32);     // fragments of the three matrices
    static_assert(NF1 * 1024 <= HW * ROWB && NF3 * 1024 <= HW * ROWB, "conv1 / conv3 weights fit a mid tensor's bytes");
    constexpr int CGM = CM / 16, CG3 = C / 16;      // channels per GroupNorm group: mid 2 / 4, block output 8 / 16
    static_assert((LOGW == 5 && C == 128 && CM == 32) || (LOGW == 4 && C == 256 && CM == 64), "depth layer1 / layer2 shapes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem + RED);              // [8 waves][64 slots]
    const int b = blockIdx.x, g = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int px0 = wave * (HW / 8);                                // the wave's pixels: px0 + j * 16 + fr
    const size_t samp = (size_t)b * HW * p.ld + (size_t)g * C;      // element offset of this (sample, trunk) inside x / y
    const unsigned map_bytes = (unsigned)(((size_t)(HW - 1) * p.ld + C) * 2);
    // position of the 16-byte chunk `ch8` (8 channels) of pixel row r inside a mid tensor
    auto o_addr = [&](int base, int r, int ch8) {
        if constexpr (ROWB == 128) return base + r * 128 + ((ch8 ^ (r & 7)) << 4);
        else return base + r * 64 + ((ch8 ^ ((4 - ((r >> 2) & 3)) & 3)) << 4);
    };
    // sum over the 16 lanes that share fg (the pixels of a fragment row)
    auto red16 = [&](float v) {
#pragma unroll
        for (int o = 1; o <= 8; o <<= 1) v += __shfl_xor(v, o, 64);
        return v;
    };

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // request weight matrix `w` ([rows][K], K contiguous) into `region` in fragment order: fragment f = i * (K / 32) + ks <- rows i * 16 + fr, k ks * 32 + fg * 8
    auto stage_w = [&](int region, const T* w, int nfrag, int K) {
        const db_v4i rs = db_make_rsrc(w, (unsigned)((size_t)nfrag * 1024));          // (rows * K * 2 bytes = nfrag KB)
        const int kf = K / 32;
        for (int f = wave; f < nfrag; f += 8) {
            const int i = f / kf, ks = f - i * kf;
            db_dma16(__builtin_amdgcn_readfirstlane(lds_base + region + f * 1024), (unsigned)(((i * 16 + fr) * K + ks * 32 + fg * 8) * 2), rs);
        }
    };
    auto wfrag = [&](int region, int f) { return *reinterpret_cast<const uint4*>(smem + region + f * 1024 + lane * 16); };
    auto wptr = [&](int blk, int which) {
        return which == 1 ? reinterpret_cast<const T*>(p.w1[blk]) + (size_t)g * CM * C
             : which == 2 ? reinterpret_cast<const T*>(p.w2[blk]) + (size_t)g * CM * 9 * CM
                          : reinterpret_cast<const T*>(p.w3[blk]) + (size_t)g * C * CM;
    };
    stage_w(O2, wptr(0, 1), NF1, C);
    stage_w(W2R, wptr(0, 2), NF2, 9 * CM);

    for (int blk = 0; blk < p.nblocks; ++blk) {
        const T* xin = reinterpret_cast<const T*>(blk == 0 ? p.x : p.y) + samp;
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xin), 0, (int)map_bytes, 0x00020000);
        // =================== conv1: 1x1, C -> CM.  A = weight fragment (16 channels x 32 k), B = 16 pixels x 32 k straight from global
        db_f32x4 acc[CF][PF];
#pragma unroll
        for (int i = 0; i < CF; ++i)
#pragma unroll
            for (int j = 0; j < PF; ++j) acc[i][j] = (db_f32x4){0.f, 0.f, 0.f, 0.f};
        {
            // the operand one K step ahead, in a second register set (all KS1 x PF loads at once would be 128 registers on the 32 x 32 maps)
            uint4 xb[2][PF];
            auto ldx = [&](int ks, uint4 (&dst)[PF]) {
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    const unsigned off = (unsigned)(((px0 + j * 16 + fr) * p.ld + ks * 32 + fg * 8) * 2);
                    dst[j] = db_load_sc1(rsx, off);       // (one path for every block: a per-load select would branch around each load)
                }
            };
            constexpr bool AHEAD = PF < 8;          // (32 x 32 maps: 8 fragments per K step -- one register set, the SIMD's other wave covers the latency)
            ldx(0, xb[0]);
            // conv1's weights (requested a whole phase ago) and conv2's have landed -- for every wave
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(AHEAD ? PF : PF) : "memory");          // (the PF operand loads just issued stay in flight)
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                if constexpr (AHEAD) { if (ks + 1 < KS1) ldx(ks + 1, xb[(ks + 1) & 1]); }
                else if (ks > 0) ldx(ks, xb[0]);
                uint4 wa[CF];
#pragma unroll
                for (int i = 0; i < CF; ++i) wa[i] = wfrag(O2, i * KS1 + ks);
#pragma unroll
                for (int i = 0; i < CF; ++i)
#pragma unroll
                    for (int j = 0; j < PF; ++j) DbMma<T>::run(acc[i][j], wa[i], xb[AHEAD ? (ks & 1) : 0][j]);
            }
        }
        // GroupNorm of a mid tensor (acc[i][j][e] = channel i * 16 + fg * 4 + e of pixel px0 + j * 16 + fr), ReLU, into LDS.
        // Groups of CGM = 2 channels (e pairs) or 4 (the lane's four): per lane NG = 4 / CGM group slots per channel fragment.
        auto gn_mid = [&](const float* gamma, const float* beta, float eps, int dst, auto&& after_stats) {
            constexpr int NG = 4 / CGM;                             // groups inside a lane's 4 channels: 2 / 1
            float sa[CF][NG], sq[CF][NG];
#pragma unroll
            for (int i = 0; i < CF; ++i)
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    float a = 0.f, q = 0.f;
#pragma unroll
                    for (int j = 0; j < PF; ++j)
#pragma unroll
                        for (int e = 0; e < CGM; ++e) { const float v = acc[i][j][n * CGM + e]; a += v; q += v * v; }
                    sa[i][n] = red16(a); sq[i][n] = red16(q);
                }
            // exchange: slot = ((i * 4 + fg) * NG + n) * 2 (+1); CF * 4 * NG * 2 = 64 floats per wave
            if (fr == 0) {
#pragma unroll
                for (int i = 0; i < CF; ++i)
#pragma unroll
                    for (int n = 0; n < NG; ++n) {
                        red[wave * 64 + ((i * 4 + fg) * NG + n) * 2] = sa[i][n];
                        red[wave * 64 + ((i * 4 + fg) * NG + n) * 2 + 1] = sq[i][n];
                    }
            }
            __syncthreads();
            after_stats();                                          // every wave has left the conv that produced acc: its operands' LDS bytes are free
            const float inv = 1.0f / (float)(HW * CGM);
#pragma unroll
            for (int i = 0; i < CF; ++i) {
                float mean[NG], rstd[NG];
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    float a = 0.f, q = 0.f;
#pragma unroll
                    for (int w = 0; w < 8; ++w) { a += red[w * 64 + ((i * 4 + fg) * NG + n) * 2]; q += red[w * 64 + ((i * 4 + fg) * NG + n) * 2 + 1]; }
                    mean[n] = a * inv;
                    rstd[n] = rsqrtf(relu_f(q * inv - mean[n] * mean[n]) + eps);
                }
                const float4 ga = *reinterpret_cast<const float4*>(gamma + g * CM + i * 16 + fg * 4), be = *reinterpret_cast<const float4*>(beta + g * CM + i * 16 + fg * 4);
                const float gaa[4] = {ga.x, ga.y, ga.z, ga.w}, bea[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
                for (int j = 0; j < PF; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][e] = relu_f((acc[i][j][e] - mean[e / CGM]) * rstd[e / CGM] * gaa[e] + bea[e]);
            }
            // pairs of channel fragments -> 8 consecutive channels of a pixel per lane -> 16 bytes into the operand layout
#pragma unroll
            for (int ip = 0; ip < CF / 2; ++ip)
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    float v[8];
                    swap_pair(acc[2 * ip][j], acc[2 * ip + 1][j], v);
                    const int r = px0 + j * 16 + fr;
                    const int ch8 = ip * 4 + (fg & 1) * 2 + (fg >> 1);
                    *reinterpret_cast<uint4*>(smem + o_addr(dst, r, ch8)) = pack_chunk<T>(v);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (weights requested by after_stats have landed)
            __syncthreads();                                        // the tensor is complete (and the reduction table free again)
        };
        gn_mid(p.g1[blk], p.b1[blk], p.eps1[blk], O1, [] {});
        // =================== conv2: 3x3 (pad 1), CM -> CM: k = tap * CM + ci; B = shifted pixel rows of O1 (zero outside the map)
#pragma unroll
        for (int i = 0; i < CF; ++i)
#pragma unroll
            for (int j = 0; j < PF; ++j) acc[i][j] = (db_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
            for (int kc = 0; kc < TK; ++kc) {
                uint4 wa[CF];
#pragma unroll
                for (int i = 0; i < CF; ++i) wa[i] = wfrag(W2R, i * (9 * TK) + tap * TK + kc);
                uint4 xb[PF];
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    const int r = px0 + j * 16 + fr;
                    const int sy = (r >> LOGW) + dy, sx = (r & (WD - 1)) + dx;
                    const bool ok = ((unsigned)sy < (unsigned)WD) & ((unsigned)sx < (unsigned)WD);
                    uint4 v = *reinterpret_cast<const uint4*>(smem + o_addr(O1, ok ? (sy << LOGW) + sx : 0, kc * 4 + fg));
                    if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
                    xb[j] = v;
                }
#pragma unroll
                for (int i = 0; i < CF; ++i)
#pragma unroll
                    for (int j = 0; j < PF; ++j) DbMma<T>::run(acc[i][j], wa[i], xb[j]);
            }
        }
        // conv2 is over for every wave after the statistics barrier: O1's bytes take conv3's weights while this GroupNorm normalises and stores O2
        gn_mid(p.g2[blk], p.b2[blk], p.eps2[blk], O2, [&] { stage_w(O1, wptr(blk, 3), NF3, CM); });
        // =================== conv3: 1x1, CM -> C in 64-channel slabs, GroupNorm (groups of CG3 = 8 / 16 channels), + identity, ReLU -> y
        {
            // B fragments of the wave's pixels: the whole K of conv3 (CM channels) -- read once, used by every slab
            uint4 xb[KS3][PF];
#pragma unroll
            for (int ks = 0; ks < KS3; ++ks)
#pragma unroll
                for (int j = 0; j < PF; ++j) xb[ks][j] = *reinterpret_cast<const uint4*>(smem + o_addr(O2, px0 + j * 16 + fr, ks * 4 + fg));
            T* yout = reinterpret_cast<T*>(p.y) + samp;
#pragma unroll 1
            for (int sl = 0; sl < NSLAB; ++sl) {
                db_f32x4 a3[SF][PF];
#pragma unroll
                for (int i = 0; i < SF; ++i)
#pragma unroll
                    for (int j = 0; j < PF; ++j) a3[i][j] = (db_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS3; ++ks) {
                    uint4 wa[SF];
#pragma unroll
                    for (int i = 0; i < SF; ++i) wa[i] = wfrag(O1, (sl * SF + i) * KS3 + ks);
#pragma unroll
                    for (int i = 0; i < SF; ++i)
#pragma unroll
                        for (int j = 0; j < PF; ++j) DbMma<T>::run(a3[i][j], wa[i], xb[ks][j]);
                }
                // identity rows of this slab in the post-swap_pair layout (8 consecutive channels of a pixel per lane), requested early
                uint4 idt[SF / 2][PF];
#pragma unroll
                for (int ip = 0; ip < SF / 2; ++ip)
#pragma unroll
                    for (int j = 0; j < PF; ++j) {
                        const unsigned off = (unsigned)(((px0 + j * 16 + fr) * p.ld + sl * (16 * SF) + ip * 32 + (fg & 1) * 16 + (fg >> 1) * 8) * 2);
                        idt[ip][j] = db_load_sc1(rsx, off);
                    }
                // statistics: CG3 = 8: a channel fragment holds two groups (fg 0-1 / fg 2-3); CG3 = 16: one
                float sa[SF], sq[SF];
#pragma unroll
                for (int i = 0; i < SF; ++i) {
                    float a = 0.f, q = 0.f;
#pragma unroll
                    for (int j = 0; j < PF; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float v = a3[i][j][e]; a += v; q += v * v; }
                    a = red16(a); q = red16(q);
                    a += __shfl_xor(a, 16, 64); q += __shfl_xor(q, 16, 64);             // the fg pair
                    if constexpr (CG3 == 16) { a += __shfl_xor(a, 32, 64); q += __shfl_xor(q, 32, 64); }
                    sa[i] = a; sq[i] = q;
                }
                if (fr == 0 && (fg & 1) == 0) {
#pragma unroll
                    for (int i = 0; i < SF; ++i) {
                        const int slot = (i * 2 + (fg >> 1)) * 2;           // (CG3 == 16: both halves carry the same sums)
                        red[wave * 64 + slot] = sa[i]; red[wave * 64 + slot + 1] = sq[i];
                    }
                }
                __syncthreads();
                // (first slab: every wave has read its conv3 operand out of O2 -- O2's bytes and conv2's weight region are free for the next block)
                if (sl == 0 && blk + 1 < p.nblocks) { stage_w(O2, wptr(blk + 1, 1), NF1, C); stage_w(W2R, wptr(blk + 1, 2), NF2, 9 * CM); }
                const float inv = 1.0f / (float)(HW * CG3);
#pragma unroll
                for (int i = 0; i < SF; ++i) {
                    float a = 0.f, q = 0.f;
                    const int slot = (i * 2 + (fg >> 1)) * 2;
#pragma unroll
                    for (int w = 0; w < 8; ++w) { a += red[w * 64 + slot]; q += red[w * 64 + slot + 1]; }
                    const float mean = a * inv;
                    const float rstd = rsqrtf(relu_f(q * inv - mean * mean) + p.eps3[blk]);
                    const int c = sl * (16 * SF) + i * 16 + fg * 4;
                    const float4 ga = *reinterpret_cast<const float4*>(p.g3[blk] + g * C + c), be = *reinterpret_cast<const float4*>(p.b3[blk] + g * C + c);
                    const float gaa[4] = {ga.x, ga.y, ga.z, ga.w}, bea[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
                    for (int j = 0; j < PF; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) a3[i][j][e] = (a3[i][j][e] - mean) * rstd * gaa[e] + bea[e];
                }
#pragma unroll
                for (int ip = 0; ip < SF / 2; ++ip)
#pragma unroll
                    for (int j = 0; j < PF; ++j) {
                        float v[8], r8[8];
                        swap_pair(a3[2 * ip][j], a3[2 * ip + 1][j], v);
                        cvt_chunk<T>(idt[ip][j], r8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = relu_f(v[e] + r8[e]);
                        const size_t off = (size_t)(px0 + j * 16 + fr) * p.ld + sl * (16 * SF) + ip * 32 + (fg & 1) * 16 + (fg >> 1) * 8;
                        *reinterpret_cast<uint4*>(yout + off) = pack_chunk<T>(v);
                    }
                __syncthreads();                                    // the reduction table is free for the next slab / block
            }
        }
        // the next block reads this block's output: this wave's own pixels -- its stores must have reached L2 (the loads bypass L1)
        if (blk + 1 < p.nblocks) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

hipError_t launch_depth_blk(const DepthBlk& d, int dt, hipStream_t s) {
    if ((dt != DT_F16 && dt != DT_BF16) || d.nblocks < 1 || d.nblocks > 4 || d.B < 1 || d.groups < 1 || (d.ld % 8) || !d.x || !d.y) return hipErrorInvalidValue;
    const bool l1 = d.side == 32 && d.C == 128 && d.CM == 32, l2 = d.side == 16 && d.C == 256 && d.CM == 64;
    if ((!l1 && !l2) || d.ld < d.groups * d.C) return hipErrorInvalidValue;
    if (d.nblocks > 1 && d.x == d.y) return hipErrorInvalidValue;      // (a run reads block 0's input from x and everything later from y)
    if ((size_t)d.side * d.side * d.ld * 2 >= 0x7FFFFFF0ull) return hipErrorInvalidValue;
    DepthBlkDev q;
    q.x = (const char*)d.x; q.y = (char*)d.y; q.ld = d.ld; q.nblocks = d.nblocks;
    for (int i = 0; i < 4; ++i) {
        q.w1[i] = (const char*)d.w1[i]; q.w2[i] = (const char*)d.w2[i]; q.w3[i] = (const char*)d.w3[i];
        q.g1[i] = d.g1[i]; q.b1[i] = d.b1[i]; q.g2[i] = d.g2[i]; q.b2[i] = d.b2[i]; q.g3[i] = d.g3[i]; q.b3[i] = d.b3[i];
        q.eps1[i] = d.eps1[i]; q.eps2[i] = d.eps2[i]; q.eps3[i] = d.eps3[i];
    }
    const void* fn;
    size_t lds;
    if (l1) {
        fn = dt == DT_BF16 ? reinterpret_cast<const void*>(depth_blk_kernel<bf16, 5, 128, 32>) : reinterpret_cast<const void*>(depth_blk_kernel<f16, 5, 128, 32>);
        lds = 2 * 1024 * 64 + 32 * 288 * 2 + 8 * 64 * 4;
    } else {
        fn = dt == DT_BF16 ? reinterpret_cast<const void*>(depth_blk_kernel<bf16, 4, 256, 64>) : reinterpret_cast<const void*>(depth_blk_kernel<f16, 4, 256, 64>);
        lds = 2 * 256 * 128 + 64 * 576 * 2 + 8 * 64 * 4;
    }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    void* args[] = {&q};
    return hipLaunchKernel(fn, dim3(d.B, d.groups), dim3(512), args, lds, s);
}

}  // namespace hcm
