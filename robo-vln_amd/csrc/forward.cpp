// Forward orchestration of the two HCM models: enqueues the HIP kernels on the caller's stream.
// No allocation and no host synchronisation inside a step: every intermediate lives in the per-handle
// workspace arena, whose size is found by running this same code once in "dry" mode at hcm_finalize().
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "model.h"

namespace hcm {

int depth_final_spatial(const hcm_config& c);
int depth_compress_channels(const hcm_config& c);

struct Act { void* p = nullptr; int B = 0, H = 0, W = 0, C = 0; };

struct Fwd {
    hcm_ctx* ctx;
    Arena& ar;
    hipStream_t s;
    int dt;
    size_t esz;
    bool dry;

    explicit Fwd(hcm_ctx* c) : ctx(c), ar(c->arena), s(c->stream), dt(c->dt_vla), esz(dt_size(c->dt_vla)), dry(c->arena.dry) {}

    // switch the storage type of the sub-network being enqueued
    void use(int d) { dt = d; esz = dt_size(d); }

    void ck(hipError_t e, const char* what) {
        if (e != hipSuccess) {
            ctx->failed = true;
            throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
        }
    }
    // fp16 calibration: which range slot the launches being enqueued belong to (-1: none)
    int calib_slot = -1;
    // pos >= 0: y is the UN-NORMALISED output of GroupNorm-trunk conv `pos` -- also reduced into that position's own range slot, from which a
    // calibration folds a power of two into the conv's weights instead of giving up fp16 (api.cpp calibrate_run)
    void calib_check(const void* y, int ydt, int rows, int cols, int ld, int pos = -1) {
        if (dry || !ctx->calib || calib_slot < 0 || ydt != DT_F16) return;
        ck(launch_absmax(y, ydt, rows, cols, ld, ctx->calib_buf + 2 * calib_slot, s), "calibration range check");
        if (pos >= 0 && calib_slot == 1) ck(launch_absmax(y, ydt, rows, cols, ld, ctx->calib_buf + 16 + 2 * pos, s), "calibration range check (conv position)");
    }
    void* alloc_t(size_t elems) { return ar.alloc(elems * esz); }
    float* alloc_f(size_t elems) { return (float*)ar.alloc(elems * 4); }

    void tap(const std::string& name, const void* p, bool is_t, Shape shape) {
        if (dry || !ctx->taps_on) return;
        size_t n = 1;
        for (auto d : shape) n *= (size_t)d;
        Tap& t = ctx->taps[name];
        if (t.cap < n) {
            if (t.dev) (void)hipFree(t.dev);
            ck(hipMalloc((void**)&t.dev, n * 4), "tap hipMalloc");
            t.cap = n;
        }
        t.n = n;
        t.shape = shape;
        if (is_t) ck(launch_convert_to_f32(p, dt, t.dev, n, s), "tap convert");
        else ck(hipMemcpyAsync(t.dev, p, n * 4, hipMemcpyDeviceToDevice, s), "tap copy");
    }

    // ---------------------------------------------------------------- GroupNorm on load (round 4)
    // A conv of the GroupNorm trunk whose output map is large (statistics from its epilogue) no longer launches gn_apply_kernel: its output stays
    // UN-normalised and a `Pending` record says how to normalise it; the conv that consumes the tensor applies it while staging its operand
    // (igemm.hip igemm_gnin_kernel).  pend_in: the record of the tensor the NEXT conv() call reads (consumed by that call); defer_out: where
    // conv_gn may leave the record of its output instead of launching the apply pass.  flush() materialises a pending tensor with the old kernel
    // (non-conv consumers, ineligible shapes, captured taps).  HCM_NO_GN_ONLOAD=1 (development build): the apply launches of rounds 1-3.
    struct Pending {
        bool valid = false;
        void* x = nullptr; const void* res = nullptr; const float* stats = nullptr; const float* gamma = nullptr; const float* beta = nullptr;
        int B = 0, hw = 0, C = 0, G = 0, ps = 0, relu = 0, dt = 0; float eps = 1e-5f;
        bool writeback = false;          // the consumer also stores the normalised tensor (a block output: the next residual add reads it)
        // round 5: `res` itself is still UN-normalised (the down-sample branch of a stage-first bottleneck: its own GroupNorm, no ReLU); flush() then
        // normalises both in ONE pass (gn_apply2_kernel), a consumer that normalises on load has the residual materialised first
        const float* res_stats = nullptr; const float* res_gamma = nullptr; const float* res_beta = nullptr; float res_eps = 1e-5f;
    };
    Pending pend_in;
    Pending* defer_out = nullptr;
    Pending* res_pending = nullptr;      // conv_gn: the residual passed to THIS call is the x of this record (consumed by the call)
    void materialise_res(Pending& q) {   // the pending residual of q -> normalised in place (the apply pass of rounds 1-4)
        if (!q.res_stats) return;
        const float* st = q.res_stats;
        q.res_stats = nullptr;
        if (dry) return;
        ck(launch_groupnorm_apply(const_cast<void*>(q.res), nullptr, q.res_gamma, q.res_beta, st, q.ps, q.dt, q.B, q.hw, q.C, q.G, q.res_eps, 0, s), "groupnorm apply (residual)");
    }
    void flush(Pending& q) {
        if (!q.valid) return;
        q.valid = false;
        if (q.res_stats) {
            const float* st = q.res_stats;
            q.res_stats = nullptr;
            if (dry) return;
            ck(launch_groupnorm_apply2(q.x, q.res, q.gamma, q.beta, q.stats, q.res_gamma, q.res_beta, st, q.ps, q.dt, q.B, q.hw, q.C, q.G, q.eps, q.res_eps, q.relu, s),
               "groupnorm apply (flush, pending residual)");
            return;
        }
        if (dry) return;
        ck(launch_groupnorm_apply(q.x, q.res, q.gamma, q.beta, q.stats, q.ps, q.dt, q.B, q.hw, q.C, q.G, q.eps, q.relu, s), "groupnorm apply (flush)");
    }
    bool gn_onload_enabled() const {
        static const bool off = dev_env("HCM_NO_GN_ONLOAD") != nullptr;
        return !off && !ctx->taps_on;
    }

    // ---------------------------------------------------------------- primitive wrappers
    // conv followed by GroupNorm (+ residual + ReLU): one launch when the output map is small enough for the fused
    // epilogue (igemm.hip), else conv + the stand-alone GroupNorm kernels
    void conv_gn(const ConvW& w, const Act& in, void* out, int stride, int pad, const void* res, const NormW& n, int G, bool relu, int Ho,
                 int Wo, int cg_true = 0) {
        static const bool no_fuse = dev_env("HCM_NO_GN_FUSE") != nullptr;
        const int C = w.groups * w.Cout, cg = C / G, hw = Ho * Wo;
        struct ClearDefer { Pending*& p; ~ClearDefer() { p = nullptr; } } clear_defer{defer_out};     // a request is for THIS call only
        Pending* rp = res_pending;       // `res` is this record's un-normalised tensor: taken over by the deferral below, else materialised first
        res_pending = nullptr;
        auto settle_res = [&]() {
            if (!rp || !rp->valid) return;
            rp->valid = false;
            if (!dry) ck(launch_groupnorm_apply(rp->x, nullptr, rp->gamma, rp->beta, rp->stats, rp->ps, rp->dt, rp->B, rp->hw, rp->C, rp->G, rp->eps, rp->relu, s), "groupnorm apply (residual)");
        };
        const float eps = 1e-5f * w.fold * w.fold;          // range-folded conv (ConvW::fold): GroupNorm((fold * x), eps * fold^2) == GroupNorm(x, eps)
        if (cg_true) {
            // zero-padded output channels (compression conv of 64*k-pixel frames, k not a power of two): the statistics count the real
            // channels of each group only -- the stand-alone slab kernel knows how
            settle_res();
            conv(w, in, out, stride, pad, nullptr, ACT_NONE, Ho, Wo);
            gn(out, res, n, in.B, hw, C, G, relu, cg_true, eps);
            return;
        }
        // the fused epilogue needs 64x128 tiles: not worth it when that leaves a long-K conv on a handful of workgroups
        // (the 3x3 compression conv: K = 18432 on 32 workgroups, 119 us fused vs 63 + 8 us separate)
        const long blocks = (long)((in.B * hw + 63) / 64) * ((w.Cout + 127) / 128) * w.groups;
        const bool fuse = !no_fuse && hw <= 64 && 64 % hw == 0 && cg % 8 == 0 && 128 % cg == 0 && w.Cout % cg == 0 && !w.bias &&
                          (blocks >= 64 || w.K < 4096);
        if (fuse) {
            settle_res();
            conv(w, in, out, stride, pad, res, relu ? ACT_RELU : ACT_NONE, Ho, Wo, &n, cg, nullptr, 0, 0, 0, eps);
        } else {
            // large maps: the statistics come out of the conv's epilogue (column sums of the f32 tile image per 64-row block and
            // group), so GroupNorm is one launch over the map instead of two
            static const bool no_cs = dev_env("HCM_NO_GN_EPISTATS") != nullptr;
            if (!no_cs && !w.bias && groupnorm_apply_ok(w.dt, hw, C, G) && w.Cout % cg == 0) {
                float* stats = alloc_f(gn_stats_floats(in.B, hw, G));
                conv(w, in, out, stride, pad, nullptr, ACT_NONE, Ho, Wo, nullptr, 0, stats, cg, hw, G);
                if (defer_out && gn_onload_enabled()) {
                    // normalised by whoever reads it (GroupNorm on load)
                    Pending& q = *defer_out;
                    q.valid = true; q.x = out; q.res = res; q.stats = stats; q.gamma = n.gamma; q.beta = n.beta;
                    q.B = in.B; q.hw = hw; q.C = C; q.G = G; q.ps = hw / 64; q.relu = relu ? 1 : 0; q.dt = w.dt; q.eps = eps; q.writeback = false;
                    q.res_stats = nullptr;
                    if (rp && rp->valid && rp->x == res && rp->hw == hw && rp->C == C && rp->G == G && rp->ps == q.ps && !rp->res && !rp->relu) {
                        // the residual stays un-normalised too: whoever materialises this tensor normalises both in one pass
                        q.res_stats = rp->stats; q.res_gamma = rp->gamma; q.res_beta = rp->beta; q.res_eps = rp->eps;
                        rp->valid = false;
                    } else settle_res();
                    defer_out = nullptr;
                    return;
                }
                settle_res();
                static const bool skip_apply = dev_env("HCM_SKIP_GN_APPLY") != nullptr;
                if (!dry && !skip_apply) ck(launch_groupnorm_apply(out, res, n.gamma, n.beta, stats, hw / 64, w.dt, in.B, hw, C, G, eps, relu ? 1 : 0, s), "groupnorm apply");
                return;
            }
            settle_res();
            conv(w, in, out, stride, pad, nullptr, ACT_NONE, Ho, Wo);
            gn(out, res, n, in.B, hw, C, G, relu, 0, eps);
        }
    }
    void conv(const ConvW& w, const Act& in, void* out, int stride, int pad, const void* res, int act, int Ho, int Wo,
              const NormW* gnw = nullptr, int gn_cg = 0, float* cs_part = nullptr, int cs_cg = 0, int cs_hw = 0, int cs_G = 0, float gn_eps = 1e-5f) {
        Pending pin = pend_in;
        pend_in.valid = false;
        if (pin.valid && pin.x != in.p) { flush(pin); }             // (a record for another tensor: cannot happen in the trunk loop; be safe)
        if (pin.valid) materialise_res(pin);                         // a consumer that normalises on load adds a NORMALISED residual
        if (dry) return;
        IGemm g;
        g.cs_part = cs_part; g.cs_cg = cs_cg; g.cs_hw = cs_hw; g.cs_G = cs_G;
        if (gnw) { g.gn_gamma = gnw->gamma; g.gn_beta = gnw->beta; g.gn_cg = gn_cg; g.gn_hw = Ho * Wo; g.gn_eps = gn_eps; }
        g.x = in.p; g.w = w.w; g.bias = w.bias; g.res = res; g.y = out;
        if (pin.valid) {
            g.gi_stats = pin.stats; g.gi_gamma = pin.gamma; g.gi_beta = pin.beta; g.gi_ps = pin.ps; g.gi_cg = pin.C / pin.G; g.gi_G = pin.G;
            g.gi_hw = pin.hw; g.gi_relu = pin.relu; g.gi_eps = pin.eps; g.gi_res = pin.res; g.gi_out = pin.writeback ? pin.x : nullptr;
        }
        g.B = in.B; g.H = in.H; g.W = in.W; g.Cin = in.C; g.xC = in.C;
        g.Ho = Ho; g.Wo = Wo; g.KH = w.KH; g.KW = w.KW; g.stride = stride; g.pad = pad;
        g.M = in.B * Ho * Wo; g.N = w.Cout; g.K = w.K; g.Kp = w.Kp; g.ldy = w.Cout; g.ldr = w.Cout; g.act = act;
        if (w.groups > 1) {     // hi|lo pair: group g reads channels [g*Cin, (g+1)*Cin) and writes [g*Cout, (g+1)*Cout)
            g.Cin = w.Cin; g.xC = in.C; g.ldy = g.ldr = w.groups * w.Cout;
            g.groups = w.groups; g.g_x = w.Cin; g.g_w = (long long)w.Cout * w.Kp; g.g_b = w.Cout; g.g_y = w.Cout;
        }
        if (pin.valid && !igemm_gnin_ok(g, w.dt)) {
            // this consumer cannot normalise on load: materialise the tensor first
            flush(pin);
            g.gi_stats = nullptr; g.gi_res = nullptr; g.gi_out = nullptr;
        }
        ck(launch_igemm(g, w.dt, s), "conv igemm");
        // (with the fused GroupNorm epilogue the un-normalised values never leave f32 registers: `out` is the normalised map)
        calib_check(out, w.dt, g.M, w.groups * w.Cout, w.groups * w.Cout, gnw ? -1 : w.calib_pos);
    }
    // y[M][ldy(+col)] = act(A[M][lda] @ W^T + b (+res))
    // Skinny long-K layers (the M = batch projections behind the encoders, SimpleCNN's 25088-wide FC) would run on a few
    // dozen workgroups with hundreds of serial K steps: they are split along K through the grouped-launch mechanism (group s
    // = columns [s*Ks, (s+1)*Ks) of A and W, f32 partials), then summed in a fixed order with bias + activation.
    // folded LayerNorm around a BERT GEMM (IGemm::ln_* / rln_*)
    struct LnFold {
        const float* ln_s = nullptr; const float* ln_t = nullptr; float* stats_out = nullptr;              // consumer side (QKV / FFN1): forced onto gemm256f_kernel
        const float* rln_stats = nullptr; const float* rln_gamma = nullptr; const float* rln_beta = nullptr;  // residual side (attention output / FFN2)
        const float* part_in = nullptr; float* part_out = nullptr; int P = 0; int force_choice = -1;          // partial row sums: producer epilogue -> consumer prologue
    };
    void linear(const LinW& w, const void* a, int M, int lda, void* y, int ldy, int act, bool out_f32,
                const void* res = nullptr, int ldr = 0, int wdt = -1, const LnFold* lf = nullptr) {
        const int wd = wdt < 0 ? w.dt : wdt;
        const int CHw = wd == DT_F32 ? 4 : 8;
        static const bool no_split = dev_env("HCM_NO_SPLITK") != nullptr;
        const int S = no_split ? 1 : splitk_slices(M, w.N, w.K, CHw, res != nullptr);
        float* part = S > 1 ? alloc_f((size_t)S * M * w.N) : nullptr;       // allocated in the dry run too
        if (dry) return;
        IGemm g;
        g.x = a; g.w = w.w; g.bias = w.bias; g.res = res; g.y = y;
        g.B = M; g.Cin = w.K; g.xC = lda; g.M = M; g.N = w.N; g.K = w.K; g.Kp = w.Kp;
        g.ldy = ldy; g.ldr = ldr ? ldr : w.N; g.act = act; g.out_f32 = out_f32 ? 1 : 0;
        if (lf) {
            g.ln_s = lf->ln_s; g.ln_t = lf->ln_t; g.ln_stats_out = lf->stats_out; g.ln_eps = 1e-12f;
            if (lf->ln_s) { g.impl = 2; g.force_gemm256 = 1; }
            g.ln_part_in = lf->part_in; g.ln_part_out = lf->part_out; g.ln_part_P = lf->P; g.force_choice = lf->force_choice;
            g.rln_stats = lf->rln_stats; g.rln_gamma = lf->rln_gamma; g.rln_beta = lf->rln_beta;
        }
        if (S > 1) {
            const int Ks = w.K / S;
            g.K = Ks; g.Cin = Ks; g.bias = nullptr; g.y = part; g.ldy = w.N; g.ldr = w.N; g.act = ACT_NONE; g.out_f32 = 1;
            g.groups = S; g.g_x = Ks; g.g_w = Ks; g.g_b = 0; g.g_y = (long long)M * w.N;
            ck(launch_igemm(g, wd, s), "linear igemm (split-K)");
            ck(launch_splitk_reduce(part, w.bias, y, wd, S, M, w.N, ldy, act, out_f32 ? 1 : 0, s), "split-K reduce");
            if (!out_f32) calib_check(y, wd, M, w.N, ldy);
            return;
        }
        ck(launch_igemm(g, wd, s), "linear igemm");
        if (!out_f32) calib_check(y, wd, M, w.N, ldy);
    }
    void gn(void* x, const void* res, const NormW& n, int B, int HW, int C, int G, bool relu, int cg_true = 0, float eps = 1e-5f) {
        float* stats = alloc_f(gn_stats_floats(B, HW, G));
        if (dry) return;
        ck(launch_groupnorm(x, res, n.gamma, n.beta, stats, dt, B, HW, C, G, eps, relu ? 1 : 0, s, cg_true), "groupnorm");
    }
    void ln(const void* x, const void* res, const NormW& n, const float* post, int post_rows, void* y, int rows, int D, float eps) {
        if (dry) return;
        ck(launch_layernorm(x, res, n.gamma, n.beta, post, post_rows, y, dt, rows, D, eps, s), "layernorm");
    }

    // ---------------------------------------------------------------- ResNet-50 trunks
    // torchvision resnet50 conv1..layer4 with BN folded (RGB), or habitat GN-ResNet50 (+compression) (depth).
    // Returns the trunk output living in arena memory.
    struct Stem { const void* x; int x_dt; float scale; int H, W, Cin; };   // raw frame feeding the 7x7/2 stem conv

    void stem_conv(const ConvW& w, const Stem& st, int B, int k, int stride, int pad, void* out, int Ho, int Wo, int act) {
        if (dry) return;
        IGemm g;
        g.x = st.x; g.w = w.w; g.bias = w.bias; g.y = out;
        g.B = B; g.H = st.H; g.W = st.W; g.Cin = st.Cin; g.xC = st.Cin;
        g.Ho = Ho; g.Wo = Wo; g.KH = k; g.KW = k; g.stride = stride; g.pad = pad;
        g.M = B * Ho * Wo; g.N = w.Cout; g.K = w.K; g.Kp = w.Kp; g.ldy = w.Cout; g.ldr = w.Cout; g.act = act;
        g.x_src_dt = st.x_dt; g.x_scale = st.scale;
        // row-run K layout (kh*24 + kw*3 + ci): the 7x7 stem's row-run weights, and any 3-channel kernel with KW = 8 (SimpleCNN's
        // 8x8/4, where it coincides with the plain layout); the vector gather exists for f32 frames only
        g.x_rowrun = (w.K == w.KH * 24 && st.Cin == 3 && st.x_dt == DT_F32) ? 1 : 0;
        ck(launch_igemm(g, w.dt, s), "stem conv");
        calib_check(out, w.dt, g.M, w.Cout, w.Cout, w.calib_pos);
    }

    // 16-bit RGB trunks: pack the frame once, then the stem is an ordinary LDS-DMA implicit GEMM (kernels.h: launch_pack_frame)
    void stem_conv_packed(const ConvW& w, const Stem& st, int B, void* out, int Ho, int Wo, int act, int hpool = 0) {
        void* pk = alloc_t(pack_frame_elems(B, st.H, st.W));
        if (dry) return;
        ck(launch_pack_frame(st.x, st.x_dt, pk, w.dt, B, st.H, st.W, st.scale, s), "pack frame");
        IGemm g;
        g.x = pk; g.w = w.w; g.bias = w.bias; g.y = out;
        g.B = B; g.H = st.H + 6; g.W = (st.W + 8) / 2; g.Cin = 32; g.xC = 8;
        g.Ho = Ho; g.Wo = Wo; g.KH = 7; g.KW = 1; g.stride = 2; g.stride_w = 1; g.pad = 0;
        g.M = B * Ho * Wo; g.N = w.Cout; g.K = w.K; g.Kp = w.Kp; g.ldy = w.Cout; g.ldr = w.Cout; g.act = act;
        g.hpool = hpool;
        ck(launch_igemm(g, w.dt, s), "stem conv (packed)");
        calib_check(out, w.dt, hpool ? B * Ho * (Wo / 2) : g.M, w.Cout, w.Cout);
    }

    Act trunk(const TrunkW& t, const Stem& st, int B, int Ho, int Wo, const std::string& tapname) {
        const int c1 = t.conv1.Cout;
        const int G = t.groups * (t.pair ? 2 : 1);                 // GroupNorm groups over the (possibly paired) channels
        auto CO = [](const ConvW& w) { return w.groups * w.Cout; };   // total output channels of a (grouped) conv
        // the largest activation: the stem conv's map, or -- when that map has an odd side, so that the pooled map is (Ho + 1) / 2 wide --
        // layer1's output (4 x the channels on the pooled map)
        const int Hp_ = (Ho + 2 - 3) / 2 + 1, Wp_ = (Wo + 2 - 3) / 2 + 1;
        const size_t max_elems = std::max((size_t)B * Ho * Wo * c1, (size_t)B * Hp_ * Wp_ * 4 * c1);
        void* slot[4];
        for (auto& p : slot) p = alloc_t(max_elems);
        // 7x7/2 stem: implicit GEMM gathering straight from the raw frame (permute, /255, dtype conversion fused)
        // f32 RGB frames take the row-run fast gather; uint8 frames and the 1-channel depth stem the element-wise one
        static const bool no_pack = dev_env("HCM_NO_STEM_PACK") != nullptr;
        const bool packed = !t.gn && !no_pack && t.conv1_packed.w != nullptr && st.Cin == 3 && !(st.W & 1) && !(st.H & 1) &&
                            (st.x_dt == DT_F32 || st.x_dt == DT_U8);
        const bool fast = !t.gn && st.x_dt == DT_F32 && t.conv1_rowrun.w != nullptr;
        // packed RGB stem: the horizontal half of the 3x3/2 max-pool rides in the conv's epilogue (the 128-channel pair map is written at
        // half width), a vertical-only pool follows; exact.  Not while taps are captured: `_conv1` is the full-width map.
        static const bool no_hpool = dev_env("HCM_NO_STEM_HPOOL") != nullptr;
        const bool hpool = packed && !no_hpool && !ctx->taps_on && Wo >= 2 && Wo <= 128 && !(Wo & (Wo - 1)) && (c1 % 64) == 0;
        static const bool no_stem_fuse = dev_env("HCM_NO_STEM_FUSE") != nullptr;
        const bool fused_stem = packed && hpool && !no_stem_fuse && t.conv1_packed.groups == 1 && rgb_stem_pool_ok(t.conv1_packed.dt, st.H, st.W, c1, t.conv1_packed.Kp);
        // ... and layer1 block 0's 1x1 reduction (64 -> 64 per model) from the pooled row while it is in registers: its launch and its read of the
        // pooled map disappear (HCM_NO_STEM_RED=1, development build: the launch)
        static const bool no_stem_red = dev_env("HCM_NO_STEM_RED") != nullptr;
        bool stem_red = false;
        if (fused_stem && !no_stem_red && !t.blocks.empty()) {
            const ConvW& q = t.blocks[0].c1;
            stem_red = q.KH == 1 && q.KW == 1 && q.Cin == 64 && q.Cout == 64 && q.K == 64 && q.Kp == 64 && q.bias && q.dt == t.conv1_packed.dt && q.groups * 64 == c1 &&
                       q.fold == 1.f;
        }
        // GroupNorm statistics of the packed depth stem from its conv's epilogue (see conv_gn)
        static const bool no_cs_stem = dev_env("HCM_NO_GN_EPISTATS") != nullptr;
        float* stem_stats = nullptr;
        if (t.gn && st.x_dt == -2 && !no_cs_stem && groupnorm_apply_ok(t.conv1_packed.dt, Ho * Wo, c1, G) && c1 % (c1 / G) == 0)
            stem_stats = alloc_f(gn_stats_floats(B, Ho * Wo, G));
        if (t.gn && st.x_dt == -2) {
            // depth stem on the packed 1-channel frame: kernel row = 8 contiguous elements (7 taps + a zero-weight slot), the GEMM's
            // "virtual pixel" = the stride of 2 elements
            if (!dry) {
                const ConvW& w = t.conv1_packed;
                IGemm g;
                g.x = st.x; g.w = w.w; g.bias = nullptr; g.y = slot[0];
                g.B = B; g.H = st.H + 6; g.W = (st.W + 8) / 2; g.Cin = 8; g.xC = 2;
                g.Ho = Ho; g.Wo = Wo; g.KH = 7; g.KW = 1; g.stride = 2; g.stride_w = 1; g.pad = 0;
                g.M = B * Ho * Wo; g.N = w.Cout; g.K = w.K; g.Kp = w.Kp; g.ldy = w.Cout; g.ldr = w.Cout; g.act = ACT_NONE;
                if (stem_stats) { g.cs_part = stem_stats; g.cs_cg = c1 / G; g.cs_hw = Ho * Wo; g.cs_G = G; }
                ck(launch_igemm(g, w.dt, s), "depth stem conv (packed)");
                calib_check(slot[0], w.dt, g.M, w.Cout, w.Cout, w.calib_pos);
            }
        } else if (fused_stem) {
            // round 6: pack -> ONE launch for conv1 + ReLU + both halves of the max-pool (stem.hip); the pooled map lands in slot[1]
            void* pk = alloc_t(pack_frame_elems(B, st.H, st.W));
            if (!dry) {
                ck(launch_pack_frame(st.x, st.x_dt, pk, t.conv1_packed.dt, B, st.H, st.W, st.scale, s), "pack frame");
                const ConvW* r1 = stem_red ? &t.blocks[0].c1 : nullptr;
                ck(launch_rgb_stem_pool(pk, t.conv1_packed.w, t.conv1_packed.bias, slot[1], t.conv1_packed.dt, B, st.H, st.W, c1, s, r1 ? r1->w : nullptr,
                                        r1 ? r1->bias : nullptr, r1 ? slot[0] : nullptr), "stem conv + max-pool (one launch)");
                // (the pooled map has the conv map's maximum: ReLU'd values, max-pooled)
                calib_check(slot[1], t.conv1_packed.dt, B * (Ho / 2) * (Wo / 2), c1, c1);
                if (r1) calib_check(slot[0], r1->dt, B * (Ho / 2) * (Wo / 2), c1, c1);
            }
        } else if (packed && hpool) stem_conv_packed(t.conv1_packed, st, B, slot[0], Ho, Wo, ACT_RELU, 1);
        else if (packed) stem_conv_packed(t.conv1_packed, st, B, slot[0], Ho, Wo, ACT_RELU);
        else stem_conv(fast ? t.conv1_rowrun : t.conv1, st, B, 7, 2, 3, slot[0], Ho, Wo, t.gn ? ACT_NONE : ACT_RELU);
        const float stem_eps = 1e-5f * t.conv1.fold * t.conv1.fold;      // (conv1 and conv1_packed carry the same fold)
        // round 5: the stem's GroupNorm + ReLU applied ON LOAD by the max-pool (maxpool_gn_kernel: the apply pass over the trunk's largest map and its
        // launch disappear; bit-identical).  Not while taps are captured (`_conv1` is the normalised map).  HCM_NO_GN_POOL=1 (development build): the A/B.
        static const bool no_gn_pool = dev_env("HCM_NO_GN_POOL") != nullptr;
        const bool gn_pool = t.gn && stem_stats && !no_gn_pool && !ctx->taps_on && !hpool && maxpool_gn_ok(t.conv1_packed.dt, c1, G);
        if (gn_pool) {
            // (nothing here: the pool below normalises)
        } else if (t.gn && stem_stats) {
            if (!dry) ck(launch_groupnorm_apply(slot[0], nullptr, t.n_conv1.gamma, t.n_conv1.beta, stem_stats, Ho * Wo / 64, t.conv1_packed.dt, B, Ho * Wo, c1, G,
                                                stem_eps, 1, s), "groupnorm apply (stem)");
        } else if (t.gn) gn(slot[0], nullptr, t.n_conv1, B, Ho * Wo, c1, G, true, 0, stem_eps);
        if (!hpool) tap(tapname + "_conv1", slot[0], true, {B, Ho, Wo, c1});
        const int Hp = (Ho + 2 - 3) / 2 + 1, Wp = (Wo + 2 - 3) / 2 + 1;
        if (fused_stem) {
            // (pooled by the stem launch)
        } else if (hpool) {
            if (!dry) ck(launch_vpool3s2(slot[0], slot[1], dt, B, Ho, Wo / 2, c1, s), "maxpool (vertical half)");
        } else if (gn_pool) {
            if (!dry) ck(launch_maxpool3x3s2_gn(slot[0], slot[1], t.n_conv1.gamma, t.n_conv1.beta, stem_stats, Ho * Wo / 64, stem_eps, G, t.conv1_packed.dt, B, Ho, Wo, c1,
                                                Hp, Wp, s), "maxpool over GroupNorm on load (stem)");
        } else if (!dry) ck(launch_maxpool3x3s2(slot[0], slot[1], dt, B, Ho, Wo, c1, Hp, Wp, s), "maxpool");
        Act x{slot[1], B, Hp, Wp, c1};
        mark(tapname + ".stem_end");
        int xi = 1;
        int bidx = 0;
        int pre = -1;          // slot already holding THIS block's 1x1 reduction output (computed by the previous block's fused launch)
        if (stem_red) pre = 0; // ... or, for the first block, by the stem launch
        Pending xpend;         // GroupNorm trunk: the block input x is still un-normalised (its first consumer, the block's c1, normalises + stores it)
        // (development build, timing only -- the results are then wrong: HCM_GN_STOP=<k> drops the launches of the GroupNorm trunks' blocks
        //  k.. and of the compression conv, HCM_SKIP_GN_APPLY=1 the stand-alone normalisation passes)
        static const int gn_stop = dev_env("HCM_GN_STOP") ? atoi(dev_env("HCM_GN_STOP")) : -1;
        const bool dry_saved = dry;
        for (size_t bi = 0; bi < t.blocks.size(); ++bi) {
            if (t.gn && gn_stop >= 0 && (int)bi >= gn_stop) dry = true;
            // GroupNorm trunk, 8 x 8 maps, 512 / 128 channels, 16 groups per trunk (the depth encoder's layer3 at 256-pixel frames): the run of
            // identity bottlenecks behind the layer's first block is ONE launch, a workgroup per (sample, trunk) -- igemm.hip depth_l3_kernel
            static const bool no_l3 = dev_env("HCM_NO_DEPTH_L3") != nullptr;
            if (t.gn && !no_l3 && !ctx->taps_on && pre < 0 && x.H * x.W == 64 && x.H == 8 && t.groups == 16) {
                flush(xpend);      // (8 x 8 maps come out of fused epilogues: nothing is pending here at 256-pixel frames)
                auto plain = [&](const BottleneckW& q) {
                    const bool dt16 = q.c1.dt == DT_F16 || q.c1.dt == DT_BF16;
                    return dt16 && !q.has_ds && q.stride == 1 && q.c1.KH == 1 && q.c1.Cin == 512 && q.c1.Cout == 128 && q.c1.Kp == 512 && q.c2.KH == 3 && q.c2.KW == 3 &&
                           q.c2.Cin == 128 && q.c2.Cout == 128 && q.c2.Kp == 1152 && q.c3.KH == 1 && q.c3.Cin == 128 && q.c3.Cout == 512 && q.c3.Kp == 128 &&
                           !q.c1.bias && !q.c2.bias && !q.c3.bias && q.c2.dt == q.c1.dt && q.c3.dt == q.c1.dt && q.c1.groups == q.c2.groups &&
                           q.c1.groups == q.c3.groups;
                };
                size_t run = 0;
                while (bi + run < t.blocks.size() && run < 6 && plain(t.blocks[bi + run]) && t.blocks[bi + run].c1.groups == t.blocks[bi].c1.groups) ++run;
                if (run >= 2 && x.C == t.blocks[bi].c1.groups * 512) {
                    int fo = 0;
                    while (fo == xi) ++fo;
                    if (!dry) {
                        DepthL3 q;
                        q.x = x.p; q.y = slot[fo]; q.ld = x.C; q.B = B; q.groups = t.blocks[bi].c1.groups; q.nblocks = (int)run;
                        for (size_t r = 0; r < run; ++r) {
                            const BottleneckW& bb = t.blocks[bi + r];
                            q.w1[r] = bb.c1.w; q.w2[r] = bb.c2.w; q.w3[r] = bb.c3.w;
                            q.g1[r] = bb.n1.gamma; q.b1[r] = bb.n1.beta; q.g2[r] = bb.n2.gamma; q.b2[r] = bb.n2.beta; q.g3[r] = bb.n3.gamma; q.b3[r] = bb.n3.beta;
                            q.eps1[r] = 1e-5f * bb.c1.fold * bb.c1.fold; q.eps2[r] = 1e-5f * bb.c2.fold * bb.c2.fold; q.eps3[r] = 1e-5f * bb.c3.fold * bb.c3.fold;
                        }
                        ck(launch_depth_l3(q, t.blocks[bi].c1.dt, s), "depth layer3 run");
                        calib_check(slot[fo], t.blocks[bi].c1.dt, B * 64, x.C, x.C);
                    }
                    x = Act{slot[fo], B, x.H, x.W, x.C};
                    xi = fo;
                    bidx += (int)run;
                    bi += run - 1;
                    if (bidx == 13) { tap(tapname + "_layer3", x.p, true, {B, x.H, x.W, x.C}); mark(tapname + ".layer3_end"); }
                    continue;
                }
            }
            // GroupNorm trunk, 32 x 32 maps with 128 / 32 channels or 16 x 16 with 256 / 64 (the depth encoder's layer1 / layer2 at 256-pixel frames): a run
            // of identity bottlenecks is ONE launch, a whole sample of one trunk per workgroup -- depth_blk.hip (round 4)
            static const bool no_blk = dev_env("HCM_NO_DEPTH_BLK") != nullptr;
            // (not in the sizing pass: the launch-per-conv form, which a step with captured taps takes, carves statistics buffers this form does not need)
            if (t.gn && !no_blk && !ctx->taps_on && !dry && pre < 0 && x.H == x.W && (x.H == 32 || x.H == 16) && t.groups == 16) {
                const int Cb = x.H == 32 ? 128 : 256, Cm = Cb / 4;
                auto plain = [&](const BottleneckW& q) {
                    const bool dt16 = q.c1.dt == DT_F16 || q.c1.dt == DT_BF16;
                    return dt16 && !q.has_ds && q.stride == 1 && q.c1.KH == 1 && q.c1.Cin == Cb && q.c1.Cout == Cm && q.c1.Kp == Cb && q.c2.KH == 3 && q.c2.KW == 3 &&
                           q.c2.Cin == Cm && q.c2.Cout == Cm && q.c2.Kp == 9 * Cm && q.c3.KH == 1 && q.c3.Cin == Cm && q.c3.Cout == Cb && q.c3.Kp == Cm &&
                           !q.c1.bias && !q.c2.bias && !q.c3.bias && q.c2.dt == q.c1.dt && q.c3.dt == q.c1.dt && q.c1.groups == q.c2.groups &&
                           q.c1.groups == q.c3.groups;
                };
                size_t run = 0;
                while (bi + run < t.blocks.size() && run < 4 && plain(t.blocks[bi + run]) && t.blocks[bi + run].c1.groups == t.blocks[bi].c1.groups) ++run;
                if (run >= 1 && x.C == t.blocks[bi].c1.groups * Cb) {
                    flush(xpend);                  // the kernel reads the materialised block input
                    int fo = 0;
                    while (fo == xi) ++fo;
                    if (!dry) {
                        DepthBlk q;
                        q.x = x.p; q.y = slot[fo]; q.ld = x.C; q.B = B; q.groups = t.blocks[bi].c1.groups; q.nblocks = (int)run; q.side = x.H; q.C = Cb; q.CM = Cm;
                        for (size_t r = 0; r < run; ++r) {
                            const BottleneckW& bb = t.blocks[bi + r];
                            q.w1[r] = bb.c1.w; q.w2[r] = bb.c2.w; q.w3[r] = bb.c3.w;
                            q.g1[r] = bb.n1.gamma; q.b1[r] = bb.n1.beta; q.g2[r] = bb.n2.gamma; q.b2[r] = bb.n2.beta; q.g3[r] = bb.n3.gamma; q.b3[r] = bb.n3.beta;
                            q.eps1[r] = 1e-5f * bb.c1.fold * bb.c1.fold; q.eps2[r] = 1e-5f * bb.c2.fold * bb.c2.fold; q.eps3[r] = 1e-5f * bb.c3.fold * bb.c3.fold;
                        }
                        ck(launch_depth_blk(q, t.blocks[bi].c1.dt, s), "depth layer1/2 run");
                        calib_check(slot[fo], t.blocks[bi].c1.dt, B * x.H * x.W, x.C, x.C);
                    }
                    x = Act{slot[fo], B, x.H, x.W, x.C};
                    xi = fo;
                    bidx += (int)run;
                    bi += run - 1;
                    if (bidx == 3 || bidx == 7) mark(tapname + ".layer" + std::to_string(bidx == 3 ? 1 : 2) + "_end");
                    continue;
                }
            }
            const BottleneckW& b = t.blocks[bi];
            int fr[3], nf = 0;
            if (pre >= 0) fr[nf++] = pre;
            for (int i = 0; i < 4; ++i) if (i != xi && i != pre) fr[nf++] = i;
            // A pending block input still needs its identity (xpend.res: the previous block's input or down-sample output) until THIS block's c1 has
            // normalised it -- and that slot counts as free here.  c1 must not write its output there: the launch would read the identity rows of some
            // pixels while other workgroups already store over them (round 4: 0.3-1 % of the steps of a 128-pixel configuration differed from run to
            // run under the three-chain step; tools/step_determinism.py, test_three_chain_step_is_deterministic).  c2 / c3 run behind c1 and may take it.
            if (nf != 3) throw std::runtime_error("trunk slot rotation: expected three free activation slots");
            if (xpend.valid && xpend.res && slot[fr[0]] == xpend.res) std::swap(fr[0], fr[1]);
            void* sa = slot[fr[0]]; void* sb = slot[fr[1]]; void* sc = slot[fr[2]];
            if (xpend.valid && xpend.res && sa == xpend.res) throw std::runtime_error("trunk slot rotation: c1 would overwrite a pending identity");
            const int Ho2 = (x.H + 2 - 3) / b.stride + 1, Wo2 = (x.W + 2 - 3) / b.stride + 1;
            Act o1{sa, B, x.H, x.W, CO(b.c1)};
            const bool have_o1 = pre >= 0;
            pre = -1;
            Pending p1, p2;
            if (have_o1) {}
            else if (t.gn) {
                if (xpend.valid) { xpend.writeback = true; pend_in = xpend; xpend.valid = false; }     // c1 normalises the block input and stores it
                defer_out = &p1;
                conv_gn(b.c1, x, sa, 1, 0, nullptr, b.n1, G, true, x.H, x.W);
            }
            else conv(b.c1, x, sa, 1, 0, nullptr, ACT_RELU, x.H, x.W);
            // BN-folded trunks, 64 / 128 mid channels (layer1, layer2), 16-bit storage: 3x3 conv + 1x1 expansion + identity in ONE
            // launch, the mid tensor stays in LDS (igemm.hip: bneck23_kernel; bit-identical to the two launches)
            static const bool no_tail = dev_env("HCM_NO_BNECK_FUSE") != nullptr;
            static const bool no_next = dev_env("HCM_NO_BNECK_NEXT") != nullptr;
            static const bool no_256 = dev_env("HCM_NO_BNECK256") != nullptr;        // layer3 (256 mid channels) as a launch per conv (A/B, toggle test)
            static const bool force_256 = dev_env("HCM_FORCE_BNECK256") != nullptr;   // ... fused whatever the grid size (the toggle test's small batch)
            const BottleneckW* nb = bi + 1 < t.blocks.size() ? &t.blocks[bi + 1] : nullptr;
            static const int next_only = dev_env("HCM_BNECK_NEXT_ONLY") ? atoi(dev_env("HCM_BNECK_NEXT_ONLY")) : 0;   // A/B aid: 64 or 128 = only blocks with that many mid channels
            const bool next = nb && !no_next && (!next_only || next_only == b.c2.Cout) && nb->c1.KH == 1 && nb->c1.KW == 1 && nb->c1.Cin == b.c3.Cout && nb->c1.Kp == nb->c1.Cin &&
                              nb->c1.bias && nb->c1.groups == b.c2.groups && nb->c1.dt == b.c2.dt &&
                              ((nb->c1.Cout == b.c2.Cout && (nb->c1.Cout == 64 || nb->c1.Cout == 128 || nb->c1.Cout == 256)) || (b.c2.Cout == 64 && nb->c1.Cout == 128));
            // (256 mid channels -- layer3: only the "tail + next block's reduction" form exists, so the layer's last block stays a launch per conv)
            // ... and only when its 128-pixel tiles fill the chip (one 149 KB workgroup per CU: a tile's ~70 us are a latency chain that a small grid
            // cannot hide -- B = 16: 69 us fused vs 34 us as three launches; B = 64: 83 vs 111)
            const long tiles256 = (long)b.c2.groups * (((long)B * ((x.H + 2 - 3) / b.stride + 1) * ((x.W + 2 - 3) / b.stride + 1) + 127) / 128);
            const bool c1_ok = b.c2.Cout == 64 || b.c2.Cout == 128 || (b.c2.Cout == 256 && next && !no_256 && (tiles256 >= 192 || force_256));
            if (!t.gn && !no_tail && (b.c2.dt == DT_BF16 || b.c2.dt == DT_F16) && c1_ok && b.c2.KH == 3 &&
                b.c2.Cin == b.c2.Cout && b.c2.Kp == 9 * b.c2.Cin && b.c3.Cout == 4 * b.c2.Cout && b.c3.Kp == b.c2.Cout && b.c2.bias && b.c3.bias &&
                b.c3.groups == b.c2.groups) {
                const void* idt = x.p;
                static const bool no_dsfold = dev_env("HCM_NO_BNECK_DSFOLD") != nullptr;
                // layer1's first block: its 1x1 down-sample conv (64 -> 256, same stride) rides in the expansion GEMM as 64 more K columns
                // ([W3 | Wds], bias b3 + bds): no down-sample launch, no identity tensor.  One rounding instead of two on that path, so
                // not bit-identical to the separate launches (closer to the fp32 oracle)
                const bool dsfold = b.has_ds && !no_dsfold && next && b.c3ds.w && b.c2.Cout == 64 && nb->c1.Cout == 64 && b.ds.Cin == 64 &&
                                    b.c3ds.Kp == 128 && b.c3ds.groups == b.c2.groups && x.C == b.c2.groups * 64;
                if (b.has_ds && !dsfold) {
                    conv(b.ds, x, sb, b.stride, 0, nullptr, ACT_NONE, Ho2, Wo2);
                    idt = sb;
                }
                // ... and the NEXT block's 1x1 reduction from the output tile in the same launch (bneck231_kernel), into a slot this
                // launch does not read: the block input's when the identity is the down-sample conv's output, else the spare one
                const int pre_slot = (b.has_ds && !dsfold) ? xi : fr[1];
                if (!dry) {
                    Bneck23 q;
                    q.x = sa; q.w2 = b.c2.w; q.b2 = b.c2.bias; q.w3 = b.c3.w; q.b3 = b.c3.bias; q.res = idt; q.y = sc;
                    q.B = B; q.H = x.H; q.W = x.W; q.C1 = b.c2.Cout; q.xC = o1.C; q.stride = b.stride;
                    q.ldy = q.ldr = CO(b.c3);
                    if (b.c2.groups > 1) {
                        q.groups = b.c2.groups; q.g_x = b.c2.Cin; q.g_w2 = (long long)b.c2.Cout * b.c2.Kp; q.g_b2 = b.c2.Cout;
                        q.g_w3 = (long long)b.c3.Cout * b.c3.Kp; q.g_b3 = b.c3.Cout; q.g_y = b.c3.Cout;
                    }
                    if (next) {
                        q.w1 = nb->c1.w; q.b1 = nb->c1.bias; q.o1 = slot[pre_slot]; q.CN = nb->c1.Cout; q.ldo = CO(nb->c1);
                        if (b.c2.groups > 1) { q.g_w1 = (long long)nb->c1.Cout * nb->c1.Kp; q.g_b1 = nb->c1.Cout; q.g_o1 = nb->c1.Cout; }
                    }
                    if (dsfold) {
                        q.w3 = b.c3ds.w; q.b3 = b.c3ds.bias; q.res = nullptr;
                        q.xd = x.p; q.xdC = x.C; q.KD = 1;
                        if (b.c2.groups > 1) { q.g_w3 = (long long)b.c3ds.Cout * b.c3ds.Kp; q.g_xd = 64; }
                    }
                    ck(launch_bneck23(q, b.c2.dt, s), "bottleneck tail");
                    calib_check(sc, b.c2.dt, B * Ho2 * Wo2, CO(b.c3), CO(b.c3));        // the block output (the mid tensor never leaves LDS)
                }
                if (next) pre = pre_slot;
                x = Act{sc, B, Ho2, Wo2, CO(b.c3)};
                xi = fr[2];
                ++bidx;
                if (bidx == 3 || bidx == 7 || bidx == 13 || bidx == 16) {
                    tap(tapname + "_layer" + std::to_string(bidx == 3 ? 1 : bidx == 7 ? 2 : bidx == 13 ? 3 : 4), x.p, true, {B, x.H, x.W, x.C});
                    mark(tapname + ".layer" + std::to_string(bidx == 3 ? 1 : bidx == 7 ? 2 : bidx == 13 ? 3 : 4) + "_end");
                }
                continue;
            }
            Act o2{sb, B, Ho2, Wo2, CO(b.c2)};
            if (t.gn) {
                pend_in = p1; p1.valid = false;
                defer_out = &p2;
                conv_gn(b.c2, o1, sb, b.stride, 1, nullptr, b.n2, G, true, Ho2, Wo2);
            }
            else conv(b.c2, o1, sb, b.stride, 1, nullptr, ACT_RELU, Ho2, Wo2);
            const void* idt = x.p;
            Pending pds;
            if (b.has_ds) {
                if (t.gn) {
                    // round 5: the down-sample branch's GroupNorm stays pending as well when the block output does (gn_apply2_kernel normalises both in
                    // one pass; HCM_NO_GN_RES2=1, development build: its own apply pass as in rounds 1-4)
                    static const bool no_res2 = dev_env("HCM_NO_GN_RES2") != nullptr;
                    if (!no_res2 && bi + 1 < t.blocks.size()) defer_out = &pds;
                    conv_gn(b.ds, x, sa, b.stride, 0, nullptr, b.nds, G, false, Ho2, Wo2);   // o1 is dead: reuse its slot
                }
                else conv(b.ds, x, sa, b.stride, 0, nullptr, ACT_NONE, Ho2, Wo2);
                idt = sa;
            }
            if (t.gn) {
                pend_in = p2; p2.valid = false;
                // the block output stays pending when another bottleneck follows: its c1 normalises, adds the identity and stores it
                if (bi + 1 < t.blocks.size()) defer_out = &xpend;
                if (pds.valid) res_pending = &pds;
                conv_gn(b.c3, o2, sc, 1, 0, idt, b.n3, G, true, Ho2, Wo2);          // relu(GN(conv) + identity)
            } else {
                conv(b.c3, o2, sc, 1, 0, idt, ACT_RELU, Ho2, Wo2);                  // relu(bn(conv) + identity), fused
            }
            x = Act{sc, B, Ho2, Wo2, CO(b.c3)};
            xi = fr[2];
            ++bidx;
            if (bidx == 3 || bidx == 7 || bidx == 13 || bidx == 16) {
                tap(tapname + "_layer" + std::to_string(bidx == 3 ? 1 : bidx == 7 ? 2 : bidx == 13 ? 3 : 4), x.p, true, {B, x.H, x.W, x.C});
                mark(tapname + ".layer" + std::to_string(bidx == 3 ? 1 : bidx == 7 ? 2 : bidx == 13 ? 3 : 4) + "_end");
            }
        }
        flush(xpend);
        if (t.gn) {
            int fr = (xi + 1) & 3;
            conv_gn(t.compress, x, slot[fr], 1, 1, nullptr, t.n_compress, t.pair ? 2 : 1, true, x.H, x.W, t.compress_true);
            x = Act{slot[fr], B, x.H, x.W, CO(t.compress)};
        }
        dry = dry_saved;
        return x;
    }

    Act rgb_trunk(const TrunkW& t, const void* rgb, int rgb_dt, int B, const std::string& tapname) {
        const int H = ctx->cfg.rgb_h, W = ctx->cfg.rgb_w;
        const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
        // permute(0,3,1,2) + `/ 255.0` (resnet_encoders.py:211-213) are folded into the stem conv's gather
        calib_slot = 2;
        Act o = trunk(t, Stem{rgb, rgb_dt, 1.0f / 255.0f, H, W, 3}, B, Ho, Wo, tapname);
        calib_slot = -1;
        return o;
    }
    Act depth_trunk(const TrunkW& t, const float* depth, int B, const std::string& tapname) {
        calib_slot = 1;
        Act o = depth_trunk_(t, depth, B, tapname);
        calib_slot = -1;
        return o;
    }
    Act depth_trunk_(const TrunkW& t, const float* depth, int B, const std::string& tapname) {
        const int H = ctx->cfg.depth_h / 2, W = ctx->cfg.depth_w / 2;
        const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
        static const bool no_pack = dev_env("HCM_NO_STEM_PACK") != nullptr;
        if (t.conv1_packed.w && !no_pack && !(W & 1) && !(H & 1)) {
            // 16-bit trunks: avg_pool2d(2) writes straight into the zero-bordered frame of the packed stem
            void* pk = alloc_t((size_t)B * (H + 6) * (W + 8) + 64);
            if (!dry) ck(launch_avgpool2_f32_padded(depth, pk, dt, B, ctx->cfg.depth_h, ctx->cfg.depth_w, s), "avgpool2 (padded)");
            return trunk(t, Stem{pk, -2, 1.0f, H, W, 1}, B, Ho, Wo, tapname);       // x_dt = -2: already packed
        }
        void* pooled = alloc_t((size_t)B * H * W);
        if (!dry) ck(launch_avgpool2_f32(depth, pooled, dt, B, ctx->cfg.depth_h, ctx->cfg.depth_w, s), "avgpool2");
        return trunk(t, Stem{pooled, dt, 1.0f, H, W, 1}, B, Ho, Wo, tapname);
    }

    // SimpleAllCNN.cnn (simple_cnns.py:76-100) -> f32 features into out[b*ld + col0 ..]
    void simple_cnn(const SimpleCnnW& w, const void* x, int x_dt, float scale, int B, float* out, int ld) {
        const int H = w.h, W = w.w;
        const int h1 = (H - 8) / 4 + 1, h2 = (h1 - 4) / 2 + 1, h3 = (h2 - 3) / 1 + 1;
        const int w1 = (W - 8) / 4 + 1, w2 = (w1 - 4) / 2 + 1, w3 = (w2 - 3) / 1 + 1;
        // Round 5: the three convolutions of the depth CNN in ONE launch, both intermediate maps in LDS (simplecnn.hip: simplecnn3_kernel; bit-identical
        // to the launches below; HCM_NO_CNN3=1 of the development build: the launches)
        static const bool no_cnn3 = dev_env("HCM_NO_CNN3") != nullptr;
        // (the fp16 calibration forward takes the launch-per-conv form: both intermediate maps of the fused launch live in LDS only, so their
        //  range would otherwise never reach calib_check and an overflow in conv1 / conv2 could not trigger the bf16 re-build or a range fold)
        const bool calibrating = ctx->calib && dt == DT_F16;
        if (!no_cnn3 && !calibrating && w.cin == 1 && x_dt == DT_F32 && H == W && w.c1_frag && w.c2_frag && w.c0_packed.w && w.c0_packed.K == 64 &&
            w.c0_packed.Kp == 64 && simplecnn3_ok(dt, H)) {
            void* y2f = alloc_t((size_t)B * h3 * w3 * 32);
            if (dry && dt == DT_F16) {
                // the sizing pass also reserves what the calibration forward's launch-per-conv form allocates beyond y2f
                (void)alloc_t((size_t)B * h1 * w1 * 32); (void)alloc_t((size_t)B * H * W + 64); (void)alloc_t((size_t)B * h2 * w2 * 64);
            }
            if (!dry) ck(launch_simplecnn3((const float*)x, w.c0_packed.w, w.c0_packed.bias, w.c1_frag, w.c1.bias, w.c2_frag, w.c2.bias, y2f, dt, B, H, s),
                         "simple cnn (three convolutions, one launch)");
            linear(w.fc, y2f, B, h3 * w3 * 32, out, ld, ACT_RELU, true);
            return;
        }
        void* y0 = alloc_t((size_t)B * h1 * w1 * 32);
        static const bool no_pack = dev_env("HCM_NO_STEM_PACK") != nullptr;
        if (w.c0_packed.w && !no_pack && (x_dt == DT_F32 || x_dt == DT_U8) && (w.cin == 3 || x_dt == DT_F32)) {
            // 16-bit path: convert / pack the frame once ([B][H][W][cp], cp = 4 for RGB, 1 for depth); a kernel row of an output
            // pixel is then one contiguous run of 8 pixels and the conv an ordinary LDS-DMA implicit GEMM over "virtual
            // pixels" of 4 real ones (the stride): KH = 8, KW = 1, Cin = 8*cp, pixel stride 4*cp elements
            const int cp = w.cin == 3 ? 4 : 1;
            static const bool no_direct = dev_env("HCM_NO_DEPTH_CONV0") != nullptr;
            const bool direct = cp == 1 && H == W && !no_direct && w.c0_packed.K == 64 && w.c0_packed.Kp == 64 && depth_conv8x8s4_ok(dt, H, ACT_RELU);
            void* pk = direct ? nullptr : alloc_t((size_t)B * H * W * cp + 64);
            if (direct) {
                // depth: one pass over the raw f32 frame (simplecnn.hip), bit-identical to convert + implicit GEMM
                if (!dry) ck(launch_depth_conv8x8s4((const float*)x, w.c0_packed.w, w.c0_packed.bias, y0, dt, B, H, ACT_RELU, s), "simple cnn conv0 (direct)");
            } else if (!dry) {
                if (cp == 4) ck(launch_pack_frame(x, x_dt, pk, dt, B, H, W, scale, s, 0), "pack frame");
                else ck(launch_convert_from_f32((const float*)x, pk, dt, (size_t)B * H * W, s), "depth convert");
                IGemm g;
                g.x = pk; g.w = w.c0_packed.w; g.bias = w.c0_packed.bias; g.y = y0;
                g.B = B; g.H = H; g.W = W / 4; g.Cin = 8 * cp; g.xC = 4 * cp;
                g.Ho = h1; g.Wo = w1; g.KH = 8; g.KW = 1; g.stride = 4; g.stride_w = 1; g.pad = 0;
                g.M = B * h1 * w1; g.N = 32; g.K = w.c0_packed.K; g.Kp = w.c0_packed.Kp; g.ldy = 32; g.ldr = 32; g.act = ACT_RELU;
                ck(launch_igemm(g, dt, s), "simple cnn conv0 (packed)");
            }
            calib_check(y0, dt, B * h1 * w1, 32, 32);
        } else {
            stem_conv(w.c0, Stem{x, x_dt, scale, H, W, w.cin}, B, 8, 4, 0, y0, h1, w1, ACT_RELU);
        }
        void* y1 = alloc_t((size_t)B * h2 * w2 * 64);
        conv(w.c1, Act{y0, B, h1, w1, 32}, y1, 2, 0, nullptr, ACT_RELU, h2, w2);
        void* y2 = alloc_t((size_t)B * h3 * w3 * 32);
        conv(w.c2, Act{y1, B, h2, w2, 64}, y2, 1, 0, nullptr, ACT_NONE, h3, w3);
        linear(w.fc, y2, B, h3 * w3 * 32, out, ld, ACT_RELU, true);
    }

    // ---------------------------------------------------------------- BERT encoder
    void bert(const BertW& w, const void* ids, int ids_dt, int B, void* x, const int* lens) {
        const hcm_config& c = ctx->cfg;
        // L is per call (ctx->cur_L <= cfg.instr_len); buffers are carved at the maximum length so that the workspace layout -- and with
        // it the cached instruction stream of HCM_ACT_REUSE_INSTRUCTION -- does not move when L changes
        const int L = ctx->cur_L, D = c.bert_hidden, rows = B * L;
        const size_t rmax = (size_t)B * c.instr_len;
        void* qkv = alloc_t(rmax * 3 * D);
        void* ctxb = alloc_t(rmax * D);
        void* tmp = alloc_t(rmax * D);
        void* hbuf = alloc_t(rmax * c.bert_inter);
        if (!dry) ck(launch_bert_embed(ids, ids_dt, w.word, w.pos, w.type0, w.ln.gamma, w.ln.beta, x, dt, B, L, D, c.bert_vocab, 1e-12f, s), "bert_embed");
        tap("hi.bert_emb", x, true, {B, L, D});
        // bf16 tiles (precision "bf16", or BERT after a range fall-back): the RESIDUAL STREAM stays in f32.  bf16 keeps 8 significant bits; rounding
        // the stream itself four times per layer (both sums in front of the LayerNorms, both LayerNorm outputs) is what costs 6-9e-3 of the 1e-2
        // record tolerance.  Here only the GEMM operands are bf16: the projections write `sum = branch + stream` in f32, the LayerNorm
        // reads that and writes the stream in f32 beside its bf16 operand copy (HCM_BERT_BF16_STREAM=0: the all-16-bit form, for the A/B).
        static const bool f32_stream_off = dev_env("HCM_BERT_BF16_STREAM") && atoi(dev_env("HCM_BERT_BF16_STREAM")) == 0;
        const bool f32_stream = dt == DT_BF16 && !f32_stream_off && (D == 768 || D == 256 || D == 512);
        float* xf = f32_stream ? alloc_f(rmax * D) : nullptr;       // the stream
        float* sumf = f32_stream ? alloc_f(rmax * D) : nullptr;     // branch + stream, in front of a LayerNorm
        if (f32_stream && !dry) ck(launch_convert_to_f32(x, dt, xf, (size_t)rows * D, s), "bert stream init");
        // round 5: the stream value is not materialised by the LayerNorm launches any more.  Two f32 buffers alternate as "the pre-LayerNorm sum": a
        // projection reads the PREVIOUS sum, rebuilds the stream value from it with the statistics its LayerNorm left (rln_apply in the epilogue: the
        // LayerNorm kernel's own expression) and writes the next sum into the other buffer; the LayerNorm writes the bf16 operand + (mean, rstd) only
        // (a 15.7 MB f32 write per LayerNorm gone at B = 64).  HCM_BERT_STREAM_MAT=1 (development build): the materialised stream of round 4.
        static const bool stream_mat = dev_env("HCM_BERT_STREAM_MAT") != nullptr;
        float* lnst = (f32_stream && !stream_mat && dev_env("HCM_BERT_FUSE") == nullptr) ? alloc_f(rmax * 2) : nullptr;        // (mean, rstd) per row of the latest LayerNorm (the fused-block experiment keeps the materialised stream)
        // (lnst is ONE (mean, rstd) buffer: the LayerNorm launch that writes it and the projection epilogue that reads it are enqueued on the same stream `s`,
        //  in that order, every time -- BERT is one chain; nothing here may move onto a second stream without giving each LayerNorm its own statistics.
        //  With HCM_BERT_FUSE set but the fused block not applicable (`blk` false below) the code takes round 4's materialised stream: correct, and covered by the
        //  toggle test's HCM_BERT_STREAM_MAT side.)
        float* cur_sum = nullptr;           // buffer holding the latest pre-LayerNorm sum (nullptr: xf holds the stream itself, as after the embedding)
        const NormW* cur_ln = nullptr;      // ... and the LayerNorm that turns it into the stream
        auto linear_res_f32 = [&](const LinW& lw, const void* a, int lda) {     // sum = a @ W^T + b + stream
            float* dst = lnst ? ((cur_sum == sumf) ? xf : sumf) : sumf;
            if (!dry) {
                IGemm g;
                g.x = a; g.w = lw.w; g.bias = lw.bias; g.res = lnst && cur_sum ? cur_sum : xf; g.res_f32 = 1; g.y = dst; g.out_f32 = 1;
                if (lnst && cur_sum) { g.rln_stats = lnst; g.rln_gamma = cur_ln->gamma; g.rln_beta = cur_ln->beta; }
                g.B = rows; g.Cin = lw.K; g.xC = lda; g.M = rows; g.N = lw.N; g.K = lw.K; g.Kp = lw.Kp; g.ldy = D; g.ldr = D; g.act = ACT_NONE;
                ck(launch_igemm(g, lw.dt, s), "bert linear (f32 stream)");
            }
            cur_sum = dst;
        };
        auto ln_stream = [&](const NormW& n) {                                   // x (16-bit operand) = LayerNorm(latest sum); the stream stays implicit
            if (lnst) {
                if (!dry) ck(launch_layernorm_f32in(cur_sum, n.gamma, n.beta, x, nullptr, dt, rows, D, 1e-12f, s, lnst), "bert layernorm (f32 stream, statistics)");
                cur_ln = &n;
            } else if (!dry) ck(launch_layernorm_f32in(sumf, n.gamma, n.beta, x, xf, dt, rows, D, 1e-12f, s), "bert layernorm (f32 stream)");
        };
        int li = 0;
        mark("bert.embed");
        // LayerNorm FOLDED into the GEMMs around it (round 4; fp16 tiles, QKV and FFN1 on the 256 x 256 kernel): the tensor between two projections
        // stays the pre-LayerNorm sum u, its consumer computes rstd * (u W'^T - mean * s) + t with the row statistics taken inside its own K loop, and
        // the next residual add rebuilds LayerNorm(u) from those statistics.  24 of the 26 LayerNorm launches of a step disappear (the embedding's is
        // fused already, the last layer's output is materialised).  HCM_NO_LN_FOLD=1 (development build): the launches.
        // MEASURED SLOWER and therefore OFF unless HCM_LN_FOLD=1 (development build): taking the row statistics inside the consumer's K loop costs 128
        // v_dot2_f32_f16 per K tile and wave, ~12 cycles each beside the MFMAs -- gemm256f 29.3 -> 39.0 us per launch, +0.23 ms of kernel time per step
        // against the 0.155 ms of LayerNorm launches removed (same-box A/B 14 463 vs 14 814 env-steps/s).  Numerically it is fine (toggle test: 3e-5 on the
        // record).  The statistics would have to come out of the PRODUCER's epilogue (partials per 32-column wave slice, combined in the consumer's
        // prologue) to pay; not built.
        static const bool no_fold = dev_env("HCM_LN_FOLD") == nullptr || dev_env("HCM_NO_LN_FOLD") != nullptr;
        static const int fold_min_rows = dev_env("HCM_LN_FOLD_MIN_ROWS") ? atoi(dev_env("HCM_LN_FOLD_MIN_ROWS")) : 4096;      // (the toggle test's small engine)
        // Decided per ENGINE (its maximum batch), not per call: the folded form is not bit-identical to the LayerNorm launches, and a refresh of two
        // environments' instruction streams (hcm_refresh_instruction: BERT at batch 2) must reproduce what the full batch computes, bit for bit --
        // so in an engine sized for big batches every bert() call folds, the folded GEMMs forced onto gemm256f_kernel whatever the row count (a row's
        // result there does not depend on the batch it is in).  Engines sized for a few environments keep the launches (the 256-wide tiles would cost
        // a single-environment step ~0.2 ms).
        bool fold = !no_fold && dt == DT_F16 && !f32_stream && !ctx->taps_on && !w.layers.empty() && w.layers[0].ff1_f.w != nullptr && (D % 64) == 0 &&
                    (3 * D) % 8 == 0 && c.bert_inter % 8 == 0 && (size_t)c.max_batch * c.instr_len >= (size_t)fold_min_rows;
        if (fold) {
            float* st1 = alloc_f(rmax * 2);                 // (mean, rstd) per row of u1 (pre-ln1) / u2 (pre-ln2)
            float* st2 = alloc_f(rmax * 2);
            // HCM_LN_FOLD=2: the row sums come out of the producing projection's register epilogue as partials per 32-column slice (forced onto the
            // 128 x 128 8-wave tiles whatever the row count) and are combined in the consumer's prologue; =1: taken inside the consumer's K loop
            static const int fold_mode = atoi(dev_env("HCM_LN_FOLD"));
            const bool parts = fold_mode >= 2 && D % 32 == 0;
            const int P = D / 32;
            float* pt1 = parts ? alloc_f(rmax * P * 2) : nullptr;
            float* pt2 = parts ? alloc_f(rmax * P * 2) : nullptr;
            const long b128 = (long)((rows + 127) / 128) * ((D + 127) / 128);
            const int ch_o = 4 * 6, ch_f2 = b128 <= 256 ? 9 * 6 : 7 * 6;       // attention output: 2-buffer ring; FFN2 (K = 3072): deep / interleaved ring
            bool xmat = true;                                // x holds a materialised LayerNorm output (the embedding's); afterwards u2 of the previous layer
            const NormW* prev_ln2 = nullptr;
            for (const BertLayerW& l : w.layers) {
                if (li) mark("bert.layer" + std::to_string(li));
                LnFold fq, fo, f1, f2;
                if (xmat) linear(l.qkv, x, rows, D, qkv, 3 * D, ACT_NONE, false);
                else {
                    fq.ln_s = l.qkv_s; fq.ln_t = l.qkv_t; fq.stats_out = st2; fq.part_in = pt2; fq.P = parts ? P : 0;
                    linear(l.qkv_f, x, rows, D, qkv, 3 * D, ACT_NONE, false, nullptr, 0, -1, &fq);
                }
                if (!dry) ck(launch_attention(qkv, (char*)qkv + (size_t)D * esz, (char*)qkv + (size_t)2 * D * esz, ctxb, dt, B, c.bert_heads,
                                              L, L, 3 * D, 3 * D, 3 * D, D, B, s, lens), "bert attention");
                // u1 = ctx Wo^T + bo + LayerNorm_prev(x)
                if (!xmat) { fo.rln_stats = st2; fo.rln_gamma = prev_ln2->gamma; fo.rln_beta = prev_ln2->beta; }
                if (parts) { fo.part_out = pt1; fo.P = P; fo.force_choice = ch_o; }
                linear(l.o, ctxb, rows, D, tmp, D, ACT_NONE, false, x, D, -1, (xmat && !parts) ? nullptr : &fo);
                f1.ln_s = l.ff1_s; f1.ln_t = l.ff1_t; f1.stats_out = st1; f1.part_in = pt1; f1.P = parts ? P : 0;
                linear(l.ff1_f, tmp, rows, D, hbuf, c.bert_inter, ACT_GELU, false, nullptr, 0, -1, &f1);
                // u2 = h W2^T + b2 + LayerNorm1(u1)
                f2.rln_stats = st1; f2.rln_gamma = l.ln1.gamma; f2.rln_beta = l.ln1.beta;
                if (parts) { f2.part_out = pt2; f2.P = P; f2.force_choice = ch_f2; }
                linear(l.ff2, hbuf, rows, c.bert_inter, x, D, ACT_NONE, false, tmp, D, -1, &f2);
                xmat = false;
                prev_ln2 = &l.ln2;
                ++li;
            }
            ln(x, nullptr, *prev_ln2, nullptr, 0, x, rows, D, 1e-12f);        // the encoder's output, materialised (in place: a row is one wave's registers)
            return;
        }
        // Round 5 experiment, OFF unless HCM_BERT_FUSE=1 (development build): attention + output projection + residual + LayerNorm as ONE launch per
        // layer (bert_block.hip; bit-identical to the three launches below).  MEASURED SLOWER -- 37.7 us against 29.8 us for the three launches at
        // B = 64, 33 against 15 us at B = 1: a workgroup that owns 48 rows x all 768 columns has to ingest the whole 1.18 MB of W_o through its CU's
        // one L1 / texture path (64 B and ONE cache-line look-up per clock: >= 9 us, 13-17 measured), behind an attention phase that is one latency
        // chain per head (17-20 us for 12 heads on 8 waves).  The chip-wide launches spread the same weight bytes over 240 CUs.  DESIGN_LOG R5.1.
        static const bool no_blk = dev_env("HCM_BERT_FUSE") == nullptr;
        const bool blk = !no_blk && !w.layers.empty() && w.layers[0].o.dt == dt && w.layers[0].o_frag && bert_attn_block_ok(dt, D, c.bert_heads, L, w.layers[0].o.Kp, 3 * D);
        for (const BertLayerW& l : w.layers) {
            if (li) mark("bert.layer" + std::to_string(li));
            linear(l.qkv, x, rows, D, qkv, 3 * D, ACT_NONE, false);
            if (blk) {
                if (!dry) ck(launch_bert_attn_block(qkv, 3 * D, l.o_frag, l.o.bias, f32_stream ? nullptr : x, f32_stream ? xf : nullptr, l.ln1.gamma, l.ln1.beta,
                                                    x, f32_stream ? xf : nullptr, dt, B, L, lens, 1e-12f, s), "bert attention block");
            } else {
            if (!dry) ck(launch_attention(qkv, (char*)qkv + (size_t)D * esz, (char*)qkv + (size_t)2 * D * esz, ctxb, dt, B, c.bert_heads,
                                          L, L, 3 * D, 3 * D, 3 * D, D, B, s, lens), "bert attention");
            if (f32_stream) {
                linear_res_f32(l.o, ctxb, D);
                ln_stream(l.ln1);
            } else {
                linear(l.o, ctxb, rows, D, tmp, D, ACT_NONE, false, x, D);
                ln(tmp, nullptr, l.ln1, nullptr, 0, x, rows, D, 1e-12f);
            }
            }
            linear(l.ff1, x, rows, D, hbuf, c.bert_inter, ACT_GELU, false);
            if (f32_stream) {
                linear_res_f32(l.ff2, hbuf, c.bert_inter);
                ln_stream(l.ln2);
            } else {
                linear(l.ff2, hbuf, rows, c.bert_inter, tmp, D, ACT_NONE, false, x, D);
                ln(tmp, nullptr, l.ln2, nullptr, 0, x, rows, D, 1e-12f);
            }
            if (li == 0) tap("hi.bert_l0", x, true, {B, L, D});
            ++li;
        }
    }

    // ---------------------------------------------------------------- recurrent step + heads
    // One recurrent step in two halves (RnnW: the input row is [x_early | h*mask | x_late]):
    //   rnn_pre    h*mask into the row, then the gate pre-activations of the first early+H columns (+ bias) -- needs only the
    //              encoders' own projections and the previous state, so callers run it beside the cross-modal block;
    //   rnn_finish adds the late columns' contribution and runs the cell (+ fused heads).
    float* rnn_pre(const RnnW& w, float* xh, int ld, int B, const float* h_in, const float* mask) {
        const int H = ctx->cfg.hidden;
        if (!dry) ck(launch_rnn_prep(h_in, mask, xh, B, H, ld, w.early, s), "rnn_prep");
        if (ctx->cfg.rnn_type != HCM_LSTM) return nullptr;
        float* gates = alloc_f((size_t)B * 4 * H);
        LinW first = w.cat;
        first.K = w.early + H;
        linear(first, xh, B, ld, gates, 4 * H, ACT_NONE, true);
        return gates;
    }
    void rnn_finish(const RnnW& w, float* xh, int ld, int B, float* gates, const float* h_in, const float* mask, float* h_out,
                    const Heads& heads_in) {
        const int H = ctx->cfg.hidden;
        Heads heads = heads_in;
        heads.bad = ctx->calib_buf ? ctx->calib_buf + hcm_ctx::kStepBadWord : nullptr;      // overflow guard of every recurrent step (hcm_query(HCM_STEP_NONFINITE))
        if (ctx->cfg.rnn_type == HCM_LSTM) {
            if (w.early < w.in) {
                LinW late = w.cat;
                late.w = (char*)w.cat.w + (size_t)(w.early + H) * 4;       // f32 weights: column offset inside every row
                late.K = w.in - w.early;
                late.bias = nullptr;
                linear(late, xh + w.early + H, B, ld, gates, 4 * H, ACT_NONE, true, gates, 4 * H);
            }
            if (!dry) ck(launch_lstm_cell(gates, h_in, mask, h_out, B, H, heads, s), "lstm_cell");
        } else {
            float* gi = alloc_f((size_t)B * 3 * H);
            float* gh = alloc_f((size_t)B * 3 * H);
            linear(w.ih, xh, B, ld, gi, 3 * H, ACT_NONE, true);
            linear(w.hh, xh + w.early, B, ld, gh, 3 * H, ACT_NONE, true);
            if (!dry) ck(launch_gru_cell(gi, gh, h_in, mask, h_out, B, H, heads, s), "gru_cell");
        }
    }
    void rnn_step(const RnnW& w, float* xh, int ld, int B, const float* h_in, const float* mask, float* h_out, const Heads& heads) {
        float* gates = rnn_pre(w, xh, ld, B, h_in, mask);
        rnn_finish(w, xh, ld, B, gates, h_in, mask, h_out, heads);
    }

    // RNNStateEncoder.forward (models/decoder/state_encoder.py:135-137): single_forward when the feature batch equals
    // the hidden batch, else seq_forward (:83-133) -- T steps over (T*N) feature rows with per-step masking (the
    // reference masks at segment starts only; inside a segment every mask is 1, so per-step masking is identical).
    void rnn_scan(const RnnW& w, float* xh, int ld, int T, int N, const float* h_in, const float* mask, float* h_out, const Heads& heads) {
        if (T == 1) { rnn_step(w, xh, ld, N, h_in, mask, h_out, heads); return; }
        const int R = ctx->cfg.rnn_type == HCM_LSTM ? 2 : 1;
        float* pp[2] = {alloc_f((size_t)R * N * ctx->cfg.hidden), alloc_f((size_t)R * N * ctx->cfg.hidden)};
        const float* cur = h_in;
        for (int t = 0; t < T; ++t) {
            float* dst = t == T - 1 ? h_out : pp[t & 1];
            Heads hd = heads;
            if (hd.out0) hd.out0 += (size_t)t * N * hd.ld0;
            if (hd.out1) hd.out1 += (size_t)t * N * hd.ld1;
            // the steps run in order on one stream: every step re-uses the same gate / split-K scratch (bounded by the dry runs at
            // T = 1..3 of hcm_finalize instead of growing with T)
            const size_t m = ar.mark();
            rnn_step(w, xh + (size_t)t * N * ld, ld, N, cur, mask + (size_t)t * N, dst, hd);
            ar.release(m);
            cur = dst;
        }
    }

    // rnn_in tap: the logical input x (without the h*mask block that sits between its early and late columns)
    void tap_rnn_in(const std::string& name, const RnnW& w, const float* xh, int ld, int B) {
        if (dry || !ctx->taps_on) return;
        float* tmp = nullptr;
        ck(hipMalloc((void**)&tmp, (size_t)B * w.in * 4), "tap scratch");
        const int H = ctx->cfg.hidden;
        ck(hipMemcpy2DAsync(tmp, (size_t)w.in * 4, xh, (size_t)ld * 4, (size_t)w.early * 4, B, hipMemcpyDeviceToDevice, s), "tap gather");
        if (w.early < w.in)
            ck(hipMemcpy2DAsync(tmp + w.early, (size_t)w.in * 4, xh + w.early + H, (size_t)ld * 4, (size_t)(w.in - w.early) * 4, B, hipMemcpyDeviceToDevice, s), "tap gather");
        tap(name, tmp, false, {B, w.in});
        ck(hipStreamSynchronize(s), "tap sync");
        (void)hipFree(tmp);
    }

    int T = 1;      // time steps packed in the batch (training / validation path); 1 = the per-step rollout call

    // ---------------------------------------------------------------- stages of Seq2Seq_HighLevel_CMA.forward
    struct HiBufs {
        void* rgb_tok = nullptr;   // (B,2112,16) of the reference, token-major [B][16][2112]      (dt_vla)
        void* dep_tok = nullptr;   // [B][S][dC]                                                    (dt_vla)
        void* emb = nullptr;       // BERT last hidden state [B][L][768]                            (dt_bert)
        float* xh = nullptr;       // [rgb_in | depth_in | ins_rgb | ins_depth | h*mask]  f32
        int ldx = 0;
        // produced inside the encoder chains (each depends on ONE encoder only), consumed by hi_tail:
        void* I = nullptr;         // LN(ReLU(ins_fc(bert))) + PE   [B*L][d]     (BERT chain)
        std::vector<void*> Q;      // per-layer fc_q(I)                            (BERT chain)
        void* kvin[2] = {nullptr, nullptr};   // LN(ReLU(vis_fc(rgb_kv / depth_kv(tokens))))  [B*max(S,L)][d]  (RGB / depth chain)
        void* kv0[2] = {nullptr, nullptr};    // layer-0 fc_k|fc_v of kvin        [B*max(S,L)][2d]
    };
    struct LoBufs { float* xh = nullptr; int ldx = 0; };   // [depth | rgb | h*mask | subtask] (seq2seq_lowlevel.py:143; RnnW layout)
    // the low-level model's recurrent input, handed to hi_tail so that its early gate GEMM can run beside the cross-modal block
    LoBufs* lo_early = nullptr;
    const float* lo_h_in_early = nullptr;
    float* lo_pre = nullptr;
    bool pred_fused = false;          // argmax + sub-task embedding done by the high-level cell kernel (hi_tail)


    HiBufs hi_alloc(int B) {
        const hcm_config& c = ctx->cfg;
        const HighW& w = ctx->hi;
        HiBufs b;
        use(ctx->dt_vla);
        b.rgb_tok = alloc_t((size_t)B * 16 * (2048 + 64));
        b.dep_tok = alloc_t((size_t)B * w.depth_S * w.depth_C);
        use(ctx->dt_bert);
        b.emb = alloc_t((size_t)B * c.instr_len * c.bert_hidden);
        b.ldx = w.rnn.in + c.hidden;
        b.xh = alloc_f((size_t)B * b.ldx);
        use(ctx->dt_vla);
        const int L = c.instr_len, d = c.d_model;
        b.I = alloc_t((size_t)B * L * d);
        b.Q.resize(w.vla.layers.size());
        for (auto& q : b.Q) q = alloc_t((size_t)B * L * d);
        for (int st = 0; st < 2; ++st) {
            const int S = st == 0 ? 16 : w.depth_S;
            b.kvin[st] = alloc_t((size_t)B * (S > L ? S : L) * d);
            b.kv0[st] = alloc_t((size_t)B * (S > L ? S : L) * 2 * d);
        }
        return b;
    }

    // The part of Visual_Ling_Attn that depends on ONE visual encoder only (transformer.py:258-262 + the layer-0 key/value
    // projection) and the encoder's own projection into the recurrent input (seq2seq_highlevel_cma.py:198-199,:213-214):
    // enqueued at the end of that encoder's chain, so it overlaps the other chains instead of sitting in the serial tail.
    void hi_vis_pre(int stream, int B, HiBufs& hb) {
        const hcm_config& c = ctx->cfg;
        const HighW& w = ctx->hi;
        const VlaW& v = w.vla;
        const int d = c.d_model, rC = 2048 + 64, dS = w.depth_S, dC = w.depth_C;
        use(ctx->dt_vla);
        calib_slot = 3;                       // range hooks of the cross-modal block's GEMM outputs (calibration forward)
        const size_t m = ar.mark();
        const int S = stream == 0 ? 16 : dS;
        const void* tok = stream == 0 ? hb.rgb_tok : hb.dep_tok;
        const int tokC = stream == 0 ? rC : dC;
        const LinW& kvproj = stream == 0 ? w.rgb_kv : w.depth_kv;
        void* vis = alloc_t((size_t)B * S * c.vis_in);
        linear(kvproj, tok, B * S, tokC, vis, c.vis_in, ACT_NONE, false);              // rgb_kv / depth_kv Conv1d(k=1)
        tap(stream == 0 ? "hi.rgb_kv" : "hi.depth_kv", vis, true, {B, S, c.vis_in});
        void* vtmp = alloc_t((size_t)B * S * d);
        linear(v.vis_fc, vis, B * S, c.vis_in, vtmp, d, ACT_RELU, false);
        ln(vtmp, nullptr, v.ln, nullptr, 0, hb.kvin[stream], B * S, d, 1e-5f);
        linear(v.layers[0].kv, hb.kvin[stream], B * S, d, hb.kv0[stream], 2 * d, ACT_NONE, false);
        if (stream == 0) {
            // rgb_linear (:213): mean over the 16 tokens -> Linear -> ReLU
            void* rmean = alloc_t((size_t)B * rC);
            if (!dry) ck(launch_mean_rows(hb.rgb_tok, rmean, dt, B, 16, rC, rC, rC, 0, s), "rgb mean");
            linear(w.rgb_linear, rmean, B, rC, hb.xh, hb.ldx, ACT_RELU, true);
        } else {
            // depth_linear (:214): Flatten -> Linear -> ReLU
            linear(w.depth_linear, hb.dep_tok, B, dS * dC, hb.xh + c.rgb_out, hb.ldx, ACT_RELU, true);
        }
        // NOTE: no ar.release(m): chains run concurrently and allocate from one bump arena
        (void)m;
        calib_slot = -1;
    }
    // The instruction stream of Visual_Ling_Attn (transformer.py:263-269): identical for both calls, depends on BERT only.
    void hi_ins_pre(int B, HiBufs& hb) {
        const hcm_config& c = ctx->cfg;
        const VlaW& v = ctx->hi.vla;
        const int L = ctx->cur_L, d = c.d_model, rows = B * L;
        const size_t rmax = (size_t)B * c.instr_len;
        use(ctx->dt_vla);
        calib_slot = 3;
        void* emb = hb.emb;
        if (ctx->dt_bert != ctx->dt_vla) {
            void* e2 = alloc_t(rmax * c.bert_hidden);
            if (!dry) ck(launch_convert(emb, ctx->dt_bert, e2, ctx->dt_vla, (size_t)rows * c.bert_hidden, s), "bert out convert");
            emb = e2;
        }
        void* tmp = alloc_t(rmax * d);
        linear(v.ins_fc, emb, rows, c.bert_hidden, tmp, d, ACT_RELU, false);
        ln(tmp, nullptr, v.ln, v.pe, L, hb.I, rows, d, 1e-5f);     // LN then + PE; identical for both calls -> computed once
        for (size_t l = 0; l < v.layers.size(); ++l) linear(v.layers[l].q, hb.I, rows, d, hb.Q[l], d, ACT_NONE, false);
        calib_slot = -1;
    }
    LoBufs lo_alloc(int B) {
        LoBufs b;
        b.ldx = ctx->lo.rnn.in + ctx->cfg.hidden;
        b.xh = alloc_f((size_t)B * b.ldx);
        return b;
    }

    // depth_encoder (seq2seq_highlevel_cma.py:178-179): GN-ResNet50 + pos-emb channels -> dep_tok
    void hi_depth(const float* depth, int B, HiBufs& hb) {
        const HighW& w = ctx->hi;
        const int dS = w.depth_S, dC = w.depth_C;
        use(ctx->dt_depth);
        Act o = depth_trunk(w.depth, depth, B, "hi.depth");
        void* tok = ctx->dt_depth == ctx->dt_vla ? hb.dep_tok : alloc_t((size_t)B * dS * dC);
        if (!dry) {
            ck(launch_adaptive_avgpool(o.p, tok, dt, B, o.H, o.W, o.C, o.H, o.W, dC, s), "depth tokens");
            ck(launch_fill_cols(w.depth_pe, (char*)tok + (size_t)o.C * esz, dt, B, dS, 64, dC, s), "depth pe");
            if (tok != hb.dep_tok) ck(launch_convert(tok, ctx->dt_depth, hb.dep_tok, ctx->dt_vla, (size_t)B * dS * dC, s), "depth tokens convert");
        }
        use(ctx->dt_vla);
        tap("hi.depth_spatial", hb.dep_tok, true, {B, dS, dC});
        hi_vis_pre(1, B, hb);
    }
    // Both depth encoders in one pass over the shared frame (hcm_act): channel-concatenated pair trunk, then the hi half
    // becomes dep_tok (+ pos-emb) and the lo half goes through visual_fc (resnet_encoders.py:56-62,:108) into the lo RNN input.
    void depth_pair(const float* depth, int B, HiBufs& hb, LoBufs& lb) {
        const HighW& w = ctx->hi;
        const int dS = w.depth_S, dC = w.depth_C;
        use(ctx->dt_depth);
        Act o = depth_trunk(w.depth_pair, depth, B, "pair.depth");        // [B, fs, fs, 2*cc]
        const int cc = o.C / 2;
        void* tok = ctx->dt_depth == ctx->dt_vla ? hb.dep_tok : alloc_t((size_t)B * dS * dC);
        void* lo_feat = alloc_t((size_t)B * dS * cc);
        if (!dry) {
            ck(launch_adaptive_avgpool(o.p, tok, dt, B, o.H, o.W, cc, o.H, o.W, dC, s, o.C), "depth tokens (hi half)");
            ck(launch_fill_cols(w.depth_pe, (char*)tok + (size_t)cc * esz, dt, B, dS, 64, dC, s), "depth pe");
            if (tok != hb.dep_tok) ck(launch_convert(tok, ctx->dt_depth, hb.dep_tok, ctx->dt_vla, (size_t)B * dS * dC, s), "depth tokens convert");
            ck(launch_adaptive_avgpool((char*)o.p + (size_t)cc * esz, lo_feat, dt, B, o.H, o.W, cc, o.H, o.W, cc, s, o.C), "depth lo half");
        }
        linear(ctx->lo.depth_fc, lo_feat, B, dS * cc, lb.xh, lb.ldx, ACT_RELU, true);     // visual_fc
        use(ctx->dt_vla);
        tap("hi.depth_spatial", hb.dep_tok, true, {B, dS, dC});
        hi_vis_pre(1, B, hb);
    }
    // Both RGB encoders in one pass (hcm_act): hi half -> adaptive_avg_pool2d(4,4) + pos-emb -> rgb_tok; lo half -> global
    // average pool -> fc -> ReLU (resnet_encoders.py:234-237) into the lo RNN input.
    void rgb_pair(const void* rgb, int rgb_dt, int B, HiBufs& hb, LoBufs& lb) {
        const HighW& w = ctx->hi;
        const hcm_config& c = ctx->cfg;
        const int rC = 2048 + 64;
        use(ctx->dt_rgb);
        Act o = rgb_trunk(w.rgb_pair, rgb, rgb_dt, B, "pair.rgb");            // [B, h, w, 2*2048]
        const int C1 = o.C / 2;
        void* tok = ctx->dt_rgb == ctx->dt_vla ? hb.rgb_tok : alloc_t((size_t)B * 16 * rC);
        void* pooled = alloc_t((size_t)B * C1);
        if (!dry) {
            ck(launch_adaptive_avgpool(o.p, tok, dt, B, o.H, o.W, C1, 4, 4, rC, s, o.C), "rgb tokens (hi half)");
            ck(launch_fill_cols(w.rgb_pe, (char*)tok + (size_t)C1 * esz, dt, B, 16, 64, rC, s), "rgb pe");
            if (tok != hb.rgb_tok) ck(launch_convert(tok, ctx->dt_rgb, hb.rgb_tok, ctx->dt_vla, (size_t)B * 16 * rC, s), "rgb tokens convert");
            ck(launch_adaptive_avgpool((char*)o.p + (size_t)C1 * esz, pooled, dt, B, o.H, o.W, C1, 1, 1, C1, s, o.C), "global avgpool (lo half)");
        }
        linear(ctx->lo.rgb_fc, pooled, B, C1, lb.xh + c.depth_out, lb.ldx, ACT_RELU, true);
        use(ctx->dt_vla);
        tap("hi.rgb_spatial", hb.rgb_tok, true, {B, 16, rC});
        hi_vis_pre(0, B, hb);
    }
    // Trunk weights identical in both models (HighW::rgb_shared / depth_shared): one trunk pass, both models' heads
    void rgb_shared(const void* rgb, int rgb_dt, int B, HiBufs& hb, LoBufs& lb) {
        const HighW& w = ctx->hi;
        const int rC = 2048 + 64;
        use(ctx->dt_rgb);
        Act o = rgb_trunk(w.rgb, rgb, rgb_dt, B, "hi.rgb");
        void* tok = ctx->dt_rgb == ctx->dt_vla ? hb.rgb_tok : alloc_t((size_t)B * 16 * rC);
        void* pooled = alloc_t((size_t)B * o.C);
        if (!dry) {
            ck(launch_adaptive_avgpool(o.p, tok, dt, B, o.H, o.W, o.C, 4, 4, rC, s), "rgb tokens");
            ck(launch_fill_cols(w.rgb_pe, (char*)tok + (size_t)2048 * esz, dt, B, 16, 64, rC, s), "rgb pe");
            if (tok != hb.rgb_tok) ck(launch_convert(tok, ctx->dt_rgb, hb.rgb_tok, ctx->dt_vla, (size_t)B * 16 * rC, s), "rgb tokens convert");
            ck(launch_adaptive_avgpool(o.p, pooled, dt, B, o.H, o.W, o.C, 1, 1, o.C, s), "global avgpool (low-level head)");
        }
        linear(ctx->lo.rgb_fc, pooled, B, o.C, lb.xh + ctx->cfg.depth_out, lb.ldx, ACT_RELU, true);
        use(ctx->dt_vla);
        tap("hi.rgb_spatial", hb.rgb_tok, true, {B, 16, rC});
        hi_vis_pre(0, B, hb);
    }
    void depth_shared(const float* depth, int B, HiBufs& hb, LoBufs& lb) {
        const HighW& w = ctx->hi;
        const int dS = w.depth_S, dC = w.depth_C;
        use(ctx->dt_depth);
        Act o = depth_trunk(w.depth, depth, B, "hi.depth");
        void* tok = ctx->dt_depth == ctx->dt_vla ? hb.dep_tok : alloc_t((size_t)B * dS * dC);
        if (!dry) {
            ck(launch_adaptive_avgpool(o.p, tok, dt, B, o.H, o.W, o.C, o.H, o.W, dC, s), "depth tokens");
            ck(launch_fill_cols(w.depth_pe, (char*)tok + (size_t)o.C * esz, dt, B, dS, 64, dC, s), "depth pe");
            if (tok != hb.dep_tok) ck(launch_convert(tok, ctx->dt_depth, hb.dep_tok, ctx->dt_vla, (size_t)B * dS * dC, s), "depth tokens convert");
        }
        linear(ctx->lo.depth_fc, o.p, B, o.H * o.W * o.C, lb.xh, lb.ldx, ACT_RELU, true);     // visual_fc of the low-level model
        use(ctx->dt_vla);
        tap("hi.depth_spatial", hb.dep_tok, true, {B, dS, dC});
        hi_vis_pre(1, B, hb);
    }
    // rgb_encoder (:180-181): ResNet50 trunk, adaptive_avg_pool2d(4,4), pos-emb channels -> rgb_tok
    void hi_rgb(const void* rgb, int rgb_dt, int B, HiBufs& hb) {
        const HighW& w = ctx->hi;
        const int rC = 2048 + 64;
        use(ctx->dt_rgb);
        Act o = rgb_trunk(w.rgb, rgb, rgb_dt, B, "hi.rgb");
        void* tok = ctx->dt_rgb == ctx->dt_vla ? hb.rgb_tok : alloc_t((size_t)B * 16 * rC);
        if (!dry) {
            ck(launch_adaptive_avgpool(o.p, tok, dt, B, o.H, o.W, o.C, 4, 4, rC, s), "rgb tokens");
            ck(launch_fill_cols(w.rgb_pe, (char*)tok + (size_t)2048 * esz, dt, B, 16, 64, rC, s), "rgb pe");
            if (tok != hb.rgb_tok) ck(launch_convert(tok, ctx->dt_rgb, hb.rgb_tok, ctx->dt_vla, (size_t)B * 16 * rC, s), "rgb tokens convert");
        }
        use(ctx->dt_vla);
        tap("hi.rgb_spatial", hb.rgb_tok, true, {B, 16, rC});
        hi_vis_pre(0, B, hb);
    }
    // ablate_depth / ablate_rgb (seq2seq_highlevel_cma.py:185-188, seq2seq_lowlevel.py:132-135): `embedding * 0` right after the
    // encoder -- for the high-level model the whole (B, C+64, S) token tensor including its positional-embedding channels, for the
    // low-level model the encoder's feature vector.  The encoder's value cannot reach any output (finite x * 0 = 0), so the trunk is
    // not run; everything downstream (rgb_kv / depth_kv, Visual_Ling_Attn, the projections) runs on the zeros as in the reference.
    void ablated_encoder(int stream, int B, HiBufs* hb, LoBufs* lb) {
        const hcm_config& c = ctx->cfg;
        const HighW& w = ctx->hi;
        use(ctx->dt_vla);
        if (hb) {
            void* tok = stream == 0 ? hb->rgb_tok : hb->dep_tok;
            const size_t n = stream == 0 ? (size_t)B * 16 * (2048 + 64) : (size_t)B * w.depth_S * w.depth_C;
            if (!dry) ck(hipMemsetAsync(tok, 0, n * esz, s), "ablate: zero tokens");
            tap(stream == 0 ? "hi.rgb_spatial" : "hi.depth_spatial", tok, true,
                stream == 0 ? Shape{B, 16, 2048 + 64} : Shape{B, w.depth_S, w.depth_C});
            hi_vis_pre(stream, B, *hb);
        }
        if (lb && !dry) {
            float* col = lb->xh + (stream == 0 ? c.depth_out : 0);
            const int ncol = stream == 0 ? c.rgb_out : c.depth_out;
            ck(hipMemset2DAsync(col, (size_t)lb->ldx * 4, 0, (size_t)ncol * 4, B, s), "ablate: zero features");
        }
    }
    // BERT (:189-195) -> emb
    void hi_bert(const void* ids, int ids_dt, int B, HiBufs& hb) {
        use(ctx->dt_bert);
        calib_slot = 0;
        bert(ctx->hi.bert, ids, ids_dt, B, hb.emb, ctx->cur_lens);
        calib_slot = -1;
        tap("hi.bert", hb.emb, true, {B, ctx->cur_L, ctx->cfg.bert_hidden});
        mark("bert.end");
        hi_ins_pre(B, hb);
        mark("bert.ins_pre_end");
    }
    // Visual_Ling_Attn x2 (the cross-modal part), poolers, state encoder, head (:200-232)
    void hi_tail(int B, HiBufs& hb, const float* h_in, const float* mask, float* logits, int ld_logits, float* h_out) {
        const hcm_config& c = ctx->cfg;
        const HighW& w = ctx->hi;
        const int L = ctx->cur_L, Lm = c.instr_len, d = c.d_model;
        const int dS = w.depth_S;
        const int ldx = hb.ldx;
        float* xh = hb.xh;
        use(ctx->dt_vla);
        calib_slot = 3;                       // (the recurrent GEMMs behind the block are fp32: their hooks are no-ops)
        const VlaW& v = w.vla;
        const int rows = B * L;
        void* I = hb.I;
        // Fused form (16-bit cross-modal block): ONE launch per layer for both Visual_Ling_Attn calls does attention (layer 0: the keys are
        // the few visual tokens) + fc_o + residual + LayerNorm + FFN + residual + LayerNorm (+ the cross_pooler mean on the last layer) with
        // the block's activations resident in LDS (vla_fused.hip); deeper layers keep their key/value projection and the L x L attention as
        // launches of their own.  HCM_NO_VLA_FUSE=1 selects the launch-per-op form below (A/B and the toggle test).
        static const bool no_vla_fuse = dev_env("HCM_NO_VLA_FUSE") != nullptr;
        const bool fused = !no_vla_fuse && vla_post_ok(dt, d, c.vla_heads, c.d_ff);
        const bool fork = ctx->concurrent && !ctx->taps_on;
        hipStream_t main_s = s;
        if (fused) {
            if (fork) fork_join_begin(2);
            void* outb[2][2];
            void* attb[2] = {nullptr, nullptr};
            void* kvb[2] = {nullptr, nullptr};
            for (int st = 0; st < 2; ++st) {
                outb[st][0] = alloc_t((size_t)B * Lm * d);
                outb[st][1] = alloc_t((size_t)B * Lm * d);
                if (v.layers.size() > 1) { attb[st] = alloc_t((size_t)B * Lm * d); kvb[st] = alloc_t((size_t)B * Lm * 2 * d); }
            }
            const int S2[2] = {16, dS};
            const bool pool_in_kernel = L <= 80;
            for (size_t l = 0; l < v.layers.size(); ++l) {
                const VlaLayerW& ly = v.layers[l];
                const bool last = l + 1 == v.layers.size();
                VlaPost q;
                q.q = hb.Q[l]; q.I = I; q.B = B; q.L = L; q.d_ff = c.d_ff; q.lens = ctx->cur_lens;
                q.wo = ly.o.w; q.bo = ly.o.bias; q.w1 = ly.ff1.w; q.b1 = ly.ff1.bias; q.w2 = ly.ff2.w; q.b2 = ly.ff2.bias;
                q.g1 = ly.ln_att.gamma; q.be1 = ly.ln_att.beta; q.g2 = ly.ln_ff.gamma; q.be2 = ly.ln_ff.beta;
                // round 6: the layer's weights in fragment order, read straight into registers (bit-identical; HCM_NO_VLA_WFRAG=1 of the development
                // build: the LDS weight ring, for the A/B and the toggle test)
                static const bool no_wfrag = dev_env("HCM_NO_VLA_WFRAG") != nullptr;
                if (!no_wfrag && ly.o_f && ly.ff1_f && ly.ff2_f) { q.wo = ly.o_f; q.w1 = ly.ff1_f; q.w2 = ly.ff2_f; q.wfrag = 1; }
                q.fuse_att = l == 0 && S2[0] <= 32 && S2[1] <= 32;
                for (int st = 0; st < 2; ++st) {
                    const int Lk = l == 0 ? S2[st] : L;
                    const void* kvl = hb.kv0[st];
                    if (l > 0) { linear(ly.kv, outb[st][(l - 1) & 1], B * L, d, kvb[st], 2 * d, ACT_NONE, false); kvl = kvb[st]; }
                    if (!q.fuse_att) {
                        void* att = attb[st] ? attb[st] : (attb[st] = alloc_t((size_t)B * Lm * d));
                        if (!dry) ck(launch_attention(hb.Q[l], kvl, (const char*)kvl + (size_t)d * esz, att, dt, B, c.vla_heads, L, Lk, d, 2 * d, 2 * d, d, B, s,
                                                      l > 0 ? ctx->cur_lens : nullptr), "vla attention");
                        q.att[st] = att;
                    }
                    q.kv[st] = kvl; q.Lk[st] = Lk;
                    q.out[st] = outb[st][l & 1];
                    if (last && pool_in_kernel) { q.pooled[st] = xh + w.rnn.xcol(c.rgb_out + c.depth_out + st * d); q.ld_pool = ldx; }
                }
                if (ctx->calib && dt == DT_F16) q.calib = ctx->calib_buf + 2 * 3;      // the kernel's LDS-only intermediates are range-checked inside it
                if (!dry) ck(launch_vla_post(q, dt, s), "fused cross-modal layer");
            }
            const size_t lastl = (v.layers.size() - 1) & 1;
            for (int st = 0; st < 2; ++st) {
                tap(st == 0 ? "hi.vla_rgb" : "hi.vla_depth", outb[st][lastl], true, {B, L, d});
                if (!pool_in_kernel && !dry)
                    ck(launch_mean_rows(outb[st][lastl], xh + w.rnn.xcol(c.rgb_out + c.depth_out + st * d), dt, B, L, d, d, ldx, 1, s, ctx->cur_lens), "cross_pooler");
            }
        } else {
        // the two Visual_Ling_Attn calls (rgb, depth) are independent until the recurrent input: run them on two streams
        if (fork) fork_join_begin(2);
        for (int stream = 0; stream < 2; ++stream) {
            if (fork) on(stream == 0 ? main_s : ctx->aux[0]);
            const int S = stream == 0 ? 16 : dS;
            int Lk = S;
            void* kv = alloc_t((size_t)B * (S > Lm ? S : Lm) * 2 * d);
            void* att = alloc_t((size_t)B * Lm * d);
            void* t2 = alloc_t((size_t)B * Lm * d);
            void* ffh = alloc_t((size_t)B * Lm * c.d_ff);
            void* out = alloc_t((size_t)B * Lm * d);
            void* kvin = hb.kvin[stream];
            for (size_t l = 0; l < v.layers.size(); ++l) {
                const VlaLayerW& ly = v.layers[l];
                const void* kvl = hb.kv0[stream];                                          // layer 0: projected in the encoder chain
                if (l > 0) { linear(ly.kv, kvin, B * Lk, d, kv, 2 * d, ACT_NONE, false); kvl = kv; }
                // layer 0 attends over the visual tokens; deeper layers over the previous layer's (B, L, d) output, of which a ragged
                // batch's sample owns the first lengths[b] rows only
                if (!dry) ck(launch_attention(hb.Q[l], kvl, (const char*)kvl + (size_t)d * esz, att, dt, B, c.vla_heads, L, Lk, d, 2 * d, 2 * d, d, B, s,
                                              l > 0 ? ctx->cur_lens : nullptr), "vla attention");
                linear(ly.o, att, rows, d, t2, d, ACT_NONE, false, I, d);                 // queries + att
                ln(t2, nullptr, ly.ln_att, nullptr, 0, att, rows, d, 1e-5f);             // MultiHeadAttention.layer_norm
                linear(ly.ff1, att, rows, d, ffh, c.d_ff, ACT_RELU, false);
                linear(ly.ff2, ffh, rows, c.d_ff, t2, d, ACT_NONE, false, att, d);
                ln(t2, nullptr, ly.ln_ff, nullptr, 0, out, rows, d, 1e-5f);
                if (l + 1 < v.layers.size()) {
                    // next layer attends over this layer's output (B,L,d)
                    if (!dry) ck(hipMemcpyAsync(kvin, out, (size_t)rows * d * esz, hipMemcpyDeviceToDevice, s), "vla copy");
                    Lk = L;
                }
            }
            tap(stream == 0 ? "hi.vla_rgb" : "hi.vla_depth", out, true, {B, L, d});
            // cross_pooler: mean over all L tokens (:209-210) -> xh columns
            if (!dry) ck(launch_mean_rows(out, xh + w.rnn.xcol(c.rgb_out + c.depth_out + stream * d), dt, B, L, d, d, ldx, 1, s, ctx->cur_lens), "cross_pooler");
        }
        }
        // meanwhile (third stream): the early halves of both recurrent steps
        float* hi_pre = nullptr;
        const bool split = T == 1;
        if (split) {
            if (fork) on(ctx->aux[1]);
            hi_pre = rnn_pre(w.rnn, xh, ldx, B, h_in, mask);
            if (lo_early) lo_pre = rnn_pre(ctx->lo.rnn, lo_early->xh, lo_early->ldx, B, lo_h_in_early, mask);
        }
        if (fork) { on(main_s); fork_join_end(2); }
        // state_encoder (:219) + linear head (:232)
        Heads hd;
        hd.w0 = w.head_w; hd.b0 = w.head_b; hd.out0 = logits; hd.r0 = c.num_actions; hd.ld0 = ld_logits;
        static const bool no_pred_fuse = dev_env("HCM_NO_PRED_FUSE") != nullptr;
        pred_fused = false;
        if (split && lo_early && !no_pred_fuse && c.num_actions <= 64) {
            // fused act(): argmax + the low-level model's sub-task embedding lookup ride in the high-level cell kernel
            const LowW& lw = ctx->lo;
            hd.pred = ctx->pred_buf; hd.emb = lw.subtask_emb; hd.emb_dim = 32; hd.emb_ld = lo_early->ldx; hd.emb_rows = c.num_sub_tasks + 1;
            hd.emb_out = lo_early->xh + lw.rnn.xcol(c.depth_out + c.rgb_out);
            pred_fused = true;
        }
        if (split) rnn_finish(w.rnn, xh, ldx, B, hi_pre, h_in, mask, h_out, hd);
        else rnn_scan(w.rnn, xh, ldx, T, B / T, h_in, mask, h_out, hd);
        tap_rnn_in("hi.rnn_in", w.rnn, xh, ldx, B);
        calib_slot = -1;
    }

    // ---------------------------------------------------------------- stages of Seq2Seq_LowLevel.forward
    void lo_depth(const float* depth, int B, LoBufs& lb) {
        const LowW& w = ctx->lo;
        use(ctx->dt_depth);
        if (w.depth_simple) {
            calib_slot = 1;
            simple_cnn(w.depth_s, depth, DT_F32, 1.0f, B, lb.xh, lb.ldx);
            calib_slot = -1;
        } else {
            Act o = depth_trunk(w.depth, depth, B, "lo.depth");
            linear(w.depth_fc, o.p, B, o.H * o.W * o.C, lb.xh, lb.ldx, ACT_RELU, true);     // visual_fc
        }
    }
    void lo_rgb(const void* rgb, int rgb_dt, int B, LoBufs& lb) {
        const LowW& w = ctx->lo;
        const hcm_config& c = ctx->cfg;
        use(ctx->dt_rgb);
        if (w.rgb_simple) {
            calib_slot = 2;
            simple_cnn(w.rgb_s, rgb, rgb_dt, 1.0f / 255.0f, B, lb.xh + c.depth_out, lb.ldx);
            calib_slot = -1;
        } else {
            Act o = rgb_trunk(w.rgb, rgb, rgb_dt, B, "lo.rgb");
            void* pooled = alloc_t((size_t)B * o.C);
            if (!dry) ck(launch_adaptive_avgpool(o.p, pooled, dt, B, o.H, o.W, o.C, 1, 1, o.C, s), "global avgpool");
            linear(w.rgb_fc, pooled, B, o.C, lb.xh + c.depth_out, lb.ldx, ACT_RELU, true);
        }
    }
    void lo_tail(int B, LoBufs& lb, const float* h_in, const float* mask, const int64_t* subtask, float* vel, int ld_vel,
                 float* stop, int ld_stop, float* h_out) {
        const hcm_config& c = ctx->cfg;
        const LowW& w = ctx->lo;
        use(ctx->dt_vla);
        if (!dry && !(pred_fused && subtask == ctx->pred_buf))
            ck(launch_embed_rows(w.subtask_emb, subtask, lb.xh, B, 32, lb.ldx, w.rnn.xcol(c.depth_out + c.rgb_out), c.num_sub_tasks + 1, s), "subtask emb");
        Heads hd;
        hd.w0 = w.lin_w; hd.b0 = w.lin_b; hd.out0 = vel; hd.r0 = c.lo_actions; hd.ld0 = ld_vel;
        hd.w1 = w.stop_w; hd.b1 = w.stop_b; hd.out1 = stop; hd.r1 = 1; hd.ld1 = ld_stop;
        if (lo_pre) rnn_finish(w.rnn, lb.xh, lb.ldx, B, lo_pre, h_in, mask, h_out, hd);     // early half done beside the high-level tail
        else rnn_scan(w.rnn, lb.xh, lb.ldx, T, B / T, h_in, mask, h_out, hd);
        tap_rnn_in("lo.rnn_in", w.rnn, lb.xh, lb.ldx, B);
    }

    // hcm_refresh_instruction: recompute the cached instruction stream (hb.I, hb.Q of the last B-sized step) for the listed
    // environments only -- BERT + ins_fc/LN/PE + fc_q at batch n, rows copied into place
    void refresh_instruction(const void* ids, int ids_dt, int B, const int32_t* idx, int n) {
        const hcm_config& c = ctx->cfg;
        ar.reset();
        HiBufs hb = hi_alloc(B);                                  // same offsets as in step(): the persistent tensors
        const int L = ctx->cur_L, d = c.d_model;
        const size_t idsz = ids_dt == DT_I64 ? 8 : 4;
        char* sub_ids = (char*)ar.alloc((size_t)n * c.instr_len * idsz);
        int* sub_lens = (ctx->cur_lens || dry) ? (int*)ar.alloc((size_t)n * sizeof(int)) : nullptr;
        HiBufs hn;
        use(ctx->dt_bert);
        hn.emb = alloc_t((size_t)n * c.instr_len * c.bert_hidden);
        use(ctx->dt_vla);
        hn.I = alloc_t((size_t)n * c.instr_len * d);
        hn.Q.resize(hb.Q.size());
        for (auto& q : hn.Q) q = alloc_t((size_t)n * c.instr_len * d);
        if (!dry) {
            for (int i = 0; i < n; ++i)
                ck(hipMemcpyAsync(sub_ids + (size_t)i * L * idsz, (const char*)ids + (size_t)idx[i] * L * idsz, (size_t)L * idsz, hipMemcpyDeviceToDevice, s), "ids gather");
            if (ctx->cur_lens)
                for (int i = 0; i < n; ++i)
                    ck(hipMemcpyAsync(sub_lens + i, ctx->cur_lens + idx[i], sizeof(int), hipMemcpyDeviceToDevice, s), "lengths gather");
        }
        use(ctx->dt_bert);
        bert(ctx->hi.bert, sub_ids, ids_dt, n, hn.emb, ctx->cur_lens ? sub_lens : nullptr);
        hi_ins_pre(n, hn);
        use(ctx->dt_vla);
        if (dry) return;                                          // (hcm_finalize's sizing pass: allocations only)
        const size_t row = (size_t)L * d * esz;
        for (int i = 0; i < n; ++i) {
            ck(hipMemcpyAsync((char*)hb.I + idx[i] * row, (char*)hn.I + i * row, row, hipMemcpyDeviceToDevice, s), "I scatter");
            for (size_t l = 0; l < hb.Q.size(); ++l)
                ck(hipMemcpyAsync((char*)hb.Q[l] + idx[i] * row, (char*)hn.Q[l] + i * row, row, hipMemcpyDeviceToDevice, s), "Q scatter");
        }
    }

    // ---------------------------------------------------------------- CMANet.forward (models/cma.py:211-333)
    void cma_step(const void* rgb, int rgb_dt, const float* depth, const void* ids, int ids_dt, int B, const float* h_in,
                  const float* mask, float* out, float* stop, float* h_out) {
        ar.reset();
        const hcm_config& c = ctx->cfg;
        const hcm_cma_config& m = ctx->cma_cfg;
        const CmaW& w = ctx->cma;
        const int L = ctx->cur_L, Lm = c.instr_len, Hi = m.instr_hidden, C = Hi * w.dirs, E = m.embedding_size;
        const int H = c.hidden, hh = H / 2, rC = 2048 + 64, dS = w.depth_S, dC = w.depth_C;
        const int R = c.rnn_type == HCM_LSTM ? 2 : 1;
        const bool multi = ctx->concurrent && !ctx->taps_on;
        hipStream_t main_s = ctx->stream;
        hipStream_t a0 = multi ? ctx->aux[0] : main_s, a1 = multi ? ctx->aux[1] : main_s;
        // buffers shared across the forked chains
        use(ctx->dt_vla);
        void* rgb_tok = alloc_t((size_t)B * 16 * rC);
        void* dep_tok = alloc_t((size_t)B * dS * dC);
        float* ins = alloc_f((size_t)B * Lm * C);                 // (B, C, L) of the reference, token-major [B][L][C]
        if (multi) fork_join_begin(2);

        // chain A (aux 0): instruction encoder (instruction_encoder.py:70-92) -- embedding, input projection of every token
        // at once, then L packed-LSTM steps per direction.  All L steps run (no host sync on the longest length): steps
        // past a sample's length emit zeros, which the attention masks exactly as pad_packed_sequence's cut does.
        on(a0);
        if (m.ablate_instruction) {
            // cma.py:236-237 `instruction_embedding * 0`: the encoder's value cannot reach an output (finite x * 0 = 0) -- not run.  Downstream the
            // reference's text_mask = (embedding == 0).all(dim=1) is then true at EVERY position: all logits equal (q . bias - 1e8), uniform weights
            // over zero values, text = 0 exactly -- which is what attn1q over a zero instruction returns for any lengths.
            if (!dry) ck(hipMemsetAsync(ins, 0, (size_t)B * Lm * C * 4, s), "ablate: zero instruction");
            tap("cma.instruction", ins, false, {B, L, C});
        } else {
            const int G = m.instr_rnn == HCM_GRU ? 3 : 4;
            const int ldx = w.ih[0].Kp;
            float* x = alloc_f((size_t)B * Lm * ldx);
            if (!dry) ck(launch_instr_embed(ids, ids_dt, w.emb, x, ctx->len_buf, B, L, E, ldx, m.vocab_size, s), "instr embed");
            float* pre[2] = {alloc_f((size_t)B * Lm * G * Hi), w.dirs > 1 ? alloc_f((size_t)B * Lm * G * Hi) : nullptr};
            float* gh = alloc_f((size_t)B * G * Hi);
            float* hc = alloc_f((size_t)2 * B * Hi);
            for (int d = 0; d < w.dirs; ++d) linear(w.ih[d], x, B * L, ldx, pre[d], G * Hi, ACT_NONE, true);
            static const bool no_scan = dev_env("HCM_NO_LSTM_SCAN") != nullptr;
            if (m.instr_rnn == HCM_GRU) {
                // INSTRUCTION_ENCODER.rnn_type = "GRU" (instruction_encoder.py:42): a launch pair per token and direction
                for (int d = 0; d < w.dirs; ++d) {
                    if (!dry) ck(hipMemsetAsync(hc, 0, (size_t)B * Hi * 4, s), "gru state reset");
                    for (int k = 0; k < L; ++k) {
                        const int t = d == 0 ? k : L - 1 - k;
                        linear(w.hh[d], hc, B, Hi, gh, 3 * Hi, ACT_NONE, true);
                        if (!dry) ck(launch_instr_gru_cell(pre[d], gh, hc, ctx->len_buf, ins, t, B, L, Hi, C, d * Hi, s), "instr gru cell");
                    }
                }
            } else if (Hi == 256 && !no_scan) {
                // both directions, all L steps: one launch
                if (!dry) ck(launch_instr_lstm_scan(pre[0], pre[1], w.hh_t[0], w.hh_t[1], ctx->len_buf, ins, B, L, Hi, w.dirs, C, s), "instr lstm scan");
            } else {
                for (int d = 0; d < w.dirs; ++d) {
                    if (!dry) ck(hipMemsetAsync(hc, 0, (size_t)2 * B * Hi * 4, s), "lstm state reset");
                    for (int k = 0; k < L; ++k) {
                        const int t = d == 0 ? k : L - 1 - k;
                        linear(w.hh[d], hc, B, Hi, gh, 4 * Hi, ACT_NONE, true);
                        if (!dry) ck(launch_instr_lstm_cell(pre[d], gh, hc, hc + (size_t)B * Hi, ctx->len_buf, ins, t, B, L, Hi, C, d * Hi, s), "instr lstm cell");
                    }
                }
            }
            tap("cma.instruction", ins, false, {B, L, C});
        }
        // chain B (aux 1): depth encoder -> spatial tokens (cma.py:220-221)
        on(a1);
        if (m.ablate_depth) {
            // cma.py:238-239 `depth_embedding * 0`: the whole (B, C + 64, S) token tensor including its positional-embedding channels
            use(ctx->dt_vla);
            if (!dry) ck(hipMemsetAsync(dep_tok, 0, (size_t)B * dS * dC * esz, s), "ablate: zero depth tokens");
            tap("cma.depth_spatial", dep_tok, true, {B, dS, dC});
        } else {
            use(ctx->dt_depth);
            Act o = depth_trunk(w.depth, depth, B, "cma.depth");
            void* tok = ctx->dt_depth == ctx->dt_vla ? dep_tok : alloc_t((size_t)B * dS * dC);
            if (!dry) {
                ck(launch_adaptive_avgpool(o.p, tok, dt, B, o.H, o.W, o.C, o.H, o.W, dC, s), "depth tokens");
                ck(launch_fill_cols(w.depth_pe, (char*)tok + (size_t)o.C * esz, dt, B, dS, 64, dC, s), "depth pe");
                if (tok != dep_tok) ck(launch_convert(tok, ctx->dt_depth, dep_tok, ctx->dt_vla, (size_t)B * dS * dC, s), "depth tokens convert");
            }
            use(ctx->dt_vla);
            tap("cma.depth_spatial", dep_tok, true, {B, dS, dC});
        }
        // chain C (caller's stream): RGB encoder -> spatial tokens (cma.py:223-224)
        on(main_s);
        if (m.ablate_rgb) {
            use(ctx->dt_vla);                                     // cma.py:240-241
            if (!dry) ck(hipMemsetAsync(rgb_tok, 0, (size_t)B * 16 * rC * esz, s), "ablate: zero rgb tokens");
            tap("cma.rgb_spatial", rgb_tok, true, {B, 16, rC});
        } else {
            use(ctx->dt_rgb);
            Act o = rgb_trunk(w.rgb, rgb, rgb_dt, B, "cma.rgb");
            void* tok = ctx->dt_rgb == ctx->dt_vla ? rgb_tok : alloc_t((size_t)B * 16 * rC);
            if (!dry) {
                ck(launch_adaptive_avgpool(o.p, tok, dt, B, o.H, o.W, o.C, 4, 4, rC, s), "rgb tokens");
                ck(launch_fill_cols(w.rgb_pe, (char*)tok + (size_t)2048 * esz, dt, B, 16, 64, rC, s), "rgb pe");
                if (tok != rgb_tok) ck(launch_convert(tok, ctx->dt_rgb, rgb_tok, ctx->dt_vla, (size_t)B * 16 * rC, s), "rgb tokens convert");
            }
            use(ctx->dt_vla);
            tap("cma.rgb_spatial", rgb_tok, true, {B, 16, rC});
        }
        if (multi) fork_join_end(2);

        // first state encoder over [rgb_in | depth_in] (cma.py:256-270)
        const int ld1 = w.rnn1.in + H;
        float* xh1 = alloc_f((size_t)B * ld1);
        void* rmean = alloc_t((size_t)B * rC);
        if (!dry) ck(launch_mean_rows(rgb_tok, rmean, dt, B, 16, rC, rC, rC, 0, s), "rgb mean");
        linear(w.rgb_linear, rmean, B, rC, xh1, ld1, ACT_RELU, true);
        linear(w.depth_linear, dep_tok, B, dS * dC, xh1 + c.rgb_out, ld1, ACT_RELU, true);
        rnn_step(w.rnn1, xh1, ld1, B, h_in, mask, h_out, Heads{});
        const float* state = h_out;                               // new h of the first encoder: (B, H)
        tap("cma.state", state, false, {B, H});
        // x = [state | text | rgb | depth] (cma.py:309-311)
        const int ldc = H + C + c.rgb_out + c.depth_out;
        float* xc = alloc_f((size_t)B * ldc);
        if (!dry) ck(hipMemcpy2DAsync(xc, (size_t)ldc * 4, state, (size_t)H * 4, (size_t)H * 4, B, hipMemcpyDeviceToDevice, s), "state copy");
        // text attention (:272-277): the query is the state, keys text_k(instruction), values the instruction itself
        float* q1 = alloc_f((size_t)B * hh);
        linear(w.state_q, state, B, H, q1, hh, ACT_NONE, true);
        float* kt = alloc_f((size_t)B * Lm * hh);
        linear(w.text_k, ins, B * L, C, kt, hh, ACT_NONE, true);
        if (!dry) ck(launch_attn1q(q1, hh, kt, hh, ins, C, m.ablate_instruction ? nullptr : ctx->len_buf, xc + H, ldc, B, L, hh, C, w.scale, s), "text attention");
        // visual attention (:281-290): query text_q(text); keys | values are the two halves of rgb_kv / depth_kv
        float* q2 = alloc_f((size_t)B * hh);
        linear(w.text_q, xc + H, B, ldc, q2, hh, ACT_NONE, true);
        float* rkv = alloc_f((size_t)B * 16 * w.rgb_kv.N);
        linear(w.rgb_kv, rgb_tok, B * 16, rC, rkv, w.rgb_kv.N, ACT_NONE, true);
        float* dkv = alloc_f((size_t)B * dS * w.depth_kv.N);
        linear(w.depth_kv, dep_tok, B * dS, dC, dkv, w.depth_kv.N, ACT_NONE, true);
        if (!dry) {
            ck(launch_attn1q(q2, hh, rkv, w.rgb_kv.N, rkv + hh, w.rgb_kv.N, nullptr, xc + H + C, ldc, B, 16, hh, c.rgb_out, w.scale, s), "rgb attention");
            ck(launch_attn1q(q2, hh, dkv, w.depth_kv.N, dkv + hh, w.depth_kv.N, nullptr, xc + H + C + c.rgb_out, ldc, B, dS, hh, c.depth_out, w.scale, s), "depth attention");
        }
        // second_state_compress + second state encoder + heads (:312-332)
        const int ld2 = w.rnn2.in + H;
        float* xh2 = alloc_f((size_t)B * ld2);
        linear(w.compress, xc, B, ldc, xh2, ld2, ACT_RELU, true);
        tap("cma.compress", xh2, false, {B, ld2});
        Heads hd;
        hd.w0 = w.lin_w; hd.b0 = w.lin_b; hd.out0 = out; hd.r0 = c.num_actions; hd.ld0 = c.num_actions;
        hd.w1 = w.stop_w; hd.b1 = w.stop_b; hd.out1 = stop; hd.r1 = 1; hd.ld1 = 1;
        rnn_step(w.rnn2, xh2, ld2, B, h_in + (size_t)R * B * H, mask, h_out + (size_t)R * B * H, hd);
    }


    // ---------------------------------------------------------------- one step: independent encoder chains run on
    // separate HIP streams (fork/join by events, capturable into a hipGraph): the two RGB ResNet-50s, the two depth
    // trunks and BERT have no data dependence until the cross-modal block / the recurrent cells.  Each chain owns a
    // disjoint arena region; nothing is released until the step is fully enqueued.
    void fork_join_begin(int n_aux) {
        if (dry || n_aux == 0) return;
        if (ctx->seg_mode && !ctx->seg_open) { hcm_ctx::SegOp op; op.kind = 0; op.n = n_aux; ctx->seg_prog.push_back(op); return; }
        ck(hipEventRecord(ctx->ev_fork, ctx->stream), "fork record");
        for (int i = 0; i < n_aux; ++i) ck(hipStreamWaitEvent(ctx->aux[i], ctx->ev_fork, 0), "fork wait");
    }
    void fork_join_end(int n_aux) {
        if (dry || n_aux == 0) return;
        if (ctx->seg_mode && !ctx->seg_open) { hcm_ctx::SegOp op; op.kind = 2; op.n = n_aux; ctx->seg_prog.push_back(op); return; }
        for (int i = 0; i < n_aux; ++i) {
            ck(hipEventRecord(ctx->ev_join[i], ctx->aux[i]), "join record");
            ck(hipStreamWaitEvent(ctx->stream, ctx->ev_join[i], 0), "join wait");
        }
    }
    void on(hipStream_t st) { s = st; }
    // segmented capture (model.h, SegOp): everything a chain enqueues between chain_begin() and chain_end() becomes one graph captured on the
    // chain's own stream; outside a segmented capture both are no-ops
    void chain_begin() {
        if (!ctx->seg_mode || dry) return;
        ck(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal), "chain capture begin");
        ctx->seg_open = true;
        ctx->seg_stream = s;
    }
    void chain_end() {
        if (!ctx->seg_mode || dry || !ctx->seg_open) return;
        hipGraph_t g = nullptr;
        ctx->seg_open = false;
        ck(hipStreamEndCapture(ctx->seg_stream, &g), "chain capture end");
        if (!g) return;                                   // (a chain that enqueued nothing, e.g. BERT with the instruction stream cached)
        size_t nodes = 0;
        hipError_t e = hipGraphGetNodes(g, nullptr, &nodes);
        if (e == hipSuccess && nodes > 0) {
            hcm_ctx::SegOp op;
            op.kind = 1;
            op.st = ctx->seg_stream;
            e = hipGraphInstantiate(&op.exec, g, nullptr, nullptr, 0);
            if (e == hipSuccess) ctx->seg_prog.push_back(op);
        }
        (void)hipGraphDestroy(g);
        ck(e, "chain graph instantiate");
    }
    // development aid: stamp the wall clock at this point of the current chain (no-op unless hcm_create saw HCM_MARKS=1 in a DEV build)
    void mark(const std::string& name) {
        if (!ctx->marks_dev || dry) return;
        size_t i = 0;
        for (; i < ctx->mark_names.size(); ++i) if (ctx->mark_names[i] == name) break;
        if (i == ctx->mark_names.size()) { if (i >= 256) return; ctx->mark_names.push_back(name); }
        ck(launch_mark(ctx->marks_dev + i, s), "mark");
    }

    void step(bool do_hi, bool do_lo, const void* rgb, int rgb_dt, const float* depth, const void* ids, int ids_dt, int B,
              const float* hi_h_in, const float* lo_h_in, const float* mask, const int64_t* subtask,
              float* logits, int ld_logits, float* vel, int ld_vel, float* stop, int ld_stop, float* hi_h_out, float* lo_h_out) {
        ar.reset();
        HiBufs hb;
        LoBufs lb;
        if (do_hi) hb = hi_alloc(B);
        if (do_lo) lb = lo_alloc(B);
        const bool multi = ctx->concurrent && !ctx->taps_on;    // taps allocate/synchronise: keep them single-stream
        hipStream_t main_s = ctx->stream;
        hipStream_t a0 = multi ? ctx->aux[0] : main_s, a1 = multi ? ctx->aux[1] : main_s, a2 = multi ? ctx->aux[2] : main_s;
#ifdef HCM_DEV_KNOBS
        static const int skip = dev_env("HCM_SKIP") ? atoi(dev_env("HCM_SKIP")) : 0;   // profiling aid (make DEV=1 builds only): drop chains (bitmask)
#else
        constexpr int skip = 0;
#endif
        if (multi) fork_join_begin(4);
        if (ctx->host_frames && !dry) {
            // HCM_ACT_HOST_FRAMES: each chain's frames come up from the pinned host buffers on that chain's own stream, so BERT and the other
            // chain's compute run beside the copies.  The RGB frames go FIRST: the copy engine works in submission order and the RGB trunks are
            // the long chain (the depth chain's kernels fill gaps, they can start 0.3 ms later)
            const size_t n_rgb = (size_t)B * ctx->cfg.rgb_h * ctx->cfg.rgb_w * 3 * (rgb_dt == DT_U8 ? 1 : 4), n_dep = (size_t)B * ctx->cfg.depth_h * ctx->cfg.depth_w * 4;
            if (ctx->seg_mode) {
                // per-chain linear graphs: the copies are enqueued by the replay itself, OUTSIDE the graphs, at the head of their chains' streams --
                // both are in the copy engines' queues a few microseconds after the call, with BERT running beside them
                hcm_ctx::SegOp c0, c1;
                c0.kind = c1.kind = 3;
                c0.st = main_s; c0.dst = ctx->stage_rgb; c0.src = rgb; c0.bytes = n_rgb;
                c1.st = a1; c1.dst = ctx->stage_depth; c1.src = depth; c1.bytes = n_dep;
                ctx->seg_prog.push_back(c0);
                ctx->seg_prog.push_back(c1);
            } else {
            ck(hipMemcpyAsync(ctx->stage_rgb, rgb, n_rgb, hipMemcpyHostToDevice, main_s), "rgb frames H2D");
            ck(hipMemcpyAsync(ctx->stage_depth, depth, n_dep, hipMemcpyHostToDevice, a1), "depth frames H2D");
            }
            rgb = ctx->stage_rgb;
            depth = ctx->stage_depth;
        }
        // Host enqueue order = start order on the GPU: the chains made of many small dependent launches go first (BERT,
        // then the depth trunks) so they are not delayed by the ~2.5 us/launch it takes to enqueue the bulk RGB chains.
        // chain 3: BERT
        on(a2);
        chain_begin();
        mark("bert.start");
        // (the workspace layout is identical from step to step, so hb.I / hb.Q of the previous step are still in place when the
        //  caller declares the instructions unchanged)
        if (do_hi && !(skip & 8) && !(ctx->reuse_instruction && !dry)) hi_bert(ids, ids_dt, B, hb);
        chain_end();
        // chains 2 and 4: the two depth trunks (small, latency-bound kernels that fill the gaps of the RGB chains)
        on(a1);
        chain_begin();
        mark("depth.start");
        const bool dshare = do_hi && do_lo && ctx->hi.depth_shared && !ctx->lo.depth_simple;
        const bool pair = do_hi && do_lo && ctx->hi.has_depth_pair && !ctx->lo.depth_simple;
        if (ctx->cfg.ablate_depth) {
            if (!(skip & 4)) ablated_encoder(1, B, do_hi ? &hb : nullptr, do_lo ? &lb : nullptr);
        } else if (dshare) {
            if (!(skip & 4)) depth_shared(depth, B, hb, lb);
        } else if (pair) {
            if (!(skip & 4)) depth_pair(depth, B, hb, lb);
        } else if (do_hi && !(skip & 4)) hi_depth(depth, B, hb);
        on(a1);
        if (do_lo && !pair && !dshare && !ctx->cfg.ablate_depth && !(skip & 4)) lo_depth(depth, B, lb);
        mark("depth.end");
        chain_end();
        // chain 0 (caller's stream): the high-level RGB trunk (or the low-level one when it is the only model)
        on(main_s);
        chain_begin();
        mark("rgb.start");
        const bool rshare = do_hi && do_lo && ctx->hi.rgb_shared && !ctx->lo.rgb_simple;
        const bool rpair = do_hi && do_lo && ctx->hi.has_rgb_pair && !ctx->lo.rgb_simple;
        if (ctx->cfg.ablate_rgb) { if (!(skip & 1)) ablated_encoder(0, B, do_hi ? &hb : nullptr, do_lo ? &lb : nullptr); }
        else if (rshare) { if (!(skip & 1)) rgb_shared(rgb, rgb_dt, B, hb, lb); }
        else if (rpair) { if (!(skip & 1)) rgb_pair(rgb, rgb_dt, B, hb, lb); }
        else if (!(skip & 1)) { if (do_hi) hi_rgb(rgb, rgb_dt, B, hb); else lo_rgb(rgb, rgb_dt, B, lb); }
        // chain 1: the low-level RGB trunk
        static const int rgb_serial = dev_env("HCM_RGB_SERIAL") ? atoi(dev_env("HCM_RGB_SERIAL")) : 1;
        on((rgb_serial || ctx->host_frames) ? main_s : a0);        // (staged frames: behind their copy)
        if (do_hi && do_lo && !rpair && !rshare && !ctx->cfg.ablate_rgb && !(skip & 2)) lo_rgb(rgb, rgb_dt, B, lb);
        on(main_s);
        mark("rgb.end");
        chain_end();
        if (multi) fork_join_end(4);
        chain_begin();
        mark("tail.start");
        if (do_hi && do_lo && T == 1) { lo_early = &lb; lo_h_in_early = lo_h_in; }
        if (do_hi) hi_tail(B, hb, hi_h_in, mask, logits, ld_logits, hi_h_out);
        const int64_t* st_ids = subtask;
        if (do_hi && do_lo) {
            // pred = argmax(output, dim=1)  (hierarchical_trainer.py:1098)
            if (!dry && !pred_fused) ck(launch_argmax(logits, ctx->pred_buf, B, ctx->cfg.num_actions, ld_logits, s), "argmax");
            st_ids = ctx->pred_buf;
        }
        if (do_lo) lo_tail(B, lb, lo_h_in, mask, st_ids, vel, ld_vel, stop, ld_stop, lo_h_out);
        mark("tail.end");
        chain_end();
    }
};

// entry point used by api.cpp
void run_step(hcm_ctx* ctx, bool do_hi, bool do_lo, const void* rgb, int rgb_dt, const float* depth, const void* ids, int ids_dt,
              int B, const float* hi_h_in, const float* lo_h_in, const float* mask, const int64_t* subtask, float* logits,
              int ld_logits, float* vel, int ld_vel, float* stop, int ld_stop, float* hi_h_out, float* lo_h_out, int T) {
    Fwd f(ctx);
    f.T = T;
    f.step(do_hi, do_lo, rgb, rgb_dt, depth, ids, ids_dt, B, hi_h_in, lo_h_in, mask, subtask, logits, ld_logits, vel, ld_vel,
           stop, ld_stop, hi_h_out, lo_h_out);
}

}  // namespace hcm

namespace hcm {
void run_refresh_instruction(hcm_ctx* ctx, const void* ids, int ids_dt, int B, const int32_t* idx, int n) {
    Fwd f(ctx);
    f.refresh_instruction(ids, ids_dt, B, idx, n);
}
void run_cma(hcm_ctx* ctx, const void* rgb, int rgb_dt, const float* depth, const void* ids, int ids_dt, int B, const float* h_in,
             const float* mask, float* out, float* stop, float* h_out) {
    Fwd f(ctx);
    f.cma_step(rgb, rgb_dt, depth, ids, ids_dt, B, h_in, mask, out, stop, h_out);
}
}  // namespace hcm
