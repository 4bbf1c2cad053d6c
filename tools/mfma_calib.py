"""Matrix-pipe calibration (tools/native/mfma_calib.hip): a kernel of nothing but v_mfma_f32_16x16x32_f16.
Plain run: sustained issue rate (cycles per MFMA per SIMD at the clock the chip holds) on random and on zero operands.
Under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`: the counters against the exact count
(tools/pmc_calib_summary.py reads the CSV).  usage: mfma_calib.py [iters]"""
import ctypes as C, os, sys, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "libmfma_calib.so"))
lib.run_mfma_only.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4000
sink = torch.zeros(4, device="cuda")
for fill in ("random", "zeros"):
    seed = (torch.randn(128 * 8, device="cuda") if fill == "random" else torch.zeros(128 * 8, device="cuda")).half().contiguous()
    for waves in (4, 8):
        run = lambda: lib.run_mfma_only(waves, seed.data_ptr(), sink.data_ptr(), iters, 256, None)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        n_mfma_simd = iters * 32 * (waves // 4)            # MFMAs per SIMD and launch
        flops = 256 * 4 * n_mfma_simd * 16 * 16 * 32 * 2
        print(f"{fill:6s} operands, {waves} waves/CU: {us:8.1f} us per launch, {flops / us / 1e6:7.0f} TFLOP/s, "
              f"{us * 1e-6 * 2.4e9 / n_mfma_simd:5.2f} cycles@2.4GHz per MFMA per SIMD; MFMAs per SIMD {n_mfma_simd}, chip-wide {256 * 4 * n_mfma_simd}")

# round 6: gemm256f_kernel's K loop (fragment reads + MFMAs, no DMA) on either matrix instruction -- same bytes and FLOPs per step
if "loop" in sys.argv[1:]:          # usage: mfma_calib.py loop
    lib.run_mfma_loop.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    it2 = 2000
    res = {}
    for rep in range(2):
        for shape in (16, 32):
            run = lambda: lib.run_mfma_loop(shape, sink.data_ptr(), it2, 256, None)
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
            flops = 256 * 8 * it2 * 2 * 32 * 16 * 16 * 32 * 2           # 8 waves x 2 K steps x 32 MFMA-equivalents of 16x16x32
            res.setdefault(shape, []).append(us)
            print(f"K loop of a 128 x 64 wave tile, v_mfma_f32_{'16x16x32' if shape == 16 else '32x32x16'}_f16, 8 waves per CU: {us:8.1f} us, {flops / us / 1e6:7.0f} TFLOP/s "
                  f"({us * 1e-6 * 2.1e9 / (it2 * 2 * 2):6.1f} cycles@2.1GHz per K step and SIMD)")
    print(f"32x32x16 / 16x16x32 time ratio: {min(res[32]) / min(res[16]):.3f}")
