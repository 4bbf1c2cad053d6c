"""Per-CU L2 read rate by access flavour (tools/native/l2_rate.hip): LDS-DMA vs plain global loads vs buffer loads into VGPRs.
build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/native/l2_rate.hip -o tools/native/libl2_rate.so"""
import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "libl2_rate.so"))
lib.run_rate.argtypes = [C.c_int, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
for mb in (1, 2, 16):
    buf = torch.randint(0, 255, (mb << 20,), dtype=torch.uint8, device="cuda")
    for blocks in (256, 512):
        for mode, name in ((0, "LDS-DMA dwordx4"), (1, "global_load_dwordx4 -> VGPR"), (2, "buffer_load_dwordx4 -> VGPR")):
            iters = 400
            run = lambda: lib.run_rate(mode, buf.data_ptr(), buf.numel(), iters, blocks, sink.data_ptr(), None)
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
            per_wg = iters * 65536
            waves = (blocks + 255) // 256
            print(f"buffer {mb:2d} MB, {blocks} workgroups, {name:30s}: {us:8.1f} us, {per_wg * blocks / us / 1e6:7.2f} TB/s aggregate, {per_wg * waves / us / 1e3:6.1f} GB/s per CU")

# GEMM-operand pattern (256-row panels, 128 B per row per step, LDS-DMA) by bytes in flight: a wave issues 4 x 1 KB, then waits until at most
# `window` of its loads are outstanding -> up to (window + 4) KB per wave, 8 waves per CU.  L2-resident matrix (2 MB) and an L2-missing one (64 MB).
lib.run_panel.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
buf = torch.randint(0, 255, (64 << 20,), dtype=torch.uint8, device="cuda")
for total in (2 << 20, 64 << 20):
    for ld in (1536, 6144):
        for window in (0, 4, 8, 12, 20, 28):
            rows = total // ld // 256 * 256
            iters = 800
            run = lambda: lib.run_panel(buf.data_ptr(), rows * ld, ld, ld, iters, 256, window, sink.data_ptr(), None)
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
            rate = iters * 32768 / us / 1e3
            print(f"panel fetch {total >> 20:2d} MB matrix, row stride {ld:5d} B, <= {(window + 4) * 8:3d} KB in flight per CU: {rate:6.1f} GB/s per CU  (implied latency {(window + 4) * 8 * 1.024 / rate * 1e3:5.0f} ns)")
