"""HBM rate of the expansion conv's memory pattern alone (tools/native/tile_stream.hip; build: hipcc --offload-arch=gfx950 -O3
-shared -fPIC tools/native/tile_stream.hip -o tools/native/libtile_stream.so)."""
import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "libtile_stream.so"))
lib.run_tile_stream.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p]
M, Cin, N = 128 * 64 * 64, 64, 256
x = torch.randn(M, Cin, device="cuda").bfloat16(); r = torch.randn(M, N, device="cuda").bfloat16(); y = torch.empty_like(r)
for rows, cols in ((128, 128), (64, 256), (128, 256), (32, 256), (256, 128), (16, 256)):
    for use_res in (0, 1):
        run = lambda: lib.run_tile_stream(x.data_ptr(), r.data_ptr(), y.data_ptr(), M, Cin, N, use_res, rows, cols, None)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        mb = (x.numel() * (N // cols) * 0 + x.numel() + y.numel() * (1 + use_res)) * 2 / 1e6
        print(f"tile {rows}x{cols} residual={use_res}: {us:.1f} us  {mb:.0f} MB -> {mb / us / 1e6 * 1e6 / 1e6:.2f} TB/s")

# the same bytes with the number of resident workgroups per CU pinned by dynamic LDS: how much memory-level parallelism the rate needs
lib.run_tile_stream_lds.argtypes = [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p]
for lds, what in ((0, "unlimited (4 per CU)"), (50 * 1024, "3 per CU"), (73 * 1024, "2 per CU"), (120 * 1024, "1 per CU")):
    run = lambda: lib.run_tile_stream_lds(x.data_ptr(), r.data_ptr(), y.data_ptr(), M, Cin, N, 1, 128, 256, lds, None)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = (x.numel() + 2 * y.numel()) * 2 / 1e6
    print(f"tile 128x256 residual=1, workgroups {what}: {us:.1f} us  {mb / us:.2f} TB/s" .replace("TB/s", "MB/us = TB/s"))
