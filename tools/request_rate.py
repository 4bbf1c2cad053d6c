"""Request-rate table (tools/native/request_rate.hip): 1 KB vector-memory requests per microsecond and CU by request shape, issuing waves
per SIMD, bytes in flight and cache level -- writes the markdown that is kept as profiles/r6_request_rate.md.
usage: python tools/request_rate.py > gpurun_out/r6_request_rate.md"""
import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "librequest_rate.so"))
lib.run_rr.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lib.run_rr_loop.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
big = torch.randint(0, 255, (1 << 30,), dtype=torch.uint8, device="cuda")
SHAPES = ((0, "1 KB contiguous"), (1, "8 rows x 128 B"), (2, "16 rows x 64 B"), (3, "32 rows x 32 B"))
LD = 1536          # row stride of the strided shapes: a K = 768 16-bit operand (BERT)


def clock_ghz():
    return 2.1     # nominal under load; the table quotes requests/us, cycles are indicative only


def timed(run, reps=5):
    for _ in range(2): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


print("# Request rate of a CU's vector-memory path by request shape (tools/request_rate.py, one 1 KB request = one wave instruction of 64 x 16 B)\n")
print("One workgroup per CU (256 workgroups), each re-walking a PRIVATE region: `L2` = 96 KB per workgroup (3 MB per XCD: misses the CU's 32 KB L1, sits in the XCD's 4 MB L2),")
print("`MALL` = 512 KB per workgroup (16 MB per XCD, 128 MB in all: misses L2, sits in the 256 MB Infinity Cache), `HBM` = 4 MB per workgroup (1 GB in all).")
print("`waves/SIMD` = issuing waves per SIMD (4 or 8 waves per workgroup); `window` = requests a wave keeps in flight.  GB/s per CU = requests/us x 1.024.\n")
for kind, kname in ((0, "LDS-DMA `buffer_load_dwordx4 ... lds`"), (1, "`buffer_load_dwordx4` into registers")):
    print(f"## {kname}\n")
    print("| level | waves/SIMD | window | " + " | ".join(f"{n}: req/us/CU (GB/s)" for _, n in SHAPES) + " | contiguous / 8x128 | contiguous / 16x64 |")
    print("|---|---|---|" + "---|" * (len(SHAPES) + 2))
    for level, nbytes in (("L2", 256 * 96 * 1024), ("MALL", 128 << 20), ("HBM", 1 << 30)):
        for nw in (4, 8):
            for win in (4, 8, 16):
                rates = []
                for shape, _ in SHAPES:
                    iters = 600 if level == "L2" else 200
                    us = timed(lambda: lib.run_rr(kind, shape, nw, win, big.data_ptr(), nbytes, LD, iters, 256, sink.data_ptr(), None))
                    rates.append(iters * win * nw / us)
                print(f"| {level} | {nw // 4} | {win} | " + " | ".join(f"{r:.1f} ({r * 1.024:.0f})" for r in rates) + f" | {rates[0] / rates[1]:.2f} | {rates[0] / rates[2]:.2f} |")
    print()
print("## LDS-DMA requests issued from inside a loop that also feeds the matrix pipe and reads LDS (8 waves, 8 requests per wave and iteration, <= 12 in flight)\n")
print("Per request the wave also issues `mfma` 16x16x32 MFMAs and `reads` `ds_read_b128` of an LDS image (no data dependence between them) -- the mix of a GEMM K loop:")
print("gemm256f_kernel has 8 MFMAs + 3 fragment reads per request, the 128 x 128 tiles 4 + 2.\n")
print("| level | mfma/request | reads/request | " + " | ".join(f"{n}: req/us/CU" for _, n in SHAPES) + " | cycles/request at 2.1 GHz (contiguous, 8x128) | contiguous / 8x128 |")
print("|---|---|---|" + "---|" * (len(SHAPES) + 2))
for level, nbytes in (("L2", 256 * 96 * 1024), ("MALL", 128 << 20), ("HBM", 1 << 30)):
    for nm, nr in ((0, 0), (0, 3), (8, 0), (4, 2), (8, 3)):
        rates = []
        for shape, _ in SHAPES:
            iters = 300
            us = timed(lambda: lib.run_rr_loop(nm, nr, shape, 8, big.data_ptr(), nbytes, LD, iters, 256, sink.data_ptr(), None))
            rates.append(iters * 64 / us)
        cyc = [2100.0 / r for r in rates[:2]]
        print(f"| {level} | {nm} | {nr} | " + " | ".join(f"{r:.1f}" for r in rates) + f" | {cyc[0]:.0f}, {cyc[1]:.0f} | {rates[0] / rates[1]:.2f} |")
print("\nFor scale: 8 MFMAs of 16x16x32 per request on 8 waves = 2 waves per SIMD x 8 x ~16.8 cycles = 269 matrix-pipe cycles per SIMD per request pair, i.e. a loop that is")
print("matrix-bound retires 64 requests per iteration in 8 x 269 = 2150 cycles = ~62 requests/us/CU at 2.1 GHz; rates above that are not visible in this row.")
