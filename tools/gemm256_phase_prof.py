"""Phase stamps (s_memtime) of the profiled builds of the 256 x 256 GEMM kernels (`make DEV=1` library): hcm_op_linear_impl variants
13 (8-phase, stamps at the clean points), 14 (8-phase, extra stamp in front of barrier 1), 15 (free-running form).
Prints per wave group (rows 0-127 / 128-255 of the tile = the two waves of every SIMD) the average cycles per K tile of every slot.
usage: gemm256_phase_prof.py [M N]   (KS=768,3072)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HCM_DEV_LIB', '1')
import torch, hcm_pkg
hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
NAMES = {13: ["load seg + barrier-1 wait", "MFMA issue (16)", "barrier-2 wait", "-", "prologue", "epilogue"],
         14: ["load seg (reads complete)", "MFMA issue (16)", "barrier-2 wait", "barrier-1 wait", "prologue", "epilogue"],
         15: ["first half (32 MFMA + 12 reads)", "lgkmcnt+vmcnt wait", "barrier", "second half (32 MFMA + 8 DMA + 12 reads)", "prologue", "epilogue"]}
buf = (C.c_uint64 * 1024)()
for K in [int(k) for k in os.environ.get('KS', '768,3072').split(',')]:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).half(); w = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.06).half(); b = torch.rand(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for var in (13, 14, 15):
        impl = 2 + 16 * var
        def run():
            rc = lib.hcm_op_linear_impl(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), 5, M, N, K, 0, 0, impl, st)
            assert rc == 0, rc
        for _ in range(5): run()
        torch.cuda.synchronize()
        assert lib.hcm_debug_gemm256_prof(buf, 1) == 0
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        assert lib.hcm_debug_gemm256_prof(buf, 1) == 0
        v = torch.tensor(list(buf), dtype=torch.float64).view(16, 8, 8)
        print(f"\n## variant {var}, M={M} N={N} K={K}: {us:.1f} us per launch under stamps ({K // 64} K tiles)")
        for grp, sl in (("rows 0-127 (waves 0-3)", slice(0, 4)), ("rows 128-255 (waves 4-7)", slice(4, 8))):
            g = v[:, sl, :].sum(dim=(0, 1))
            tiles, launches = g[6].item(), g[7].item()
            if tiles == 0: continue
            parts = []
            for i in range(4):
                if NAMES[var][i] != "-": parts.append(f"{NAMES[var][i]} {g[i].item() / tiles:7.0f}")
            loop = sum(g[i].item() for i in range(4)) / tiles
            print(f"  {grp}: per K tile: " + " | ".join(parts) + f" | sum {loop:7.0f} cycles;  per launch: prologue {g[4].item() / launches:7.0f}, epilogue {g[5].item() / launches:7.0f}")
