"""Few-row GEMM (csrc/skinny.hip, impl 3) against the implicit-GEMM tiles (impl 1) per shape: time of one launch inside a chain of dependent
launches on one stream (events around 200 launches), so the figure holds the launch boundary as a step does.
usage: python tools/skinny_bench.py [md]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
SHAPES = [  # (dtype code, M, N, K, act, res, out_f32, what)
    (5, 80, 2304, 768, 0, 0, 0, "BERT QKV, one environment (L = 80)"),
    (5, 80, 768, 768, 0, 1, 0, "BERT attention output"),
    (5, 80, 3072, 768, 2, 0, 0, "BERT FFN1 + GELU"),
    (5, 80, 768, 3072, 0, 1, 0, "BERT FFN2"),
    (5, 80, 256, 768, 0, 0, 0, "ins_fc"),
    (5, 20, 2304, 768, 0, 0, 0, "BERT QKV, L = 20"),
    (5, 80, 2304, 768, 0, 0, 0, "configs[0]: 4 x 20 rows"),
    (5, 160, 2304, 768, 0, 0, 0, "two environments"),
    (5, 160, 768, 3072, 0, 1, 0, "two environments, FFN2"),
    (5, 320, 2304, 768, 0, 0, 0, "four environments"),
    (5, 320, 3072, 768, 2, 0, 0, "four environments, FFN1"),
    (5, 320, 768, 3072, 0, 1, 0, "four environments, FFN2"),
    (5, 640, 2304, 768, 0, 0, 0, "eight environments (beyond the default rule)"),
    (5, 64, 256, 2112, 0, 0, 0, "rgb_linear at B = 64"),
    (5, 64, 768, 768, 0, 1, 0, "M = 64 projection"),
    (0, 64, 2048, 896, 0, 0, 1, "LSTM gates, early half, B = 64 (f32)"),
    (0, 64, 2048, 512, 0, 1, 1, "LSTM gates, late half, B = 64 (f32)"),
    (0, 64, 2048, 32, 0, 1, 1, "low-level LSTM late half (sub-task embedding), B = 64 (f32)"),
    (0, 1, 2048, 896, 0, 0, 1, "LSTM gates, one environment (f32)"),
    (0, 256, 2048, 896, 0, 0, 1, "LSTM gates, B = 256 (f32)"),
]
def run(dt, M, N, K, act, res, of32, impl, n=200):
    tdt = torch.float32 if dt == 0 else torch.float16
    x = torch.randn(M, K, device="cuda").to(tdt); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(tdt)
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda").to(tdt) if res else None
    y = torch.empty(M, N, device="cuda", dtype=torch.float32 if of32 else tdt)
    f = lambda: lib.hcm_op_linear_impl(p(x), p(w), p(b), p(r), p(y), dt, M, N, K, act, of32, impl, None)
    if f() != 0: return float("nan")
    for _ in range(20): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
print("| shape | what | implicit-GEMM tiles us | few-row kernel us | ratio |\n|---|---|---|---|---|")
for dt, M, N, K, act, res, of32, what in SHAPES:
    a, b = run(dt, M, N, K, act, res, of32, 1), run(dt, M, N, K, act, res, of32, 3)
    print(f"| {'f32' if dt == 0 else 'f16'} M={M} N={N} K={K}{' res' if res else ''}{' gelu' if act == 2 else ''} | {what} | {a:.1f} | {b:.1f} | {a / b:.2f} |", flush=True)
