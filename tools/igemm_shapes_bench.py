"""Per-shape micro-benchmark of the implicit-GEMM kernel over every conv / linear shape of one HCM step at batch B
(the layers of both ResNet-50 trunks, BERT-base, the cross-modal block).  Prints a markdown table sorted by time.
usage: python tools/igemm_shapes_bench.py [B] [dtype bf16|fp16]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg
hcm_pkg.load()
from robo_vln_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
code, tdt = {"bf16": (_lib.HCM_BF16, torch.bfloat16), "fp16": (_lib.HCM_F16, torch.float16)}[prec]
lib = _lib.lib()
shapes = {}   # (kind, H, W, Cin, Cout, K, stride, pad) -> count per step


def add(key, n=1):
    shapes[key] = shapes.get(key, 0) + n


def resnet(base, hw, cin_first, count):
    # conv1 as GEMM over im2col
    ho = hw // 2
    kp = (49 * cin_first + 31) // 32 * 32
    add(("lin", B * ho * ho, base, kp), count)
    h = ho // 2
    inpl = base
    for li, nb in enumerate((3, 4, 6, 3)):
        planes = base << li
        for bi in range(nb):
            stride = 2 if (li > 0 and bi == 0) else 1
            add(("conv", h, h, inpl, planes, 1, 1, 0), count)
            add(("conv", h, h, planes, planes, 3, stride, 1), count)
            h2 = h // stride
            add(("conv", h2, h2, planes, planes * 4, 1, 1, 0), count)
            if bi == 0:
                add(("conv", h, h, inpl, planes * 4, 1, stride, 0), count)
            inpl = planes * 4
            h = h2
    return h, inpl


resnet(64, 256, 3, 2)                     # RGB trunk x2 (hi + lo)
if prec == "fp16":
    shapes.clear()
    h, c = resnet(32, 128, 1, 2)          # depth trunk x2
    add(("conv", h, h, c, 128, 3, 1, 1), 2)
else:
    L = 80
    rows = B * L
    add(("lin", rows, 2304, 768), 12); add(("lin", rows, 768, 768), 12); add(("lin", rows, 3072, 768), 12); add(("lin", rows, 768, 3072), 12)
    add(("lin", rows, 256, 768), 1); add(("lin", rows, 256, 256), 1 + 2); add(("lin", rows, 1024, 256), 2); add(("lin", rows, 256, 1024), 2)
    add(("lin", B * 16, 256, 2112), 1); add(("lin", B * 16, 256, 192), 1); add(("lin", B * 16, 256, 256), 2); add(("lin", B * 16, 512, 256), 2)
    add(("lin", B, 256, 2112), 1); add(("lin", B, 128, 3072), 1); add(("lin", B, 256, 2048), 1); add(("lin", B, 128, 2048), 1)

st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rows_out = []
for key, cnt in shapes.items():
    if key[0] == "conv":
        _, H, W, Cin, Cout, K, stride, pad = key
        Ho = (H + 2 * pad - K) // stride + 1
        x = torch.randn(B, H, W, Cin, device="cuda").to(tdt)
        w = (torch.randn(Cout, K, K, Cin, device="cuda") * 0.05).to(tdt)
        b = torch.randn(Cout, device="cuda")
        y = torch.empty(B, Ho, Ho, Cout, device="cuda", dtype=tdt)
        M, N, KK = B * Ho * Ho, Cout, K * K * Cin
        run = lambda: lib.hcm_op_conv2d(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), code, B, H, W, Cin, Cout, K, K, stride, pad, 1, st)
        name = f"conv{K}x{K}/{stride} {Cin}->{Cout} @{H}"
    else:
        _, M, N, KK = key
        x = torch.randn(M, KK, device="cuda").to(tdt)
        w = (torch.randn(N, KK, device="cuda") * 0.05).to(tdt)
        b = torch.randn(N, device="cuda")
        y = torch.empty(M, N, device="cuda", dtype=tdt)
        run = lambda: lib.hcm_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), code, M, N, KK, 0, 0, st)
        name = f"linear {KK}->{N}"
    for _ in range(3):
        assert run() == 0
    for _ in range(60):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = 2.0 * M * N * KK
    mb = (x.numel() + w.numel() + y.numel()) * 2 / 1e6
    rows_out.append((us * cnt, name, M, N, KK, cnt, us, fl / us / 1e6, mb / us))
rows_out.sort(reverse=True)
tot = sum(r[0] for r in rows_out)
totf = sum(2.0 * r[2] * r[3] * r[4] * r[5] for r in rows_out)
print(f"# igemm per-shape timing, B={B}, {prec}: total {tot/1e3:.3f} ms per step for {totf/1e9:.1f} GFLOP -> {totf/tot/1e6:.1f} TFLOP/s\n")
print("| layer | M | N | K | count/step | us/launch | TFLOP/s | TB/s (x+w+y) | ms/step |")
print("|---|---|---|---|---|---|---|---|---|")
for t, name, M, N, KK, cnt, us, tf, tb in rows_out:
    print(f"| {name} | {M} | {N} | {KK} | {cnt} | {us:.1f} | {tf:.0f} | {tb:.2f} | {t/1e3:.3f} |")
