"""Run-to-run bit equality of the one-launch RGB stem (csrc/stem.hip) under a competing stream: N repetitions of the same launch, every output
compared with the first.  usage: python tools/stem_determinism.py [B] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
H = 256
side = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
for Cout in (64, 128):
    for red in (0, 1):
        x = (torch.rand(B, H, H, 3, device="cuda") * 255).to(torch.uint8)
        w = (torch.randn(Cout, 224, device="cuda") * 0.05).half()
        b = torch.randn(Cout, device="cuda")
        w1 = (torch.randn(Cout, 64, device="cuda") * 0.1).half()
        b1 = torch.randn(Cout, device="cuda")
        scratch = torch.empty(lib.hcm_op_stem_scratch_bytes(B, H, H), device="cuda", dtype=torch.uint8)
        ys, os_ = [], []
        bad = 0
        y0 = o0 = None
        for i in range(reps):
            y = torch.full((B, H // 4, H // 4, Cout), float("nan"), device="cuda", dtype=torch.float16)
            o = torch.full_like(y, float("nan"))
            if i % 2:
                with torch.cuda.stream(side):
                    for _ in range(3): junk @ junk
            if red:
                rc = lib.hcm_op_stem_pool_fused_red(x.data_ptr(), _lib.HCM_U8, w.data_ptr(), b.data_ptr(), y.data_ptr(), _lib.HCM_F16, B, H, H, Cout, 1 / 255.0,
                                                    scratch.data_ptr(), w1.data_ptr(), b1.data_ptr(), o.data_ptr(), None)
            else:
                rc = lib.hcm_op_stem_pool_fused(x.data_ptr(), _lib.HCM_U8, w.data_ptr(), b.data_ptr(), y.data_ptr(), _lib.HCM_F16, B, H, H, Cout, 1 / 255.0,
                                                scratch.data_ptr(), None)
            assert rc == 0
            torch.cuda.synchronize()
            if y0 is None:
                y0, o0 = y, o
            else:
                dy = (y.view(torch.int16) != y0.view(torch.int16))
                do = (o.view(torch.int16) != o0.view(torch.int16)) if red else torch.zeros(1, dtype=torch.bool, device="cuda")
                if bool(dy.any()) or bool(do.any()):
                    bad += 1
                    if bad <= 3:
                        idx = torch.nonzero(dy if bool(dy.any()) else do)[:4].tolist()
                        print(f"  rep {i}: {int(dy.sum())} pooled / {int(do.sum())} reduced elements differ, first at {idx}")
        print(f"B={B} Cout={Cout} red={red}: {bad} of {reps - 1} repetitions differ from the first")
