"""Race screen for the 256-mid-channel fused bottleneck launch: the same inputs many times, while a second stream keeps the chip busy with
other launches (uneven load), every output word compared with the first run's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import _lib
lib = _lib.lib()
CODE, tdt = _lib.HCM_F16, torch.float16
P = lambda t: t.data_ptr()
for (C1, B, H, W, stride) in ((256, 128, 16, 16, 1), (256, 3, 8, 8, 1), (256, 5, 12, 20, 1), (256, 16, 16, 16, 2), (64, 3, 17, 15, 1), (64, 5, 9, 11, 2), (128, 1, 24, 20, 1),
                              (64, 7, 16, 16, 1), (128, 5, 8, 8, 1)):
    C3, CN = 4 * C1, C1
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    x = torch.randn(B, H, W, C1, device="cuda").to(tdt)
    w2 = (torch.randn(C1, 3, 3, C1, device="cuda") * 0.03).to(tdt); b2 = torch.randn(C1, device="cuda")
    w3 = (torch.randn(C3, 1, 1, C1, device="cuda") * 0.05).to(tdt); b3 = torch.randn(C3, device="cuda")
    w1 = (torch.randn(CN, 1, 1, C3, device="cuda") * 0.03).to(tdt); b1 = torch.randn(CN, device="cuda")
    r = torch.randn(B, Ho, Wo, C3, device="cuda").to(tdt)
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
    ref = None
    bad = 0
    for it in range(300):
        y = torch.full((B, Ho, Wo, C3), float("nan"), device="cuda", dtype=tdt)
        o1 = torch.full((B, Ho, Wo, CN), float("nan"), device="cuda", dtype=tdt)
        if it % 3 == 1:
            with torch.cuda.stream(side):
                for _ in range(3): big2 = big @ big
        if it % 3 == 2:
            with torch.cuda.stream(side):
                for _ in range(20): big.add_(1.0)
        assert lib.hcm_op_bottleneck_tail_next(P(x), P(w2), P(b2), P(w3), P(b3), P(r), P(y), P(w1), P(b1), P(o1), CODE, B, H, W, C1, stride, CN, None) == 0
        torch.cuda.synchronize()
        if ref is None:
            ref = (y.clone(), o1.clone())
        else:
            if not (torch.equal(y.view(torch.int16), ref[0].view(torch.int16)) and torch.equal(o1.view(torch.int16), ref[1].view(torch.int16))):
                bad += 1
                if bad <= 3:
                    dy = (y.float() - ref[0].float()).abs(); do = (o1.float() - ref[1].float()).abs()
                    print(f"  iteration {it}: y differs in {(dy > 0).sum().item()} words (max {dy.max().item():.4g}), o1 in {(do > 0).sum().item()} (max {do.max().item():.4g})")
    print(f"C1={C1} B={B} {H}x{W} stride {stride}: {bad} of 299 repeats differ from the first run")
