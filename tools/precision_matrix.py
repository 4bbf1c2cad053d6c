"""Record error vs the CPU oracle per storage-type assignment of the four 16-bit sub-networks (depth / bert / vla / rgb), at the
full batch over three consecutive steps.  Usage: python tools/precision_matrix.py [case ...]"""
import sys; sys.path.insert(0, ".")
import hcm_pkg; hcm_pkg.load()
import torch
from tests import parity_util

torch.set_num_threads(min(16, torch.get_num_threads()))
COMBOS = [("all fp16", {}),
          ("all bf16", dict(depth="bf16", bert="bf16", vla="bf16", rgb="bf16")),
          ("bert bf16", dict(bert="bf16")),
          ("bert+vla bf16", dict(bert="bf16", vla="bf16")),
          ("bert+vla+rgb bf16", dict(bert="bf16", vla="bf16", rgb="bf16")),
          ("depth bf16", dict(depth="bf16")),
          ("rgb bf16", dict(rgb="bf16"))]
CASES = [("cfg1_256_L80_N1", 64), ("gru_128_L20", 64), ("cfg0_128_L20_N2", 16)]
names = sys.argv[1:]
for case, B in CASES:
    if names and case not in names:
        continue
    for label, sub in COMBOS:
        rep = parity_util.run_case(case, "fp16", taps=False, batch=B, steps=3, sub_precision=sub)
        print(f"{case} B={B} [{label}]: max_abs per step", ["%.2e" % s["max_abs"] for s in rep["steps"]],
              "hidden rel %.1e %.1e" % (rep["hi_hidden"][3], rep["lo_hidden"][3]), flush=True)
