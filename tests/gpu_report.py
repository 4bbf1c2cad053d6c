"""Diagnostic script for a GPU box: prints per-tap / per-step parity of the golden cases (not a test).
usage: gpu_report.py <precisions> <batch|-> [sub=depth:bf16,bert:fp16] case..."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcm_pkg
hcm_pkg.load()
from tests import parity_util

precs = sys.argv[1].split(",")
batch = None if sys.argv[2] == "-" else int(sys.argv[2])
rest = sys.argv[3:]
sub = None
if rest and rest[0].startswith("sub="):
    sub = dict(kv.split(":") for kv in rest[0][4:].split(",") if kv)
    rest = rest[1:]
for n in rest or ["cfg0_128_L20_N2"]:
    for p in precs:
        try:
            print(parity_util.format_report(parity_util.run_case(n, p, sub_precision=sub if p == "bf16" else None, batch=batch)), "sub=", sub, "batch=", batch, flush=True)
        except Exception as e:
            import traceback
            traceback.print_exc()
            print(f"== {n} [{p}] FAILED: {e}", flush=True)
