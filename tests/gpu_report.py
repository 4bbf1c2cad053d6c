"""Diagnostic script for a GPU box: prints per-tap / per-step parity of the golden cases (not a test)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcm_pkg
hcm_pkg.load()
from tests import parity_util

names = sys.argv[2:] or ["cfg0_128_L20_N2"]
precs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fp32", "bf16"]
for n in names:
    for p in precs:
        try:
            print(parity_util.format_report(parity_util.run_case(n, p)), flush=True)
        except Exception as e:
            import traceback
            traceback.print_exc()
            print(f"== {n} [{p}] FAILED: {e}", flush=True)
