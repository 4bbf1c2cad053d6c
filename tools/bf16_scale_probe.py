import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcm_pkg; hcm_pkg.load()
from tests.test_bf16_margin_gpu import _rollout
from robo_vln_amd.config import HCMConfig
cfg = HCMConfig().validate()
for sc in (0.5, 1.0, 1.41, 2.0):
    for prec in ("fp32", "fp16", "bf16"):
        try:
            _rollout(prec, sc, 8, 4, [0, 3, 7], cfg, tol=10.0)
        except Exception as e:
            print("ERR", prec, sc, str(e)[:200])
