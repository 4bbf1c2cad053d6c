import sys; sys.path.insert(0, ".")
import hcm_pkg; hcm_pkg.load()
from tests import parity_util
from oracle import cases
for name in cases.CASES:
    for prec in ("fp16", "fp32"):
        rep = parity_util.run_case(name, prec, taps=False)
        print(name, prec, "max_abs per step:", ["%.2e" % s["max_abs"] for s in rep["steps"]], "hidden rel:", "%.1e %.1e" % (rep["hi_hidden"][3], rep["lo_hidden"][3]))
rep = parity_util.run_case("cfg1_256_L80_N1", "fp16", taps=False, batch=64)
print("cfg1 B=64 bf16:", ["%.2e" % s["max_abs"] for s in rep["steps"]])
