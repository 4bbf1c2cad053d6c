"""Determinism probe for tests/test_fusion_toggles_gpu.py: run the test's step script several times per environment and compare bitwise."""
import importlib.util, os, sys, tempfile, numpy as np
spec = importlib.util.spec_from_file_location("tg", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "test_fusion_toggles_gpu.py"))
tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
envs = {"default": {}, "no256": {"HCM_NO_BNECK256": "1"}, "novla": {"HCM_NO_VLA_FUSE": "1"}, "nodsfold": {"HCM_NO_BNECK_DSFOLD": "1"}}
res = {}
with tempfile.TemporaryDirectory() as d:
    for name, e in envs.items():
        res[name] = [tg._run(e, os.path.join(d, f"{name}{i}.npz")) for i in range(N)]
for name, rs in res.items():
    diffs = [max(float(np.abs(r[k] - rs[0][k]).max()) for k in ("rec", "hh", "lh")) for r in rs]
    print(name, "deterministic" if max(diffs) == 0 else "NON-DETERMINISTIC", diffs)
