"""Integration check of the fused launches of the RGB trunk (bottleneck tail, next block's reduction, horizontal half of the stem
max-pool) and of the serial tail (argmax + sub-task embedding inside the high-level cell kernel): they are bit-identical rewrites, so a whole act() step with them must equal the step with every one of them switched off
(HCM_NO_* knobs; read once per process, hence sub-processes).  The down-sample fold changes one rounding and is compared to
tolerance.  Covers the slot / pointer plumbing in forward.cpp that the operator-level tests cannot see."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import synth
from robo_vln_amd.config import HCMConfig
from robo_vln_amd.policy import HCMEngine
cfg = HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, vla_layers=1, bert_layers=1).validate()
B = 3
hi_sd, lo_sd = synth.make_weights(cfg, seed=5)
eng = HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="bf16", graph=False)
obs = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.make_observations(cfg, B, step=0, seed=5).items()}
R = cfg.num_recurrent_layers
hh = torch.zeros(R, B, cfg.hidden, device="cuda"); lh = torch.zeros(R, B, cfg.hidden, device="cuda")
rec, hh2, lh2 = eng.act(obs, hh, lh, torch.zeros(B, device="cuda"))
torch.cuda.synchronize()
np.savez(sys.argv[1], rec=rec.cpu().numpy(), hh=hh2.cpu().numpy(), lh=lh2.cpu().numpy())
eng.close()
""" % ROOT


def _run(env_extra, path):
    env = dict(os.environ)
    env.update(env_extra)
    subprocess.run([sys.executable, "-c", SCRIPT, path], check=True, env=env, cwd=ROOT, timeout=600)
    return dict(np.load(path))


def test_fused_rgb_trunk_launches_equal_the_separate_ones():
    with tempfile.TemporaryDirectory() as d:
        # the down-sample fold off in both runs: everything else must then agree to the bit
        fused = _run({"HCM_NO_BNECK_DSFOLD": "1"}, os.path.join(d, "a.npz"))
        plain = _run({"HCM_NO_BNECK_DSFOLD": "1", "HCM_NO_BNECK_FUSE": "1", "HCM_NO_STEM_HPOOL": "1", "HCM_NO_PRED_FUSE": "1"}, os.path.join(d, "b.npz"))
        nonext = _run({"HCM_NO_BNECK_DSFOLD": "1", "HCM_NO_BNECK_NEXT": "1"}, os.path.join(d, "c.npz"))
        default = _run({}, os.path.join(d, "e.npz"))
    for k in ("rec", "hh", "lh"):
        assert np.array_equal(fused[k], plain[k]), k
        assert np.array_equal(fused[k], nonext[k]), k
    assert np.isfinite(default["rec"]).all()
    # shipped configuration (down-sample conv folded into the expansion GEMM): one rounding fewer on that path
    assert np.abs(default["rec"] - plain["rec"]).max() <= 1e-2
