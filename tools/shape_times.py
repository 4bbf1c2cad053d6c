"""Stand-alone duration of every implicit-GEMM launch of one act() step, per shape (development build of the library:
HCM_DEV_LIB=1 HCM_IGEMM_TIME=1 HCM_GRAPH=0 HCM_SERIAL=1 python tools/shape_times.py [B] 2> table.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hcm_pkg
hcm_pkg.load()
from robo_vln_amd import synth
from robo_vln_amd.config import baseline_config
from robo_vln_amd.policy import HCMEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = baseline_config(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
hi, lo = synth.make_weights(cfg, 0)
eng = HCMEngine(cfg, hi, lo, max_batch=B, precision="fp16")
o = synth.make_observations(cfg, B, 0, 0, rgb_uint8=True)
obs = {k: torch.from_numpy(v).cuda() for k, v in o.items()}
R = cfg.num_recurrent_layers
hh = torch.zeros(R, B, cfg.hidden, device="cuda"); lh = torch.zeros_like(hh); m = torch.ones(B, device="cuda")
for _ in range(4):
    rec, hh, lh = eng.act(obs, hh, lh, m)
torch.cuda.synchronize()
eng.close()
