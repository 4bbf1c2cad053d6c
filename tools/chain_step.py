"""A few act() steps at B (default 64) on the DEV library, for tracing one encoder chain alone (HCM_SKIP mask set by the caller:
12 RGB only, 11 depth only, 7 BERT only).  usage: HCM_DEV_LIB=1 HCM_SKIP=12 rocprofv3 --kernel-trace ... -- python tools/chain_step.py [B] [graph]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import synth
from robo_vln_amd.config import HCMConfig
from robo_vln_amd.policy import HCMEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = HCMConfig().validate()
hi, lo = synth.make_weights(cfg, seed=0)
eng = HCMEngine(cfg, hi, lo, max_batch=B, precision="fp16", graph=len(sys.argv) > 2)
obs = {k: torch.from_numpy(v).cuda() for k, v in synth.make_observations(cfg, B, rgb_uint8=True).items()}
R = cfg.num_recurrent_layers
hh = torch.zeros(R, B, cfg.hidden, device="cuda"); lh = torch.zeros(R, B, cfg.hidden, device="cuda"); m = torch.ones(B, device="cuda")
for _ in range(6): eng.act(obs, hh, lh, m)
torch.cuda.synchronize()
