"""Stand-alone timing of the serial tail: act() with the encoders' outputs cached is not possible from outside, so time whole steps at B=64
with the fused cross-modal layer on / off (HCM_NO_VLA_FUSE) -- and the kernel itself from the trace.  usage: python tools/vla_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hcm_pkg
hcm_pkg.load()
from robo_vln_amd import synth
from robo_vln_amd.config import baseline_config
from robo_vln_amd.policy import HCMEngine
cfg = baseline_config(1); B = 64
hi, lo = synth.make_weights(cfg, 0)
eng = HCMEngine(cfg, hi, lo, max_batch=B, precision="fp16", graph=True)
o = synth.make_observations(cfg, B, 0, 0, rgb_uint8=True)
obs = {k: torch.from_numpy(v).cuda() for k, v in o.items()}
R = cfg.num_recurrent_layers
hh = torch.zeros(R, B, cfg.hidden, device="cuda"); lh = torch.zeros_like(hh); m = torch.ones(B, device="cuda")
for _ in range(60): rec, hh, lh = eng.act(obs, hh, lh, m)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): rec, hh, lh = eng.act(obs, hh, lh, m)
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 200 * 1e3)
