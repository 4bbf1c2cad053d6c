import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import synth
from robo_vln_amd.config import baseline_config
from robo_vln_amd.policy import HCMEngine
cfg = baseline_config(1); B = 64
hi_sd, lo_sd = synth.make_weights(cfg, seed=0)
for k, v in hi_sd.items():
    if k.startswith(("rgb_encoder.cnn.", "depth_encoder.visual_encoder.")): lo_sd[k] = v
eng = HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="fp16", graph=True)
obs = {k: torch.from_numpy(v).cuda() for k, v in synth.make_observations(cfg, B, rgb_uint8=True).items()}
hh = torch.zeros(2, B, 512, device="cuda"); lh = torch.zeros(2, B, 512, device="cuda"); m = torch.ones(B, device="cuda")
for reuse in (False, True):
    for _ in range(4): r, hh, lh = eng.act(obs, hh, lh, m, reuse_instruction=reuse)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): r, hh, lh = eng.act(obs, hh, lh, m, reuse_instruction=reuse)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"identical hi/lo trunk weights (shared trunks), reuse_instruction={reuse}: {dt*1e3:.2f} ms/step, {B/dt:.0f} env-steps/s")
