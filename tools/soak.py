"""Soak run of one engine the way a long evaluation drives it: a different unpadded instruction length every "episode" (graph cache churn:
the library keeps 8 captured graphs), random episode resets, ragged batches now and then, device memory watched for growth.
usage: python tools/soak.py [steps=1500] [B=8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import synth
from robo_vln_amd.config import HCMConfig
from robo_vln_amd.policy import HCMEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = HCMConfig(rgb_hw=128, depth_hw=128, bert_layers=2).validate()
hi, lo = synth.make_weights(cfg, seed=0)
eng = HCMEngine(cfg, hi, lo, max_batch=B, precision="fp16", max_instr_len=256, graph=True)
rng = np.random.default_rng(0)
obs0 = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.make_observations(cfg, B, step=0, seed=0, rgb_uint8=True).items()}
R = cfg.num_recurrent_layers
hh = torch.zeros(R, B, cfg.hidden, device="cuda"); lh = torch.zeros(R, B, cfg.hidden, device="cuda")
free0 = None
t0 = time.time()
L = 80
for t in range(steps):
    if t % 7 == 0:                                   # new episode: another instruction length
        L = int(rng.integers(3, 256))
    ids = torch.from_numpy(rng.integers(1000, cfg.bert_vocab, size=(B, L)).astype(np.int64)).cuda()
    obs = dict(obs0); obs["instruction"] = ids
    if t % 11 == 5:                                  # ragged batch
        obs["instruction_lengths"] = torch.from_numpy(rng.integers(1, L + 1, size=(B,)).astype(np.int32)).cuda()
    mask = torch.from_numpy((rng.random(B) > 0.1).astype(np.float32)).cuda()
    rec, hh, lh = eng.act(obs, hh, lh, mask)
    if t % 100 == 99:
        torch.cuda.synchronize()
        assert torch.isfinite(rec).all() and torch.isfinite(hh).all() and torch.isfinite(lh).all(), t
        free, total = torch.cuda.mem_get_info()
        if free0 is None: free0 = free
        print(f"step {t + 1}: L={L} free {free / 2**20:.0f} MiB (drift {(free0 - free) / 2**20:+.1f} MiB) graph/eager {eng.query(7)}/{eng.query(8)}  {(t + 1) / (time.time() - t0):.0f} steps/s", flush=True)
eng.close()
print("soak ok")
