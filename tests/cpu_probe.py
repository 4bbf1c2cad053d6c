"""Probe of the GPU box's host CPU: how the torch-CPU oracle scales with thread count (for bench.py's cpu_baseline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import synth
from robo_vln_amd.config import HCMConfig
from oracle import hcm_oracle
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
try:
    print([l for l in open("/proc/cpuinfo") if "model name" in l][0].strip())
except Exception: pass
cfg = HCMConfig()
hi_sd, lo_sd = synth.make_weights(cfg, 0)
ora = hcm_oracle.PolicyOracle(cfg, hi_sd, lo_sd)
B = 8
obs = synth.make_observations(cfg, B, 0, 0)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    hh = torch.zeros(2, B, 512); lh = torch.zeros(2, B, 512)
    ora.act(obs, hh, lh, np.zeros(B, np.float32))
    t0 = time.time(); ora.act(obs, hh, lh, np.ones(B, np.float32)); dt = time.time() - t0
    print(f"threads {nt}: {B/dt:.2f} env-steps/s ({dt:.2f}s per step of B={B})", flush=True)
    if dt > 20: break
