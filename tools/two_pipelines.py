"""Two half-batch engines stepped concurrently in ONE process (each on its own stream, each a captured hipGraph) against one engine at the
full batch: does a second, phase-shifted pipeline fill the first one's gaps?  usage: python tools/two_pipelines.py [B=64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hcm_pkg
hcm_pkg.load()
from robo_vln_amd import synth
from robo_vln_amd.config import baseline_config
from robo_vln_amd.policy import HCMEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = baseline_config(1)
hi, lo = synth.make_weights(cfg, 0)
R = cfg.num_recurrent_layers


def make(b):
    eng = HCMEngine(cfg, hi, lo, max_batch=b, precision="fp16", graph=True)
    o = synth.make_observations(cfg, b, 0, 0, rgb_uint8=True)
    obs = {k: torch.from_numpy(v).cuda() for k, v in o.items()}
    st = {"hh": torch.zeros(R, b, cfg.hidden, device="cuda"), "lh": torch.zeros(R, b, cfg.hidden, device="cuda"), "m": torch.ones(b, device="cuda")}
    return eng, obs, st


def run(engs, n):
    streams = [torch.cuda.Stream() for _ in engs]
    def step():
        for (eng, obs, st), s in zip(engs, streams):
            with torch.cuda.stream(s):
                _, st["hh"], st["lh"] = eng.act(obs, st["hh"], st["lh"], st["m"])
    for _ in range(20): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


one = [make(B)]
ms1 = run(one, 60)
print(f"one engine  B={B}:      {ms1:.3f} ms/step  {B / ms1 * 1e3:8.0f} env-steps/s", flush=True)
one[0][0].close()
two = [make(B // 2), make(B // 2)]
ms2 = run(two, 60)
print(f"two engines B={B // 2} each: {ms2:.3f} ms per pair of steps  {B / ms2 * 1e3:8.0f} env-steps/s", flush=True)
