import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import synth
from robo_vln_amd.config import HCMConfig
from robo_vln_amd.policy import HCMEngine, Policy, Seq2Seq_HighLevel_CMA, Seq2Seq_LowLevel
cfg = HCMConfig().validate()
hi, lo = synth.make_weights(cfg, seed=0)
for B in (1, 4):
    for graph in (False, True):
        eng = HCMEngine(cfg, hi, lo, max_batch=B, precision="fp16", graph=graph)
        H, Lw = Seq2Seq_HighLevel_CMA(eng), Seq2Seq_LowLevel(eng)
        obs = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.make_observations(cfg, B, step=0, seed=0).items()}
        R = cfg.num_recurrent_layers
        hh = torch.zeros(R, B, cfg.hidden, device="cuda"); lh = torch.zeros(R, B, cfg.hidden, device="cuda")
        prev = torch.zeros(B, 2, device="cuda"); m = torch.ones(B, 2, device="cuda")
        def split():
            global hh, lh
            o = dict(obs)
            out, hh2 = H((o, hh, prev, m))
            pred = torch.argmax(out, dim=1)
            o2 = dict(obs)
            vel, stop, lh2 = Lw((o2, lh, prev, m, pred))
            return vel
        def fused():
            return eng.act(obs, hh, lh, m[:, 0])[0]
        for name, fn in (("high_level() + argmax + low_level()", split), ("act()", fused)):
            for _ in range(5): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): r = fn()
            torch.cuda.synchronize()
            print(f"B={B} graph={graph} {name}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per step", flush=True)
        eng.close()
