"""Which replay form is flaky?  Every engine runs every step TWICE from the same inputs; counts steps where an engine disagrees with itself and where
the forms disagree with each other.  usage: step_determinism.py [B] [steps]   env R4_MODES=forked,eager[,chain] R4_FULL=1 R4_DEPTH_HW=<n>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import synth
from robo_vln_amd.config import HCMConfig
from robo_vln_amd.policy import HCMEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
# R4_FULL=1: BASELINE configs[1] (256-pixel frames, L = 80, 12 BERT layers) instead of the small configuration; R4_DEPTH_HW=<n>: depth frame size
cfg = (HCMConfig() if os.environ.get("R4_FULL") else
       HCMConfig(rgb_hw=128, depth_hw=int(os.environ.get("R4_DEPTH_HW", "128")), instr_len=20, bert_layers=2)).validate()
hi, lo = synth.make_weights(cfg, seed=7)
modes = {"chain": dict(graph=True, chain_graphs=True), "forked": dict(graph=True, chain_graphs=False), "eager": dict(graph=False)}
want = os.environ.get("R4_MODES", "forked,eager").split(",")
engs = {k: HCMEngine(cfg, hi, lo, max_batch=B, precision="fp16", **modes[k]) for k in want}
frames = [synth.make_observations(cfg, B, step=t, seed=7) for t in range(8)]
obs = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in frames[0].items()}
R = cfg.num_recurrent_layers
hh0 = torch.zeros(R, B, cfg.hidden, device="cuda"); lh0 = torch.zeros(R, B, cfg.hidden, device="cuda")
m = torch.ones(B, device="cuda")
selfbad = {k: 0 for k in engs}; cross = 0; shown = 0
for t in range(T):
    f = frames[t % 8]
    obs["rgb"].copy_(torch.from_numpy(f["rgb"]).cuda()); obs["depth"].copy_(torch.from_numpy(f["depth"]).cuda())
    torch.cuda.synchronize()
    outs = {}
    for k, e in engs.items():
        runs = []
        for rep in range(2):
            r, hh, lh = e.act(obs, hh0, lh0, m)
            runs.append((r.clone(), hh.clone(), lh.clone()))
            torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(*runs))
        if not same:
            selfbad[k] += 1
            if shown < 6:
                shown += 1
                print(f"step {t}: {k} run 1 != run 2: " + " ".join(f"{n} {float((a - b).abs().max()):.2e}" for n, a, b in zip(("rec", "hh", "lh"), *runs)))
        outs[k] = runs
    ks = list(engs)
    if not all(torch.equal(a, b) for a, b in zip(outs[ks[0]][0], outs[ks[-1]][0])):
        cross += 1
    hh0, lh0 = outs[ks[-1]][1][1], outs[ks[-1]][1][2]
print(f"B={B} steps={T}: steps where a form disagreed with itself {selfbad}; first form != last form (first runs) {cross}")
