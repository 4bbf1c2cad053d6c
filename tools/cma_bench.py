"""CMANet step time at the BASELINE frame size (B=64, 256x256 RGB-D, L=80, bidirectional LSTM instruction encoder)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd.config import CMAConfig
from robo_vln_amd import synth
from robo_vln_amd.cma import CMAEngine
L = int(sys.argv[sys.argv.index("--L") + 1]) if "--L" in sys.argv else 80
cfg = CMAConfig(instr_len=L).validate(); B = 64
eng = CMAEngine(cfg, synth.make_cma_weights(cfg, 0), max_batch=B, precision="fp16", graph="--no-graph" not in sys.argv)
obs = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.make_cma_observations(cfg, B, rgb_uint8=True).items()}
hid = torch.zeros(cfg.num_recurrent_layers, B, cfg.hidden, device="cuda"); m = torch.ones(B, device="cuda")
for _ in range(60): out, stop, hid = eng.forward(obs, hid, m)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(60): out, stop, hid = eng.forward(obs, hid, m)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60
print(f"CMANet B=64 256x256 L={L} bf16 (graph={eng._graph}, graph launches {eng.query(7)}): {dt*1e3:.2f} ms/step, {B/dt:.0f} env-steps/s")
