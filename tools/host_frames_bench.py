"""PCIe-inclusive step time of the ways to hand frames to act() (B = 64, configs[1]): resident device frames, frames copied in front of
the step ("prestage"), and pinned host frames handed to the library (HCM_ACT_HOST_FRAMES: one copy per encoder chain) -- on torch's default
stream and on a side stream, eager engine and hipGraph engine.  usage: python tools/host_frames_bench.py [B]"""
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
o = synth.make_observations(cfg, B, 0, 0, rgb_uint8=True)
host = {"rgb": torch.from_numpy(o["rgb"]).pin_memory(), "depth": torch.from_numpy(o["depth"]).pin_memory()}
dev = {k: v.cuda() for k, v in host.items()}
ids = torch.from_numpy(o["instruction"]).cuda()
R = cfg.num_recurrent_layers
m = torch.ones(B, device="cuda")
for graph in (False, True):
    eng = HCMEngine(cfg, hi, lo, max_batch=B, precision="fp16", graph=graph)
    for side in (False, True):
        stream = torch.cuda.Stream() if side else torch.cuda.current_stream()
        for mode in ("resident", "prestage", "host_frames"):
            hh = torch.zeros(R, B, cfg.hidden, device="cuda"); lh = torch.zeros_like(hh)
            def step():
                global hh, lh
                if mode == "prestage":
                    dev["rgb"].copy_(host["rgb"], non_blocking=True); dev["depth"].copy_(host["depth"], non_blocking=True)
                if mode == "host_frames":
                    r, hh, lh = eng.act({"rgb": host["rgb"], "depth": host["depth"], "instruction": ids}, hh, lh, m, host_frames=True)
                else:
                    r, hh, lh = eng.act({"rgb": dev["rgb"], "depth": dev["depth"], "instruction": ids}, hh, lh, m)
            with torch.cuda.stream(stream):
                for _ in range(25): step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(40): step()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 40 * 1e3
            print(f"engine graph={graph!s:5} {'side stream   ' if side else 'default stream'} {mode:12} {ms:6.3f} ms/step  {B / ms * 1e3:8.0f} env-steps/s", flush=True)
    eng.close()
