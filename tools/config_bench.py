"""Timing of the BASELINE.json configurations that are NOT the bench line (parity-test shapes), for DESIGN.md:
configs[3] low-level model with SimpleCNN encoders at B=256 (memory-bound), configs[4] high-level model with L=160, N=6
at B=128 (MFMA-bound).  usage: python tools/config_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import synth
from robo_vln_amd.config import HCMConfig
from robo_vln_amd.policy import HCMEngine

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

# configs[4]: hi model, B=128, L=160, N=6
cfg = HCMConfig(instr_len=160, vla_layers=6).validate(); B = 128
hi_sd = synth.materialize(synth.high_level_spec(cfg), "hi", 0)
eng = HCMEngine(cfg, hi_sd, None, max_batch=B, precision="fp16")
obs = {k: torch.from_numpy(v).cuda() for k, v in synth.make_observations(cfg, B, rgb_uint8=True).items()}
h = torch.zeros(cfg.num_recurrent_layers, B, cfg.hidden, device="cuda"); m = torch.ones(B, device="cuda")
dt = timeit(lambda: eng.high_forward(dict(obs), h, m))
gf = 42.87
print(f"configs[4] hi model B=128 L=160 N=6: {dt * 1e3:.2f} ms/step, {B / dt:.0f} env-steps/s, {B / dt * gf / 1e3:.0f} TFLOP/s ({B / dt * gf / 1e3 / 2500 * 100:.1f} % of bf16 dense peak)")
eng.close()
# configs[3]: lo model with SimpleCNN encoders, B=256
cfg = HCMConfig(depth_encoder="SimpleDepthCNN", rgb_encoder="SimpleRGBCNN").validate(); B = 256
lo_sd = synth.materialize(synth.low_level_spec(cfg), "lo", 0)
eng = HCMEngine(cfg, None, lo_sd, max_batch=B, precision="fp16")
obs = {k: torch.from_numpy(v).cuda() for k, v in synth.make_observations(cfg, B, rgb_uint8=True).items()}
h = torch.zeros(cfg.num_recurrent_layers, B, cfg.hidden, device="cuda"); m = torch.ones(B, device="cuda")
st = torch.zeros(B, dtype=torch.int64, device="cuda")
dt = timeit(lambda: eng.low_forward(obs, h, m, st))
mb = B * (256 * 256 * 3 + 256 * 256 * 4) / 1e6
print(f"configs[3] lo model SimpleCNN encoders B=256: {dt * 1e3:.3f} ms/step, {B / dt:.0f} env-steps/s, input frames {mb:.0f} MB -> {mb / dt / 1e6:.2f} TB/s of frame reads")
eng.close()
