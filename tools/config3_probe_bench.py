"""BASELINE.json configs[3] micro-benchmark (SURVEY 8d "Config 4", HBM-bound): SimpleDepthCNN -> one token -> Visual_Ling_Attn(N=1)
at B=256, depth 256x256, L=80, pre-computed instruction tensor.  GPU time by HIP events around the whole operator chain
(ctypes launch overhead overlaps: the chain is enqueued asynchronously), algorithmic bytes per SURVEY 8d."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import synth
from robo_vln_amd.config import HCMConfig
from robo_vln_amd.probe import DepthCnnVlaProbe
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
cfg = HCMConfig(vla_layers=1).validate(); B, L = 256, cfg.instr_len
cnn_sd = synth.materialize(synth.simple_cnn_spec("", 1, 256, 128), "probe_cnn", 0)
vla_sd = synth.materialize(synth.vla_spec("", cfg, vis_in=128), "probe_vla", 0)
tdt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[prec]
probe = DepthCnnVlaProbe(cnn_sd, vla_sd, precision=prec, fused_layer=os.environ.get("PROBE_UNFUSED") is None, overlap=os.environ.get("PROBE_SERIAL") is None, graph=os.environ.get("PROBE_GRAPH") is not None)
depth = torch.rand(B, 256, 256, 1, device="cuda"); ins = (torch.rand(B, L, 768, device="cuda") * 2 - 1).to(tdt)
for _ in range(3): out = probe.forward(depth, ins)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
e0.record()
for _ in range(n): out = probe.forward(depth, ins)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
esz = ins.element_size()
alg = B * (256 * 256 * 4 + L * 768 * esz + L * 256 * esz) + (3.26e6 + 1.02e6) * esz       # depth in + ins in + out + weights once
print(f"configs[3] probe [{prec}] B={B}: {ms:.3f} ms/step, {B / ms * 1e3:.0f} samples/s; algorithmic {alg / 1e6:.0f} MB -> {alg / ms / 1e9:.3f} TB/s "
      f"({alg / ms / 1e9 / 8.0 * 100:.1f} % of 8 TB/s)")
