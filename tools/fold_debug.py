"""Range-fold exactness probe: the engine built from depth weights scaled by 2^16 (folded by the calibration) against the engine of the
unscaled weights, tap by tap."""
import sys; sys.path.insert(0, ".")
import hcm_pkg; hcm_pkg.load()
import numpy as np, torch
from robo_vln_amd import synth
from robo_vln_amd.config import HCMConfig
from robo_vln_amd.policy import HCMEngine

cfg = HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=2).validate()
hi, lo = synth.make_weights(cfg, seed=3)
k = "depth_encoder.visual_encoder.backbone.layer1.0.convs.3.weight"
def scaled(f):
    a, b = dict(hi), dict(lo)
    a[k] = a[k] * np.float32(f); b[k] = b[k] * np.float32(f)
    return a, b
obs = {kk: torch.from_numpy(v).cuda() for kk, v in synth.make_observations(cfg, 2, seed=3).items()}
R = cfg.num_recurrent_layers
z = torch.zeros(R, 2, cfg.hidden, device="cuda")
out = {}
for name, f in (("x1", 1.0), ("x8", 8.0), ("x2^16", 65536.0)):
    a, b = scaled(f)
    eng = HCMEngine(cfg, a, b, max_batch=2, precision="fp16")
    eng.enable_taps(True)
    rec, _, _ = eng.act(obs, z, z, torch.zeros(2, device="cuda"))
    taps = {t: eng.get_tap(t) for t in ("pair.depth_conv1", "hi.depth_spatial", "hi.depth_kv", "hi.rnn_in", "lo.rnn_in")}
    out[name] = (rec.cpu().numpy(), taps, eng.calibration_report())
    eng.close()
for name in ("x8", "x2^16"):
    print(name, out[name][2])
    print("  record diff vs x1:", np.abs(out[name][0] - out["x1"][0]).max())
    for t in out[name][1]:
        print("  tap", t, np.abs(out[name][1][t] - out["x1"][1][t]).max(), "of", np.abs(out["x1"][1][t]).max())
