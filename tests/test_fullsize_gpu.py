"""BASELINE.json configs[3] and configs[4] at their FULL sizes on the GPU (configs[1] at B=64: tests/test_properties_gpu.py,
tests/test_parity_gpu.py::test_baseline_config2_batch64_bf16).  The CPU oracle is too slow for a whole batch, and it does not
have to be: every op of the path is per-sample, so a few rows of the full-size call are checked against the oracle
run on those rows alone, and the rest through size-independent properties: determinism, batch-permutation equivariance, a row
of the big call equals the same environment run alone."""
import numpy as np
import pytest
import torch

from oracle import cases, hcm_oracle
from robo_vln_amd import synth
from robo_vln_amd.config import HCMConfig, baseline_config

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ configs[4]: ResNet50 RGB + 6-layer decoder, L=160, B=128
@pytest.fixture(scope="module")
def cfg4():
    from robo_vln_amd.policy import HCMEngine
    cfg = baseline_config(4).validate()
    assert cfg.instr_len == 160 and cfg.vla_layers == 6 and cfg.rgb_hw == 256
    B = 128
    hi_sd = synth.materialize(synth.high_level_spec(cfg), "hi", cases.SEED)
    eng = HCMEngine(cfg, hi_sd, None, max_batch=B, precision="fp16")
    obs_np = synth.make_observations(cfg, B, step=0, seed=7, rgb_uint8=True)
    obs = {k: torch.from_numpy(v).cuda() for k, v in obs_np.items()}
    R = cfg.num_recurrent_layers
    hh = ((torch.rand(R, B, cfg.hidden, generator=torch.Generator().manual_seed(2)) - 0.5) * 0.5)
    mask = torch.ones(B)
    mask[::5] = 0
    yield cfg, eng, hi_sd, obs_np, obs, hh, mask, B
    eng.close()


def _hi(eng, obs, hh, mask):
    logits, h = eng.high_forward(dict(obs), hh.cuda(), mask.cuda())
    torch.cuda.synchronize()
    return logits.clone(), h.clone()


def test_config4_full_size_rows_match_oracle(cfg4):
    """B=128, 256x256, L=160, N=6, high-level model, 16-bit path: three rows against the CPU oracle (1e-2 on the logits, 1e-2
    relative on the hidden state)."""
    cfg, eng, hi_sd, obs_np, obs, hh, mask, B = cfg4
    logits, h = _hi(eng, obs, hh, mask)
    assert logits.shape == (B, 4) and torch.isfinite(logits).all()
    rows = [0, 61, 127]
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ora = hcm_oracle.HighLevelOracle(cfg, hi_sd)
    sub = {k: v[rows].astype(np.float32) if k == "rgb" else v[rows] for k, v in obs_np.items()}
    ref, rh = ora.forward(sub, hh[:, rows], mask[rows])
    err = (logits.cpu()[rows] - ref).abs().max().item()
    rel = (torch.linalg.norm(h.cpu()[:, rows] - rh) / torch.linalg.norm(rh)).item()
    print(f"configs[4] B=128 full size: logits max-abs {err:.3e}, hidden rel-l2 {rel:.3e}")
    assert err <= 1e-2 and rel <= 1e-2


def test_config4_full_size_properties(cfg4):
    cfg, eng, hi_sd, obs_np, obs, hh, mask, B = cfg4
    a = _hi(eng, obs, hh, mask)
    b = _hi(eng, obs, hh, mask)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])                       # deterministic
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5))
    p = _hi(eng, {k: v[perm.cuda()].contiguous() for k, v in obs.items()}, hh[:, perm].contiguous(), mask[perm].contiguous())
    assert torch.equal(p[0], a[0][perm.cuda()]) and torch.equal(p[1], a[1][:, perm.cuda()])   # permutation equivariant, bit for bit
    for rows in ([3], [100, 101, 102]):                                              # a row == the environment run alone
        idx = torch.tensor(rows)
        s = _hi(eng, {k: v[idx.cuda()].contiguous() for k, v in obs.items()}, hh[:, idx].contiguous(), mask[idx].contiguous())
        assert (s[0] - a[0][idx.cuda()]).abs().max().item() <= 2e-3
        assert (s[1] - a[1][:, idx.cuda()]).abs().max().item() <= 2e-3


# ------------------------------------------------------------------ configs[3]: SimpleDepthCNN + 1-layer VLA, B=256
def _probe_inputs(B, L):
    depth = synth.uniform01("probe/depth", B * 256 * 256, 3).reshape(B, 256, 256, 1)
    ins = (synth.uniform01("probe/ins", B * L * 768, 3).reshape(B, L, 768) * 2 - 1).astype(np.float32)
    return depth, ins


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_config3_full_size_probe(precision):
    """SimpleDepthCNN(obs,128) -> one visual token -> Visual_Ling_Attn(N=1, vis_in=128) at B=256 (SURVEY 8a note on config 4):
    rows vs the oracle restatements of the two reference classes, determinism, permutation, row-vs-alone."""
    from robo_vln_amd.probe import DepthCnnVlaProbe
    cfg = HCMConfig(vla_layers=1).validate()
    B, L = 256, cfg.instr_len
    cnn_sd = synth.materialize(synth.simple_cnn_spec("", 1, cfg.depth_hw, 128), "probe_cnn", 0)
    vla_sd = synth.materialize(synth.vla_spec("", cfg, vis_in=128), "probe_vla", 0)
    depth, ins = _probe_inputs(B, L)
    tdt = {"fp16": torch.float16, "bf16": torch.bfloat16}[precision]
    probe = DepthCnnVlaProbe(cnn_sd, vla_sd, depth_hw=256, instr_len=L, precision=precision)
    d_dev, i_dev = torch.from_numpy(depth).cuda(), torch.from_numpy(ins).to(tdt).cuda()
    out = probe.forward(d_dev, i_dev)
    torch.cuda.synchronize()
    assert out.shape == (B, L, 256) and torch.isfinite(out.float()).all()
    rows = [0, 100, 255]
    tok = hcm_oracle.simple_depth_cnn(torch.from_numpy(depth[rows]), hcm_oracle.Weights(cnn_sd))
    ref = hcm_oracle.visual_ling_attn(torch.from_numpy(ins[rows]).to(tdt).float(), tok[:, None, :], hcm_oracle.Weights(vla_sd), 1, cfg.vla_heads)
    dd = out[rows].float().cpu() - ref
    rel = (dd.norm() / ref.norm()).item()
    print(f"configs[3] probe B=256 [{precision}]: rel-l2 {rel:.3e}, max-abs {dd.abs().max().item():.3e}")
    assert rel <= (2e-3 if precision == "fp16" else 1e-2)
    again = probe.forward(d_dev, i_dev)
    torch.cuda.synchronize()
    assert torch.equal(out, again)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
    p = probe.forward(d_dev[perm].contiguous(), i_dev[perm].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(p, out[perm])
    one = probe.forward(d_dev[7:8].contiguous(), i_dev[7:8].contiguous())
    torch.cuda.synchronize()
    assert (one.float() - out[7:8].float()).abs().max().item() <= (4e-3 if precision == "fp16" else 3e-2)
    # graph replay (all launches of the probe captured once) equals the eager launches bit for bit
    gp = DepthCnnVlaProbe(cnn_sd, vla_sd, depth_hw=256, instr_len=L, precision=precision, graph=True)
    for _ in range(2):
        g_out = gp.forward(d_dev, i_dev)
        torch.cuda.synchronize()
        assert torch.equal(g_out, out)
    # more distinct input buffers than the in-place captures: the remaining ones go through the static-input graph
    keep = []
    for k in range(gp._INPLACE + 2):
        dk, ik = d_dev.clone(), i_dev.clone()
        keep.append((dk, ik))
        g_out = gp.forward(dk, ik)
        torch.cuda.synchronize()
        assert torch.equal(g_out, out), k
    assert (B,) in gp._graphs
    # the one-launch cross-modal layer against the launch-per-op form: LayerNorm reduction order only
    up = DepthCnnVlaProbe(cnn_sd, vla_sd, depth_hw=256, instr_len=L, precision=precision, fused_layer=False, overlap=False)
    u_out = up.forward(d_dev, i_dev)
    torch.cuda.synchronize()
    assert (u_out.float() - out.float()).abs().max().item() <= (1.6e-2 if precision == "fp16" else 6.4e-2)      # a few ulps at |x| ~ 4


def test_config3_encoder_low_level_model_full_size():
    """The model the reference CAN build with configs[3]'s encoders: Seq2Seq_LowLevel with SimpleDepthCNN + SimpleRGBCNN, B=256."""
    from robo_vln_amd.policy import HCMEngine
    cfg = HCMConfig(depth_encoder="SimpleDepthCNN", rgb_encoder="SimpleRGBCNN").validate()
    B = 256
    lo_sd = synth.materialize(synth.low_level_spec(cfg), "lo", cases.SEED)
    eng = HCMEngine(cfg, None, lo_sd, max_batch=B, precision="fp16")
    obs_np = synth.make_observations(cfg, B, step=1, seed=9, rgb_uint8=True)
    obs = {k: torch.from_numpy(v).cuda() for k, v in obs_np.items() if k != "instruction"}
    R = cfg.num_recurrent_layers
    lh = (torch.rand(R, B, cfg.hidden, generator=torch.Generator().manual_seed(4)) - 0.5) * 0.5
    mask = torch.ones(B)
    mask[::9] = 0
    st = torch.from_numpy(cases.fixed_subtask(B, 1))
    vel, stop, h = eng.low_forward(obs, lh.cuda(), mask.cuda(), st.cuda())
    v2, s2, h2 = eng.low_forward(obs, lh.cuda(), mask.cuda(), st.cuda())
    torch.cuda.synchronize()
    assert torch.equal(vel, v2) and torch.equal(stop, s2) and torch.equal(h, h2)
    rows = [0, 128, 255]
    sub = {"rgb": obs_np["rgb"][rows].astype(np.float32), "depth": obs_np["depth"][rows]}
    rv, rs, rh = hcm_oracle.LowLevelOracle(cfg, lo_sd).forward(sub, lh[:, rows], mask[rows], st[rows])
    err = max((vel.cpu()[rows] - rv).abs().max().item(), (stop.cpu()[rows] - rs).abs().max().item())
    print(f"configs[3] encoders, low-level model B=256: record max-abs {err:.3e}")
    assert err <= 1e-2
    assert (torch.linalg.norm(h.cpu()[:, rows] - rh) / torch.linalg.norm(rh)).item() <= 1e-2
    eng.close()
