"""CMANet flat baseline on the GPU through the C ABI: parity against the goldens captured from the
imported reference (tests/golden/cma_*.npz) and against the CPU oracle at a larger batch."""
import os

import numpy as np
import pytest
import torch

from oracle import cases, hcm_oracle
from robo_vln_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = {"fp32": 1e-3, "fp16": 1e-2, "bf16": 1e-2}       # BASELINE.json tolerance on the outputs


def _engine(cfg, sd, B, prec):
    from robo_vln_amd.cma import CMAEngine, CMANet
    return CMANet(CMAEngine(cfg, sd, max_batch=B, precision=prec))


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
@pytest.mark.parametrize("name", list(cases.CMA_CASES))
def test_cma_matches_reference_golden(name, prec):
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    cfg, B, T = cases.cma_case_config(name)
    net = _engine(cfg, synth.make_cma_weights(cfg, cases.SEED), B, prec)
    assert net.num_recurrent_layers == cfg.num_recurrent_layers
    hid = torch.zeros(cfg.num_recurrent_layers, B, cfg.hidden, device="cuda")
    for t in range(T):
        obs = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_cma_observations(cfg, B, step=t, seed=cases.SEED).items()}
        out, stop, hid = net((obs, hid, torch.zeros(B, 2), torch.from_numpy(cases.step_masks(B, t))))
        assert "instruction" not in obs
        torch.cuda.synchronize()
        assert np.abs(out.cpu().numpy() - gold["out"][t]).max() <= TOL[prec], (name, t)
        assert np.abs(stop.cpu().numpy() - gold["stop"][t]).max() <= TOL[prec], (name, t)
    h = hid.cpu().numpy()
    rel = np.linalg.norm(h - gold["hidden"]) / max(1e-12, np.linalg.norm(gold["hidden"]))
    assert rel <= (1e-4 if prec == "fp32" else 1e-2), rel


def test_cma_instruction_encoder_tap_fp32():
    """The packed (bi)LSTM instruction encoder: zero beyond each row's length, equal to the oracle elsewhere."""
    name = "cma_128_L20"
    cfg, B, T = cases.cma_case_config(name)
    sd = synth.make_cma_weights(cfg, cases.SEED)
    net = _engine(cfg, sd, B, "fp32")
    net.engine.enable_taps(True)
    obs_np = synth.make_cma_observations(cfg, B, step=0, seed=cases.SEED)
    obs = {k: torch.from_numpy(np.asarray(v)) for k, v in obs_np.items()}
    hid = torch.zeros(cfg.num_recurrent_layers, B, cfg.hidden, device="cuda")
    net((obs, hid, None, torch.zeros(B)))
    torch.cuda.synchronize()
    ins = net.engine.get_tap("cma.instruction")                    # (B, L, C)
    ref, lengths = hcm_oracle.instruction_encoder(torch.from_numpy(obs_np["instruction"]), hcm_oracle.Weights(sd).sub("instruction_encoder."),
                                                  cfg.instr_hidden, cfg.bidirectional)
    ref = ref.permute(0, 2, 1).numpy()                             # (B, Lmax, C)
    lmax = ref.shape[1]
    assert np.abs(ins[:, :lmax] - ref).max() <= 1e-5
    assert (ins[:, lmax:] == 0).all()
    for b in range(B):
        assert (ins[b, int(lengths[b]):] == 0).all()


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_cma_batch16_vs_oracle(prec):
    """Batch 16 at 128x128, L=24, three steps with an episode reset, uint8 RGB frames: HIP path vs the CPU oracle."""
    cfg = cases.CMAConfig(rgb_hw=128, depth_hw=128, instr_len=24).validate()
    B = 16
    sd = synth.make_cma_weights(cfg, 3)
    net = _engine(cfg, sd, B, prec)
    orc = hcm_oracle.CMAOracle(cfg, sd)
    hid = torch.zeros(cfg.num_recurrent_layers, B, cfg.hidden, device="cuda")
    hid_o = torch.zeros(cfg.num_recurrent_layers, B, cfg.hidden)
    for t in range(3):
        obs_np = synth.make_cma_observations(cfg, B, step=t, seed=3, rgb_uint8=True)
        m = cases.step_masks(B, t)
        out, stop, hid = net(({k: torch.from_numpy(np.asarray(v)) for k, v in obs_np.items()}, hid, None, torch.from_numpy(m)))
        o2, s2, hid_o = orc.forward(obs_np, hid_o, m)
        torch.cuda.synchronize()
        assert (out.cpu() - o2).abs().max().item() <= TOL[prec]
        assert (stop.cpu() - s2).abs().max().item() <= TOL[prec]
    rel = (hid.cpu() - hid_o).norm().item() / hid_o.norm().item()
    assert rel <= (1e-4 if prec == "fp32" else 1e-2), rel


def test_cma_rejects_bad_input():
    cfg, B, T = cases.cma_case_config("cma_gru_uni_128_L12")
    sd = synth.make_cma_weights(cfg, cases.SEED)
    from robo_vln_amd.cma import CMAEngine
    bad = dict(sd)
    bad.pop("text_q.bias")
    with pytest.raises(KeyError):
        CMAEngine(cfg, bad, max_batch=2, precision="fp32")
    bad = dict(sd)
    bad["state_q.weight"] = np.zeros((3, 3), dtype=np.float32)
    with pytest.raises(ValueError):
        CMAEngine(cfg, bad, max_batch=2, precision="fp32")
    eng = CMAEngine(cfg, sd, max_batch=2, precision="fp32")
    obs = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_cma_observations(cfg, 2).items()}
    with pytest.raises(ValueError):
        eng.forward(obs, torch.zeros(1, 2, cfg.hidden), torch.zeros(2))       # GRU: R = 2 (two encoders), not 1


def test_cma_hipgraph_replay_equals_eager():
    """Engine graph mode (static I/O buffers, captured hipGraph replay) is bitwise equal to the eager calls."""
    from robo_vln_amd.cma import CMAEngine
    cfg, B, T = cases.cma_case_config("cma_128_L20")
    sd = synth.make_cma_weights(cfg, cases.SEED)
    eager = CMAEngine(cfg, sd, max_batch=B, precision="fp16")
    graph = CMAEngine(cfg, sd, max_batch=B, precision="fp16", graph=True)
    he = torch.zeros(cfg.num_recurrent_layers, B, cfg.hidden, device="cuda")
    hg = he.clone()
    for t in range(5):
        obs = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.make_cma_observations(cfg, B, step=t % 3, seed=cases.SEED).items()}
        m = torch.from_numpy(cases.step_masks(B, t % 3)).cuda()
        oe, se, he = eager.forward(obs, he, m)
        og, sg, hg = graph.forward(obs, hg, m)
        torch.cuda.synchronize()
        assert torch.equal(oe, og) and torch.equal(se, sg) and torch.equal(he, hg)
        hg = hg.clone()
    assert graph.query(7) >= 3        # HCM_GRAPH_LAUNCHES: steps served by graph replay


def test_cma_two_engines_bitwise_deterministic():
    """Two engines built from the same state_dict, the same inputs, several calls: identical bits (guards the one-launch
    instruction-encoder scan and the multi-stream schedule against races)."""
    from robo_vln_amd.cma import CMAEngine
    cfg, B, T = cases.cma_case_config("cma_128_L20")
    sd = synth.make_cma_weights(cfg, cases.SEED)
    e1 = CMAEngine(cfg, sd, max_batch=B, precision="fp16")
    e2 = CMAEngine(cfg, sd, max_batch=B, precision="fp16")
    h = torch.zeros(cfg.num_recurrent_layers, B, cfg.hidden, device="cuda")
    obs = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.make_cma_observations(cfg, B, step=0, seed=cases.SEED).items()}
    m = torch.from_numpy(cases.step_masks(B, 0)).cuda()
    ref = None
    for e in (e1, e1, e2, e1, e2):
        out = [t.clone() for t in e.forward(obs, h, m)]
        torch.cuda.synchronize()
        if ref is None:
            ref = out
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, out))


def test_cma_batch_split_consistency():
    """A batch of 12 environments equals its three sub-batches of 4 run separately (fp32 path): the instruction-encoder scan, which
    owns a couple of samples per workgroup, the grouped trunk launches and the attention kernels do not mix samples."""
    from robo_vln_amd.cma import CMAEngine
    cfg, _, _ = cases.cma_case_config("cma_128_L20")
    B = 12
    sd = synth.make_cma_weights(cfg, cases.SEED)
    eng = CMAEngine(cfg, sd, max_batch=B, precision="fp32")
    obs_np = synth.make_cma_observations(cfg, B, step=1, seed=cases.SEED + 3)
    obs = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in obs_np.items()}
    h = ((torch.rand(cfg.num_recurrent_layers, B, cfg.hidden, generator=torch.Generator().manual_seed(9)) - 0.5) * 0.2).cuda()
    m = torch.ones(B, device="cuda")
    full = [t.clone() for t in eng.forward(obs, h, m)]
    for lo in range(0, B, 4):
        sub_obs = {k: v[lo:lo + 4].contiguous() for k, v in obs.items()}
        part = eng.forward(sub_obs, h[:, lo:lo + 4].contiguous(), m[lo:lo + 4].contiguous())
        torch.cuda.synchronize()
        assert (full[0][lo:lo + 4] - part[0]).abs().max().item() <= 2e-5
        assert (full[1][lo:lo + 4] - part[1]).abs().max().item() <= 2e-5
        assert (full[2][:, lo:lo + 4] - part[2]).abs().max().item() <= 2e-5
    eng.close()
