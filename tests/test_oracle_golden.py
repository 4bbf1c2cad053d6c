"""The CPU oracle (oracle/hcm_oracle.py) replayed against the golden vectors that
oracle/gen_golden.py captured from the imported reference (tests/golden/*.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import cases, hcm_oracle
from robo_vln_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 5e-5   # fp32 CPU restatement vs fp32 CPU reference (different op order only)


@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_matches_reference_golden(name):
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    cfg, B, T, which = cases.case_config(name)
    R = cfg.num_recurrent_layers
    hi = lo = None
    if which in ("both", "hi"):
        hi = hcm_oracle.HighLevelOracle(cfg, synth.materialize(synth.high_level_spec(cfg), "hi", cases.SEED))
    if which in ("both", "lo"):
        lo = hcm_oracle.LowLevelOracle(cfg, synth.materialize(synth.low_level_spec(cfg), "lo", cases.SEED))
    hi_h = torch.zeros(R, B, cfg.hidden)
    lo_h = torch.zeros(R, B, cfg.hidden)
    for t in range(T):
        obs = synth.make_observations(cfg, B, step=t, seed=cases.SEED)
        m = cases.step_masks(B, t)
        th, tl = {}, {}
        if hi is not None:
            logits, hi_h = hi.forward(obs, hi_h, m, th)
            pred = torch.argmax(logits, 1)
        else:
            logits, pred = torch.zeros(B, 4), torch.from_numpy(cases.fixed_subtask(B, t))
        if lo is not None:
            vel, stop, lo_h = lo.forward(obs, lo_h, m, pred, tl)
        else:
            vel, stop = torch.zeros(B, 2), torch.zeros(B, 1)
        rec = torch.cat([logits, vel, stop], 1).numpy()
        assert rec.shape == (B, 7)
        np.testing.assert_allclose(rec, gold["records"][t], atol=TOL, rtol=0)
        if t == 0:
            for pre, taps in (("tap.hi.", th), ("tap.lo.", tl)):
                for k, v in taps.items():
                    if pre + k in gold:
                        np.testing.assert_allclose(cases.subsample(v.numpy()), gold[pre + k], atol=TOL, rtol=0)
    np.testing.assert_allclose(hi_h.numpy(), gold["hi_hidden"], atol=TOL, rtol=0)
    np.testing.assert_allclose(lo_h.numpy(), gold["lo_hidden"], atol=TOL, rtol=0)


@pytest.mark.parametrize("name", list(cases.VARLEN_CASES))
def test_oracle_matches_reference_unpadded_variable_length(name):
    """The reference eval loop's operating point: unpadded (1, L) instructions whose length changes from step to step."""
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    cfg, B, lens = cases.varlen_case_config(name)
    hi_sd = synth.materialize(synth.high_level_spec(cfg), "hi", cases.SEED)
    lo_sd = synth.materialize(synth.low_level_spec(cfg), "lo", cases.SEED)
    ora = hcm_oracle.PolicyOracle(cfg, hi_sd, lo_sd)
    R = cfg.num_recurrent_layers
    hh, lh = torch.zeros(R, B, cfg.hidden), torch.zeros(R, B, cfg.hidden)
    for t, L in enumerate(lens):
        obs = synth.make_observations(cfg, B, step=t, seed=cases.SEED)
        obs["instruction"] = cases.varlen_ids(cfg, L, t)
        rec, hh, lh = ora.act(obs, hh, lh, cases.step_masks(B, t))
        np.testing.assert_allclose(rec.numpy(), gold["records"][t], atol=TOL, rtol=0)
    np.testing.assert_allclose(hh.numpy(), gold["hi_hidden"], atol=TOL, rtol=0)
    np.testing.assert_allclose(lh.numpy(), gold["lo_hidden"], atol=TOL, rtol=0)


def test_padding_changes_the_reference_result_and_ragged_oracle_undoes_it():
    """Why L is a per-call argument: BERT has no attention mask and the poolers average over all L positions, so a zero-padded
    instruction gives a different record than the unpadded one; the ragged form (lengths) is per-environment unpadded."""
    cfg = cases.HCMConfig(rgb_hw=64, depth_hw=64, bert_layers=1, instr_len=12).validate()
    hi_sd, lo_sd = synth.make_weights(cfg, seed=1)
    ora = hcm_oracle.PolicyOracle(cfg, hi_sd, lo_sd)
    B, R = 2, cfg.num_recurrent_layers
    obs = synth.make_observations(cfg, B, step=0, seed=1)
    lens = np.array([7, 12], np.int32)
    ids = obs["instruction"].copy()
    ids[0, lens[0]:] = 0
    obs["instruction"] = ids
    z = torch.zeros(R, B, cfg.hidden)
    m = np.zeros(B, np.float32)
    padded, _, _ = ora.act(obs, z, z, m)
    ragged, _, _ = ora.act(obs, z, z, m, lengths=lens)
    one = {k: v[:1] for k, v in obs.items()}
    one["instruction"] = ids[:1, :7]
    single, _, _ = ora.act(one, z[:, :1], z[:, :1], m[:1])
    assert (padded[0, :4] - single[0, :4]).abs().max().item() > 1e-4          # padding is visible in the high-level logits
    assert (ragged[0, :4] - single[0, :4]).abs().max().item() < 1e-5
    assert (ragged[1] - padded[1]).abs().max().item() < 1e-5                  # the full-length row is unaffected


@pytest.mark.parametrize("name", list(cases.CMA_CASES))
def test_cma_oracle_matches_reference_golden(name):
    """CMANet flat baseline (models/cma.py:211-333): outputs, final hidden state and step-0 intermediates."""
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    cfg, B, T = cases.cma_case_config(name)
    orc = hcm_oracle.CMAOracle(cfg, synth.make_cma_weights(cfg, cases.SEED))
    hid = torch.zeros(cfg.num_recurrent_layers, B, cfg.hidden)
    for t in range(T):
        taps = {}
        out, stop, hid = orc.forward(synth.make_cma_observations(cfg, B, step=t, seed=cases.SEED), hid, cases.step_masks(B, t), taps)
        np.testing.assert_allclose(out.numpy(), gold["out"][t], atol=TOL, rtol=0)
        np.testing.assert_allclose(stop.numpy(), gold["stop"][t], atol=TOL, rtol=0)
        if t == 0:
            for k in ("instruction", "state", "compress", "rnn2_out"):
                np.testing.assert_allclose(cases.subsample(taps[k].numpy()), gold["tap." + k], atol=TOL, rtol=0)
    np.testing.assert_allclose(hid.numpy(), gold["hidden"], atol=TOL, rtol=0)


def test_instruction_encoder_packed_semantics():
    """Packed (bi)LSTM: outputs beyond a row's length are zero, the output is cut to the longest row, and the reverse
    direction of a short row starts at ITS last token (instruction_encoder.py:79-92)."""
    cfg = cases.CMAConfig(rgb_hw=128, depth_hw=128, instr_len=10)
    w = hcm_oracle.Weights(synth.make_cma_weights(cfg, 1)).sub("instruction_encoder.")
    ids = torch.tensor([[5, 6, 7, 0, 0, 0, 0, 0, 0, 0], [9, 8, 7, 6, 5, 4, 0, 0, 0, 0]])
    out, lengths = hcm_oracle.instruction_encoder(ids, w, cfg.instr_hidden, True)
    assert out.shape == (2, 512, 6) and lengths.tolist() == [3, 6]
    assert (out[0, :, 3:] == 0).all() and (out[0, :, :3] != 0).any()
    alone, _ = hcm_oracle.instruction_encoder(ids[:1, :3], w, cfg.instr_hidden, True)
    np.testing.assert_allclose(out[0, :, :3].numpy(), alone[0].numpy(), atol=1e-6)


def test_spatial_embedding_view_quirk():
    """channel c, pixel (y,x) of the pos-emb block reads E.flat[c*16+y*4+x]."""
    E = torch.arange(16 * 64, dtype=torch.float32).view(16, 64)
    x = torch.zeros(1, 3, 4, 4)
    y = hcm_oracle._spatial_cat(x, E)
    assert y.shape == (1, 67, 4, 4)
    for c, yy, xx in ((0, 0, 0), (5, 2, 3), (63, 3, 3)):
        assert y[0, 3 + c, yy, xx].item() == c * 16 + yy * 4 + xx


def test_sinusoid_table():
    pe = hcm_oracle.sinusoid_table(7, 8)
    assert pe.shape == (7, 8)
    assert abs(pe[3, 0].item() - np.sin(3.0)) < 1e-6
    assert abs(pe[3, 1].item() - np.cos(3.0)) < 1e-6
    assert abs(pe[3, 2].item() - np.sin(3.0 / 10000 ** (2 / 8))) < 1e-6


def test_synth_is_deterministic():
    a = synth.uniform01("k", 1000, 3)
    b = synth.uniform01("k", 1000, 3)
    assert (a == b).all() and a.min() >= 0 and a.max() < 1
    assert abs(a.mean() - 0.5) < 0.05
    assert (synth.uniform01("k2", 1000, 3) != a).any()
    # known-answer: first values are a pure function of (key, seed)
    assert a.dtype == np.float32


def test_oracle_seq_forward_matches_reference_golden():
    """Training-path call (T*N frames, (R,N,H) hidden): RNNStateEncoder.seq_forward, golden from the imported reference."""
    name = "seq_T4_N2_gru"
    kw, T, N = cases.SEQ_CASES[name]
    cfg = cases.HCMConfig(**kw).validate()
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    hi = hcm_oracle.HighLevelOracle(cfg, synth.materialize(synth.high_level_spec(cfg), "hi", cases.SEED))
    lo = hcm_oracle.LowLevelOracle(cfg, synth.materialize(synth.low_level_spec(cfg), "lo", cases.SEED))
    obs = cases.seq_observations(cfg, T, N)
    m = cases.seq_masks(T, N)
    h0 = torch.from_numpy(gold["h0"])
    logits, hh = hi.forward(obs, h0.clone(), m)
    vel, stop, lh = lo.forward(obs, h0.clone(), m, torch.from_numpy(cases.fixed_subtask(T * N, 1)))
    for got, key in ((logits, "logits"), (hh, "hi_hidden"), (vel, "vel"), (stop, "stop"), (lh, "lo_hidden")):
        np.testing.assert_allclose(got.numpy(), gold[key], atol=TOL, rtol=0)
    # the scan must differ from treating the rows as independent first steps (state is carried, reset at t=2 for env 1)
    l1, _ = hi.forward({k: v[N:2 * N] for k, v in obs.items()}, torch.zeros_like(h0), np.zeros(N, np.float32))
    assert np.abs(l1.numpy() - gold["logits"][N:2 * N]).max() > 1e-4
