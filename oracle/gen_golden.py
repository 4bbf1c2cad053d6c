"""Generate tests/golden/*.npz by running the REAL reference models (imported from /root/reference
through oracle/ref_shims.py) on the deterministic synthetic weights/inputs of robo-vln_amd/synth.py.

Run in the build container only:   python oracle/gen_golden.py [case ...]
The goldens hold OUTPUTS (+ subsampled intermediates); inputs and weights are regenerated from
the seed by whoever replays a case.  Also prints the max-abs difference between the reference and
the oracle restatement (oracle/hcm_oracle.py) for each stored tensor.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcm_pkg  # noqa: E402

hcm_pkg.load()
from robo_vln_amd import synth  # noqa: E402
from oracle import cases, ref_shims, hcm_oracle  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _hook(store, name, pick=None):
    def fn(mod, inp, out):
        t = out if pick is None else pick(out)
        store.setdefault(name, []).append(t.detach().clone())
    return fn


def run_case(name):
    cfg, B, T, which = cases.case_config(name)
    torch.manual_seed(0)
    hi_sd = lo_sd = None
    if which in ("both", "hi"):
        hi_sd = synth.materialize(synth.high_level_spec(cfg), "hi", cases.SEED)
    if which in ("both", "lo"):
        lo_sd = synth.materialize(synth.low_level_spec(cfg), "lo", cases.SEED)
    hi, lo = ref_shims.build_models(cfg, hi_sd, lo_sd, want_hi=hi_sd is not None, want_lo=lo_sd is not None)
    taps = {}
    if hi is not None:
        hi.depth_encoder.register_forward_hook(_hook(taps, "hi.depth_spatial"))
        hi.rgb_encoder.register_forward_hook(_hook(taps, "hi.rgb_spatial"))
        hi.embedding_layer.register_forward_hook(_hook(taps, "hi.bert", lambda o: o[0]))
        hi.image_cm_encoder.register_forward_hook(_hook(taps, "hi.vla"))
        hi.state_encoder.register_forward_hook(_hook(taps, "hi.rnn_out", lambda o: o[0]))
        hi.state_encoder.register_forward_pre_hook(lambda m, i: taps.setdefault("hi.rnn_in", []).append(i[0].detach().clone()))
    if lo is not None:
        lo.depth_encoder.register_forward_hook(_hook(taps, "lo.depth_flat"))
        lo.rgb_encoder.register_forward_hook(_hook(taps, "lo.rgb_flat"))
        lo.state_encoder.register_forward_pre_hook(lambda m, i: taps.setdefault("lo.rnn_in", []).append(i[0].detach().clone()))

    R = cfg.num_recurrent_layers
    hi_h = torch.zeros(R, B, cfg.hidden)
    lo_h = torch.zeros(R, B, cfg.hidden)
    prev = torch.zeros(B, 2, dtype=torch.long)
    records = []
    t0 = time.time()
    with torch.no_grad():
        for t in range(T):
            obs_np = synth.make_observations(cfg, B, step=t, seed=cases.SEED)
            # batch_obs contract: every sensor float32 (common/utils.py:78-83)
            obs = {k: torch.from_numpy(v.astype(np.float32)) for k, v in obs_np.items()}
            masks = ref_shims.ref_masks(cases.step_masks(B, t))
            rec = []
            if hi is not None:
                logits, hi_h = hi((dict(obs), hi_h, prev, masks))     # copy: forward deletes 'instruction'
                rec.append(logits)
                pred = torch.argmax(logits, dim=1)
            else:
                rec.append(torch.zeros(B, 4))
                pred = torch.from_numpy(cases.fixed_subtask(B, t))
            if lo is not None:
                vel, stop, lo_h = lo((dict(obs), lo_h, prev, masks, pred))
                rec += [vel, stop]
            else:
                rec += [torch.zeros(B, 2), torch.zeros(B, 1)]
            records.append(torch.cat(rec, dim=1))
    dt = time.time() - t0
    out = {"records": torch.stack(records).numpy(), "hi_hidden": hi_h.numpy(), "lo_hidden": lo_h.numpy()}
    # the hooks sit on the encoder modules, i.e. BEFORE the `* 0` of an ablated encoder: those tensors reach no output, drop them
    dropped = (("hi.depth_spatial", "lo.depth_flat") if cfg.ablate_depth else ()) + (("hi.rgb_spatial", "lo.rgb_flat") if cfg.ablate_rgb else ())
    for k, v in taps.items():
        if k in dropped:
            continue
        if k == "hi.vla":
            out["tap.hi.vla_rgb"] = cases.subsample(v[0].numpy())
            out["tap.hi.vla_depth"] = cases.subsample(v[1].numpy())
        else:
            out["tap." + k] = cases.subsample(v[0].numpy())
    meta = dict(case=name, config=repr(cfg.to_dict()), batch=B, steps=T, which=which, seed=cases.SEED,
                torch=torch.__version__,
                note="reference modules imported from /root/reference via oracle/ref_shims.py; torchvision resnet50 "
                     "shim uses AdaptiveAvgPool2d(1) (SURVEY 8a-a3); BERT = transformers.BertModel(BertConfig) random init; "
                     "taps are step-0 tensors flattened and subsampled by cases.subsample")
    out["meta"] = np.array(repr(meta))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)

    # cross-check the oracle restatement right here
    worst = compare_oracle(name, out, hi_sd, lo_sd)
    print(f"[{name}] reference ran {T} steps B={B} in {dt:.1f}s; oracle-vs-reference worst max-abs {worst:.3e}")
    return worst


def compare_oracle(name, gold, hi_sd, lo_sd):
    cfg, B, T, which = cases.case_config(name)
    R = cfg.num_recurrent_layers
    hi_o = hcm_oracle.HighLevelOracle(cfg, hi_sd) if hi_sd is not None else None
    lo_o = hcm_oracle.LowLevelOracle(cfg, lo_sd) if lo_sd is not None else None
    hi_h = torch.zeros(R, B, cfg.hidden)
    lo_h = torch.zeros(R, B, cfg.hidden)
    worst = 0.0
    for t in range(T):
        obs = synth.make_observations(cfg, B, step=t, seed=cases.SEED)
        m = cases.step_masks(B, t)
        taps_hi, taps_lo = {}, {}
        if hi_o is not None:
            logits, hi_h = hi_o.forward(obs, hi_h, m, taps_hi if t == 0 else None)
            pred = torch.argmax(logits, 1)
        else:
            logits = torch.zeros(B, 4)
            pred = torch.from_numpy(cases.fixed_subtask(B, t))
        if lo_o is not None:
            vel, stop, lo_h = lo_o.forward(obs, lo_h, m, pred, taps_lo if t == 0 else None)
        else:
            vel, stop = torch.zeros(B, 2), torch.zeros(B, 1)
        rec = torch.cat([logits, vel, stop], 1).numpy()
        d = np.abs(rec - gold["records"][t]).max()
        worst = max(worst, d)
        if t == 0:
            for pre, k, v in [("tap.hi.", k, v) for k, v in taps_hi.items()] + [("tap.lo.", k, v) for k, v in taps_lo.items()]:
                if True:
                    if pre + k in gold:
                        g = gold[pre + k]
                        dd = np.abs(cases.subsample(v.numpy()) - g).max()
                        print(f"    {pre + k:24s} max-abs {dd:.3e}  (|gold| max {np.abs(g).max():.3f} std {g.std():.3f})")
                        worst = max(worst, dd)
        print(f"    step {t} record max-abs {d:.3e}   record={np.array2string(rec[0], precision=4)}")
    return worst


def run_varlen_case(name):
    """One reference model pair stepped through instructions of different, unpadded lengths ((1, L) ids), state carried."""
    cfg, B, lens = cases.varlen_case_config(name)
    hi_sd = synth.materialize(synth.high_level_spec(cfg), "hi", cases.SEED)
    lo_sd = synth.materialize(synth.low_level_spec(cfg), "lo", cases.SEED)
    hi, lo = ref_shims.build_models(cfg, hi_sd, lo_sd)
    ora = hcm_oracle.PolicyOracle(cfg, hi_sd, lo_sd)
    R = cfg.num_recurrent_layers
    hi_h = torch.zeros(R, B, cfg.hidden); lo_h = torch.zeros(R, B, cfg.hidden)
    o_hh = torch.zeros(R, B, cfg.hidden); o_lh = torch.zeros(R, B, cfg.hidden)
    prev = torch.zeros(B, 2, dtype=torch.long)
    records, worst = [], 0.0
    with torch.no_grad():
        for t, L in enumerate(lens):
            obs_np = synth.make_observations(cfg, B, step=t, seed=cases.SEED)
            obs_np["instruction"] = cases.varlen_ids(cfg, L, t)
            obs = {k: torch.from_numpy(v.astype(np.float32)) for k, v in obs_np.items()}
            m = cases.step_masks(B, t)
            masks = ref_shims.ref_masks(m)
            logits, hi_h = hi((dict(obs), hi_h, prev, masks))
            vel, stop, lo_h = lo((dict(obs), lo_h, prev, masks, torch.argmax(logits, dim=1)))
            records.append(torch.cat([logits, vel, stop], dim=1))
            rec_o, o_hh, o_lh = ora.act(obs_np, o_hh, o_lh, m)
            worst = max(worst, (rec_o - records[-1]).abs().max().item())
    out = {"records": torch.stack(records).numpy(), "hi_hidden": hi_h.numpy(), "lo_hidden": lo_h.numpy(),
           "meta": np.array(repr(dict(case=name, config=repr(cfg.to_dict()), batch=B, lengths=lens, seed=cases.SEED, torch=torch.__version__,
                                      note="reference hi/lo models stepped with UNPADDED (1, L) instructions of these lengths, one step each")))}
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    worst = max(worst, (o_hh - hi_h).abs().max().item(), (o_lh - lo_h).abs().max().item())
    print(f"[{name}] unpadded lengths {lens} B={B}: oracle-vs-reference worst max-abs {worst:.3e}")
    return worst


def run_seq_case(name):
    """Reference models called on T*N frames with an (R,N,H) hidden state: RNNStateEncoder.seq_forward path."""
    kw, T, N = cases.SEQ_CASES[name]
    cfg = cases.HCMConfig(**kw).validate()
    hi_sd = synth.materialize(synth.high_level_spec(cfg), "hi", cases.SEED)
    lo_sd = synth.materialize(synth.low_level_spec(cfg), "lo", cases.SEED)
    hi, lo = ref_shims.build_models(cfg, hi_sd, lo_sd)
    obs_np = cases.seq_observations(cfg, T, N)
    obs = {k: torch.from_numpy(v.astype(np.float32)) for k, v in obs_np.items()}
    m = cases.seq_masks(T, N)
    masks = torch.from_numpy(m).view(-1, 1).expand(-1, 2).contiguous()
    R = cfg.num_recurrent_layers
    g = torch.Generator().manual_seed(3)
    h0 = (torch.rand(R, N, cfg.hidden, generator=g) - 0.5)
    st = torch.from_numpy(cases.fixed_subtask(T * N, 1))
    prev = torch.zeros(T * N, 2, dtype=torch.long)
    with torch.no_grad():
        logits, hi_h = hi((dict(obs), h0.clone(), prev, masks))
        vel, stop, lo_h = lo((dict(obs), h0.clone(), prev, masks, st))
    out = {"logits": logits.numpy(), "vel": vel.numpy(), "stop": stop.numpy(), "hi_hidden": hi_h.numpy(), "lo_hidden": lo_h.numpy(),
           "h0": h0.numpy(),
           "meta": np.array(repr(dict(case=name, T=T, N=N, config=repr(cfg.to_dict()),
                                      note="reference hi/lo forward with T*N frames and an (R,N,H) hidden state -> seq_forward")))}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    # oracle cross-check
    ho = hcm_oracle.HighLevelOracle(cfg, hi_sd)
    lo_o = hcm_oracle.LowLevelOracle(cfg, lo_sd)
    l2, h2 = ho.forward(obs_np, h0.clone(), m)
    v2, s2, lh2 = lo_o.forward(obs_np, h0.clone(), m, st)
    worst = max(np.abs(l2.numpy() - out["logits"]).max(), np.abs(h2.numpy() - out["hi_hidden"]).max(),
                np.abs(v2.numpy() - out["vel"]).max(), np.abs(s2.numpy() - out["stop"]).max(), np.abs(lh2.numpy() - out["lo_hidden"]).max())
    print(f"[{name}] seq_forward T={T} N={N}: oracle-vs-reference worst max-abs {worst:.3e}")
    return worst


def run_cma_case(name):
    """Reference CMANet (models/cma.py) stepped T times with the scripted episode-reset masks."""
    cfg, B, T = cases.cma_case_config(name)
    sd = synth.make_cma_weights(cfg, cases.SEED)
    net = ref_shims.build_cma(cfg, sd)
    taps = {}
    net.instruction_encoder.register_forward_hook(_hook(taps, "instruction"))
    net.state_encoder.register_forward_hook(_hook(taps, "state", lambda o: o[0]))
    net.second_state_compress.register_forward_hook(_hook(taps, "compress"))
    net.second_state_encoder.register_forward_hook(_hook(taps, "rnn2_out", lambda o: o[0]))
    R = cfg.num_recurrent_layers
    hid = torch.zeros(R, B, cfg.hidden)
    orc = hcm_oracle.CMAOracle(cfg, sd)
    hid_o = torch.zeros(R, B, cfg.hidden)
    outs, stops, worst = [], [], 0.0
    for t in range(T):
        obs_np = synth.make_cma_observations(cfg, B, step=t, seed=cases.SEED)
        obs = {k: torch.from_numpy(np.asarray(v).astype(np.float32)) for k, v in obs_np.items()}
        m = cases.step_masks(B, t)
        with torch.no_grad():
            out, stop, hid = net((obs, hid.clone(), torch.zeros(B, 2), ref_shims.ref_masks(m)))
        assert "instruction" not in obs                      # cma.py:228 deletes the key
        outs.append(out.numpy()); stops.append(stop.numpy())
        o2, s2, hid_o = orc.forward(obs_np, hid_o, m)
        worst = max(worst, np.abs(o2.numpy() - outs[-1]).max(), np.abs(s2.numpy() - stops[-1]).max(), np.abs(hid_o.numpy() - hid.numpy()).max())
    gold = {"out": np.stack(outs), "stop": np.stack(stops), "hidden": hid.numpy(),
            "meta": np.array(repr(dict(case=name, B=B, T=T, config=repr(cfg.to_dict()),
                                       note="reference CMANet.forward, masks (B,2,1) workaround, taps from step 0")))}
    for k, v in taps.items():
        gold["tap." + k] = cases.subsample(v[0].numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **gold)
    print(f"[{name}] CMANet T={T} B={B}: oracle-vs-reference worst max-abs {worst:.3e}")
    return worst


if __name__ == "__main__":
    names = sys.argv[1:] or (list(cases.CASES) + list(cases.VARLEN_CASES) + list(cases.SEQ_CASES) + list(cases.CMA_CASES))
    bad = 0
    for n in names:
        w = (run_cma_case(n) if n in cases.CMA_CASES else run_seq_case(n) if n in cases.SEQ_CASES
             else run_varlen_case(n) if n in cases.VARLEN_CASES else run_case(n))
        bad |= (w > 1e-4)
    sys.exit(1 if bad else 0)
