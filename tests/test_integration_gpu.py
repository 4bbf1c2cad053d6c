"""End-to-end on the GPU, the way the reference's eval loop would use the pieces together (hierarchical_trainer.py:1052-1159):
checkpoint file -> engine (checkpoint.py), per-environment observation dicts -> pinned staging (obs.py), the rollout
driver with episode resets (rollout.py) -- against the CPU oracle stepping the same script."""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import hcm_oracle
from robo_vln_amd import synth
from robo_vln_amd.config import HCMConfig

pytestmark = pytest.mark.gpu


def test_checkpoint_stager_rollout_vs_oracle():
    from robo_vln_amd.checkpoint import engine_from_checkpoint, save_checkpoint
    from robo_vln_amd.obs import ObsStager
    from robo_vln_amd.policy import Policy
    from robo_vln_amd.rollout import records_to_actions, rollout
    cfg = HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, vla_layers=2, bert_layers=2).validate()
    B, T = 4, 4
    hi_sd, lo_sd = synth.make_weights(cfg, seed=2)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "ckpt.0.pth")
        save_checkpoint(path, hi_sd, lo_sd, config={"MODEL": {"STATE_ENCODER": {"rnn_type": "LSTM"}}})
        eng = engine_from_checkpoint(path, cfg, max_batch=B, precision="fp32")
    pol = Policy(eng)
    stager = ObsStager(B, cfg.rgb_hw, cfg.depth_hw, cfg.instr_len, device=torch.device("cuda"))
    frames = [synth.make_observations(cfg, B, step=t, seed=2, rgb_uint8=True) for t in range(T)]
    dones = [torch.tensor([False, t == 1, False, t == 2]) for t in range(T)]

    def obs_fn(t, lo, hi):
        per_env = [{k: v[e] for k, v in frames[t].items()} for e in range(lo, hi)]
        return stager.stage(per_env, instruction_changed=(t == 0))

    recs = rollout(pol, obs_fn, lambda t, lo, hi: dones[t][lo:hi], B, T, cfg.num_recurrent_layers, cfg.hidden, torch.device("cuda"))
    torch.cuda.synchronize()
    assert recs.shape == (T, B, 7)
    # oracle stepping the same script
    ora = hcm_oracle.PolicyOracle(cfg, hi_sd, lo_sd)
    R = cfg.num_recurrent_layers
    hh, lh = torch.zeros(R, B, cfg.hidden), torch.zeros(R, B, cfg.hidden)
    mask = np.zeros(B, np.float32)
    for t in range(T):
        ob = dict(frames[t]); ob["rgb"] = ob["rgb"].astype(np.float32); ob["instruction"] = frames[0]["instruction"]
        logits, hh = ora.hi.forward(ob, hh, mask)
        pred = torch.argmax(recs[t, :, :4].cpu(), 1)
        vel, stop, lh = ora.lo.forward(ob, lh, mask, pred)
        ref = torch.cat([logits, vel, stop], 1)
        assert (recs[t].cpu() - ref).abs().max().item() <= 1e-3, t
        mask = (~dones[t]).float().numpy()
    sub, lin, ang, stop_flag = records_to_actions(recs[-1])
    assert sub.shape == (B,) and ang.abs().max().item() <= 1.0 and set(stop_flag.cpu().tolist()) <= {0.0, 1.0}
    eng.close()


def test_c_abi_error_codes():
    """The entry points return status codes instead of crashing: call-order violations, null pointers, wrong handle kind,
    out-of-range batch (hcm.h: hcm_status)."""
    import ctypes as C
    from robo_vln_amd import _lib
    from robo_vln_amd.policy import _to_struct
    lib = _lib.lib()
    cfg = HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=1).validate()
    st = _to_struct(cfg, 2, "fp16", True, True)
    h = C.c_void_p()
    assert lib.hcm_create(C.byref(st), C.byref(h)) == 0
    d = torch.zeros(64, device="cuda")
    p = C.c_void_p(d.data_ptr())
    # forward before finalize -> HCM_ERR_STATE (-2)
    assert lib.hcm_act(h, p, _lib.HCM_F32, p, p, _lib.HCM_I64, None, 1, 20, p, p, p, p, p, p, None) == -2
    assert b"finalize" in lib.hcm_last_error(h)
    # finalize with missing tensors -> HCM_ERR_KEY (-3), message names a key
    assert lib.hcm_finalize(h) == -3 and b"Missing key" in lib.hcm_last_error(h)
    lib.hcm_destroy(h)
    # a finalized engine: null pointer / bad batch / wrong handle kind
    hi_sd, lo_sd = synth.make_weights(cfg, seed=0)
    from robo_vln_amd.policy import HCMEngine
    eng = HCMEngine(cfg, hi_sd, lo_sd, max_batch=2, precision="fp16")
    hh = eng._h
    assert lib.hcm_act(hh, None, _lib.HCM_F32, p, p, _lib.HCM_I64, None, 1, 20, p, p, p, p, p, p, None) == -1          # null rgb
    assert lib.hcm_act(hh, p, _lib.HCM_F32, p, p, _lib.HCM_I64, None, 3, 20, p, p, p, p, p, p, None) == -1             # B > max_batch
    assert lib.hcm_act(hh, p, _lib.HCM_BF16, p, p, _lib.HCM_I64, None, 1, 20, p, p, p, p, p, p, None) == -1            # bad rgb dtype
    assert lib.hcm_act(hh, p, _lib.HCM_F32, p, p, _lib.HCM_I64, None, 1, 21, p, p, p, p, p, p, None) == -1             # L > max L of the engine
    assert b"instruction length 21" in lib.hcm_last_error(hh)
    assert lib.hcm_act(hh, p, _lib.HCM_F32, p, p, _lib.HCM_I64, None, 1, 0, p, p, p, p, p, p, None) == -1              # L < 1
    assert lib.hcm_cma_forward(hh, p, _lib.HCM_F32, p, p, _lib.HCM_I64, 1, 20, p, p, p, p, p, None) == -2        # not a CMANet handle
    out = C.c_int64()
    assert lib.hcm_query(hh, 99, C.byref(out)) == -1
    assert lib.hcm_query(hh, _lib.HCM_RECORD_WIDTH, C.byref(out)) == 0 and out.value == 7
    eng.close()


def test_rollout_with_cached_instructions_equals_recomputing():
    """rollout(cache_instruction=True): BERT only for the environments whose episode just ended -- same records, bit for bit."""
    from robo_vln_amd.policy import HCMEngine, Policy
    from robo_vln_amd.rollout import rollout
    cfg = HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, vla_layers=1, bert_layers=2).validate()
    n, T = 6, 5
    hi_sd, lo_sd = synth.make_weights(cfg, seed=8)
    eng = HCMEngine(cfg, hi_sd, lo_sd, max_batch=n, precision="fp16")
    pol = Policy(eng)
    frames = [synth.make_observations(cfg, n, step=t, seed=8, rgb_uint8=True) for t in range(T)]
    dones = [torch.tensor([t == 1 and e in (1, 4) or t == 3 and e == 0 for e in range(n)]) for t in range(T)]
    # an environment whose episode ended gets a new instruction from the next step on
    instr = [frames[0]["instruction"].copy()]
    for t in range(1, T):
        cur = instr[-1].copy()
        for e in torch.nonzero(dones[t - 1]).flatten().tolist():
            cur[e] = frames[t]["instruction"][(e + 1) % n]
        instr.append(cur)

    def obs_fn(t, lo, hi):
        return {"rgb": torch.from_numpy(frames[t]["rgb"][lo:hi]).cuda(), "depth": torch.from_numpy(frames[t]["depth"][lo:hi]).cuda(),
                "instruction": torch.from_numpy(instr[t][lo:hi]).cuda()}
    args = (pol, obs_fn, lambda t, lo, hi: dones[t][lo:hi], n, T, cfg.num_recurrent_layers, cfg.hidden, torch.device("cuda"))
    a = rollout(*args)
    b = rollout(*args, cache_instruction=True)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    eng.close()


@pytest.mark.parametrize("B", [1, 3])
@pytest.mark.parametrize("reuse", [False, True])
def test_chain_graphs_equal_the_forked_graph(B, reuse):
    """HCM_ACT_CHAIN_GRAPHS: the step replayed as one linear hipGraph per encoder chain (stitched by events outside the graphs) gives the bits of the
    single graph captured across the forked streams and of eager launches -- over several steps whose frames change in place (so the replays read new
    data and the recurrent state threads through), also with the instruction stream cached (the BERT chain's graph is then empty), and at the batch
    of one where chain_graphs="auto" switches it on."""
    from robo_vln_amd import _lib
    from robo_vln_amd.policy import HCMEngine
    cfg = HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=2).validate()
    T = 6
    hi_sd, lo_sd = synth.make_weights(cfg, seed=7)
    engs = [HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="fp16", graph=True, chain_graphs=True),
            HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="fp16", graph=True, chain_graphs=False),
            HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="fp16", graph=False),
            HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="fp16", graph=True)]          # "auto"
    frames = [synth.make_observations(cfg, B, step=t, seed=7) for t in range(T)]
    obs = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in frames[0].items()}
    R = cfg.num_recurrent_layers
    state = [(torch.zeros(R, B, cfg.hidden, device="cuda"), torch.zeros(R, B, cfg.hidden, device="cuda")) for _ in engs]
    m = torch.ones(B, device="cuda")
    for t in range(T):
        obs["rgb"].copy_(torch.from_numpy(frames[t]["rgb"]).cuda())
        obs["depth"].copy_(torch.from_numpy(frames[t]["depth"]).cuda())
        outs = []
        for i, e in enumerate(engs):
            r, hh, lh = e.act(obs, state[i][0], state[i][1], m, reuse_instruction=reuse and t > 0)
            r, hh, lh = r.clone(), hh.clone(), lh.clone()
            state[i] = (hh, lh)
            outs.append((r, hh, lh))
        torch.cuda.synchronize()
        for o in outs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(outs[0], o)), t
        assert torch.isfinite(outs[0][0]).all()
    assert engs[0].query(_lib.HCM_GRAPH_LAUNCHES) >= 2 and engs[1].query(_lib.HCM_GRAPH_LAUNCHES) >= 2
    for e in engs:
        e.close()


@pytest.mark.parametrize("depth_hw", [128, 192])
def test_three_chain_step_is_deterministic(depth_hw):
    """Race screen (round 4): every replay form runs every step TWICE from the same inputs and must agree with itself and with the others, 300 steps of
    changing frames at frame sizes whose depth trunk takes the generic bottleneck path.  GroupNorm on load once let a block's first conv write its
    output over the identity rows the same launch was still reading (a slot counted as free while a pending record needed it): 0.3-1 % of the steps
    differed from run to run -- only with the other chains competing for the chip, which no single-kernel test sees (tools/step_determinism.py)."""
    from robo_vln_amd.policy import HCMEngine
    cfg = HCMConfig(rgb_hw=128, depth_hw=depth_hw, instr_len=20, bert_layers=2).validate()
    B, T = 3, 300
    hi_sd, lo_sd = synth.make_weights(cfg, seed=7)
    engs = [HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="fp16", graph=True, chain_graphs=True),
            HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="fp16", graph=True, chain_graphs=False),
            HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="fp16", graph=False)]
    frames = [synth.make_observations(cfg, B, step=t, seed=7) for t in range(6)]
    obs = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in frames[0].items()}
    R = cfg.num_recurrent_layers
    hh = torch.zeros(R, B, cfg.hidden, device="cuda"); lh = torch.zeros(R, B, cfg.hidden, device="cuda")
    m = torch.ones(B, device="cuda")
    bad = 0
    for t in range(T):
        obs["rgb"].copy_(torch.from_numpy(frames[t % 6]["rgb"]).cuda())
        obs["depth"].copy_(torch.from_numpy(frames[t % 6]["depth"]).cuda())
        torch.cuda.synchronize()
        runs = []
        for e in engs:
            for _ in range(2):
                r, h2, l2 = e.act(obs, hh, lh, m)
                runs.append((r.clone(), h2.clone(), l2.clone()))
                torch.cuda.synchronize()
        bad += sum(not all(torch.equal(a, b) for a, b in zip(runs[0], o)) for o in runs[1:])
        hh, lh = runs[0][1], runs[0][2]
    for e in engs:
        e.close()
    assert bad == 0, f"{bad} of {5 * T} repeated steps differed"


@pytest.mark.parametrize("graph", [False, True, "chain"])
@pytest.mark.parametrize("rgb_uint8", [True, False])
def test_host_frames_equal_device_frames(graph, rgb_uint8):
    """HCM_ACT_HOST_FRAMES: pinned host frames handed to the library (one host->device copy per encoder chain, inside the captured step)
    give the same bits as frames the caller copied to the device first -- over several steps whose frames change IN the same pinned
    buffers (a graph replay must read the new contents), eager and hipGraph, uint8 and float RGB."""
    from robo_vln_amd.obs import ObsStager
    from robo_vln_amd.policy import HCMEngine
    cfg = HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=2).validate()
    B, T = 3, 6
    hi_sd, lo_sd = synth.make_weights(cfg, seed=5)
    # ("chain": per-chain linear graphs, where the replay enqueues the copies itself, outside the graphs, at the head of the chains' streams)
    eng_d = HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="fp16", graph=bool(graph), chain_graphs=graph == "chain")
    eng_h = HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="fp16", graph=bool(graph), chain_graphs=graph == "chain")
    frames = [synth.make_observations(cfg, B, step=t, seed=5, rgb_uint8=rgb_uint8) for t in range(T)]
    host = {"rgb": torch.empty(B, 128, 128, 3, dtype=torch.uint8 if rgb_uint8 else torch.float32).pin_memory(),
            "depth": torch.empty(B, 128, 128, 1).pin_memory()}
    ids = torch.from_numpy(frames[0]["instruction"]).cuda()
    R = cfg.num_recurrent_layers
    hh_d = lh_d = hh_h = lh_h = torch.zeros(R, B, cfg.hidden, device="cuda")
    m = torch.ones(B, device="cuda")
    for t in range(T):
        dev = {"rgb": torch.from_numpy(frames[t]["rgb"]).cuda(), "depth": torch.from_numpy(frames[t]["depth"]).cuda(), "instruction": ids}
        rd, hh_d, lh_d = eng_d.act(dev, hh_d, lh_d, m)
        rd, hh_d, lh_d = rd.clone(), hh_d.clone(), lh_d.clone()
        torch.cuda.synchronize()                                   # the previous step has consumed the pinned buffers
        host["rgb"].copy_(torch.from_numpy(frames[t]["rgb"]))
        host["depth"].copy_(torch.from_numpy(frames[t]["depth"]))
        rh, hh_h, lh_h = eng_h.act({"rgb": host["rgb"], "depth": host["depth"], "instruction": ids}, hh_h, lh_h, m, host_frames=True)
        rh, hh_h, lh_h = rh.clone(), hh_h.clone(), lh_h.clone()
        torch.cuda.synchronize()
        assert torch.equal(rd, rh) and torch.equal(hh_d, hh_h) and torch.equal(lh_d, lh_h), t
    if graph:
        from robo_vln_amd import _lib
        assert eng_h.query(_lib.HCM_GRAPH_LAUNCHES) >= 2           # (the copies are nodes of the captured graph)
    with pytest.raises(ValueError):
        eng_h.act({"rgb": torch.zeros(B, 128, 128, 3, dtype=torch.uint8), "depth": host["depth"], "instruction": ids}, hh_h, lh_h, m, host_frames=True)
    eng_d.close(); eng_h.close()


def test_rollout_raises_on_poisoned_frames_and_recovers():
    """rollout() answers for its own steps: a NaN pixel in one environment's depth frame makes it raise FloatingPointError (the
    engine's overflow guard), the next clean rollout on the same engine passes."""
    from robo_vln_amd.policy import HCMEngine, Policy
    from robo_vln_amd.rollout import rollout
    cfg = HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=1).validate()
    B, T = 2, 3
    hi_sd, lo_sd = synth.make_weights(cfg, seed=4)
    pol = Policy(HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="fp16"))
    frames = [{k: torch.from_numpy(v).cuda() for k, v in synth.make_observations(cfg, B, step=t, seed=4).items()} for t in range(T)]
    dev = torch.device("cuda")
    never = lambda t, lo, hi: torch.zeros(hi - lo, dtype=torch.bool)
    recs = rollout(pol, lambda t, lo, hi: frames[t], never, B, T, cfg.num_recurrent_layers, cfg.hidden, dev)
    assert torch.isfinite(recs).all()
    bad = [dict(f) for f in frames]
    bad[1]["depth"] = frames[1]["depth"].clone()
    bad[1]["depth"][0, 5, 7, 0] = float("nan")
    with pytest.raises(FloatingPointError):
        rollout(pol, lambda t, lo, hi: bad[t], never, B, T, cfg.num_recurrent_layers, cfg.hidden, dev)
    recs2 = rollout(pol, lambda t, lo, hi: frames[t], never, B, T, cfg.num_recurrent_layers, cfg.hidden, dev)
    assert torch.equal(recs, recs2)
    pol.engine.close()


@pytest.mark.parametrize("graph", [False, True])
def test_library_all_gather_world_one(graph):
    """hcm_comm_unique_id / hcm_comm_init / hcm_act_gather on a world of ONE rank (the box has one GPU): the library creates its own RCCL
    communicator from an id passed through a torch.distributed group (gloo here), and act(gather=True) -- the step plus ONE ncclAllGather
    enqueued by the library on the step's stream -- returns the gathered record, which for one rank is the step's own record, bit for bit;
    the raw C entry points answer a missing communicator with HCM_ERR_STATE."""
    import torch.distributed as dist
    from robo_vln_amd import _lib
    from robo_vln_amd.policy import HCMEngine
    cfg = HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=1).validate()
    B = 2
    hi_sd, lo_sd = synth.make_weights(cfg, seed=4)
    eng = HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="fp16", graph=graph)
    obs = {k: torch.from_numpy(v).cuda() for k, v in synth.make_observations(cfg, B, step=0, seed=4).items()}
    R = cfg.num_recurrent_layers
    z = torch.zeros(R, B, cfg.hidden, device="cuda")
    m = torch.zeros(B, device="cuda")
    with pytest.raises(RuntimeError):
        eng.act(obs, z, z, m, gather=True)                       # no communicator yet
    lib = _lib.lib()
    buf = torch.empty(B, 7, device="cuda")
    rc = lib.hcm_act_gather(eng._h, obs["rgb"].data_ptr(), _lib.HCM_F32, obs["depth"].data_ptr(), obs["instruction"].data_ptr(), _lib.HCM_I64, None, B, 20,
                            z.data_ptr(), z.data_ptr(), m.data_ptr(), buf.data_ptr(), z.clone().data_ptr(), z.clone().data_ptr(), 0, buf.data_ptr(), None)
    assert rc == -2                                              # HCM_ERR_STATE
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29567", rank=0, world_size=1)
    try:
        assert eng.comm_init() == 1
        ref = [eng.act(obs, z, z, m)[0].clone() for _ in range(3)][-1]
        for _ in range(4):                                       # eager, capture, replay
            got, hh, lh = eng.act(obs, z, z, m, gather=True)
        torch.cuda.synchronize()
        assert got.shape == (B, 7) and torch.equal(got, ref)
        with pytest.raises(RuntimeError):
            eng.comm_init()                                      # one communicator per handle
        # a rank whose own step fails still takes part in the step's collective, with an all-NaN record, and reports its error afterwards
        # (a rank that simply returned would leave its peers blocked in ncclAllGather: the private communicator has no watchdog)
        gat = torch.zeros(B, 7, device="cuda")
        loc = torch.zeros(B, 7, device="cuda")
        rc = lib.hcm_act_gather(eng._h, obs["rgb"].data_ptr(), _lib.HCM_F32, obs["depth"].data_ptr(), obs["instruction"].data_ptr(), _lib.HCM_I64, None, B, 513,
                                z.data_ptr(), z.data_ptr(), m.data_ptr(), loc.data_ptr(), z.clone().data_ptr(), z.clone().data_ptr(), 0, gat.data_ptr(),
                                None)                            # L = 513 > BERT's position table: the step is refused
        torch.cuda.synchronize()
        assert rc != 0 and b"NaN record" in lib.hcm_last_error(eng._h)
        assert torch.isnan(gat).all() and torch.isnan(loc).all()
        # env-sharded ranks: an overflow-guard alarm inside act(gather=True) is only recorded; guard_check() raises it (on every rank at once)
        eng._guard_every = 1
        bad = dict(obs)
        bad["depth"] = obs["depth"].clone()
        bad["depth"][1, 5, 7, 0] = float("nan")
        for _ in range(4):
            eng.act(bad, z, z, m, gather=True)                   # must not raise
            torch.cuda.synchronize()
        assert eng.guard_alarm > 0
        with pytest.raises(FloatingPointError):
            eng.guard_check()
        assert eng.guard_alarm == 0
        eng.comm_abort()                                         # ncclCommAbort; the handle may create a new communicator afterwards
        assert eng.comm_world == 0
        with pytest.raises(RuntimeError):
            eng.act(obs, z, z, m, gather=True)
    finally:
        if created:
            dist.destroy_process_group()
    eng.close()
