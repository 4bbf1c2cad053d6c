"""The rollout glue and the N>1 (environment-sharded) path on CPU: world_size-2 gloo processes, each running
the ORACLE policy as a stand-in for the GPU engine (the sharding/gather logic is device-independent), must
reproduce the single-process rollout exactly."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _OraclePolicy:
    def __init__(self, cfg, hi_sd, lo_sd):
        from oracle import hcm_oracle
        self.o = hcm_oracle.PolicyOracle(cfg, hi_sd, lo_sd)

        self.engine = self          # rollout(cache_instruction=True) talks to policy.engine
        self.refreshed = []

    def act(self, obs, hh, lh, prev, masks, reuse_instruction=False):
        # the oracle recomputes the instruction stream every step, which is what the cached path must be equal to
        return self.o.act(obs, hh, lh, masks.numpy())

    def refresh_instruction(self, instruction, env_indices):
        self.refreshed.append(list(map(int, env_indices)))


def _setup(G=4, vocab=30522):
    import hcm_pkg
    hcm_pkg.load()
    from robo_vln_amd import synth
    from robo_vln_amd.config import HCMConfig
    cfg = HCMConfig(rgb_hw=64, depth_hw=64, instr_len=8, bert_layers=1, bert_vocab=vocab)
    hi_sd, lo_sd = synth.make_weights(cfg, seed=1)
    allobs = [synth.make_observations(cfg, G, step=t, seed=1) for t in range(3)]

    def obs_fn(t, lo, hi):
        return {k: v[lo:hi] for k, v in allobs[t].items()}

    def done_fn(t, lo, hi):
        # uneven pattern: env 2 ends after step 0; with 16 envs also envs 5, 6 (one rank loses both its envs) and 15 after step 1
        d = torch.zeros(G, dtype=torch.bool)
        if t == 0:
            d[2] = True
        if t == 1 and G >= 16:
            d[5] = d[6] = d[15] = True
        return d[lo:hi]
    return cfg, hi_sd, lo_sd, obs_fn, done_fn, G


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception as e:      # surface the failure to the parent instead of a queue timeout
        import traceback
        q.put("worker %d failed: %s\n%s" % (rank, e, traceback.format_exc()))
        raise


def _worker_body(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import hcm_pkg
    hcm_pkg.load()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2 if world <= 2 else 1)
    from robo_vln_amd.rollout import rollout
    big = world > 2
    cfg, hi_sd, lo_sd, obs_fn, done_fn, G = _setup(16, 2048) if big else _setup()
    pol = _OraclePolicy(cfg, hi_sd, lo_sd)
    rec = rollout(pol, obs_fn, done_fn, G // world, 3, cfg.num_recurrent_layers, cfg.hidden, "cpu", world, rank, cache_instruction=big)
    if big:
        # every rank holds ALL ranks' records after the gather (not only rank 0), and refreshed exactly its own finished environments
        allr = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(allr, rec)
        assert all(torch.equal(a, rec) for a in allr)
        lo = rank * (G // world)
        want = [[e - lo for e in ends if lo <= e < lo + G // world] for ends in ([2], [5, 6, 15])]
        assert pol.refreshed == [w for w in want if w], (rank, pol.refreshed)
    if rank == 0:
        q.put(rec.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_rollout_matches_single_process():
    from robo_vln_amd.rollout import rollout
    cfg, hi_sd, lo_sd, obs_fn, done_fn, G = _setup()
    single = rollout(_OraclePolicy(cfg, hi_sd, lo_sd), obs_fn, done_fn, G, 3, cfg.num_recurrent_layers, cfg.hidden, "cpu").numpy()
    assert single.shape == (3, G, 7)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    sharded = q.get(timeout=240)
    assert not isinstance(sharded, str), sharded
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_allclose(sharded, single, atol=2e-6, rtol=0)


def test_episode_reset_equals_fresh_state():
    """mask=0 for an environment must equal restarting it from zero hidden state (hierarchical_trainer.py:1143-1159)."""
    from robo_vln_amd.rollout import rollout, shard_range, records_to_actions
    cfg, hi_sd, lo_sd, obs_fn, done_fn, G = _setup()
    pol = _OraclePolicy(cfg, hi_sd, lo_sd)
    rec = rollout(pol, obs_fn, done_fn, G, 2, cfg.num_recurrent_layers, cfg.hidden, "cpu")
    # env 2 was done at t=0 -> its step-1 record equals a fresh single-step rollout on step-1 observations
    R = cfg.num_recurrent_layers
    z = torch.zeros(R, G, cfg.hidden)
    fresh, _, _ = pol.act(obs_fn(1, 0, G), z, z.clone(), None, torch.zeros(G))
    np.testing.assert_allclose(rec[1, 2].numpy(), fresh[2].numpy(), atol=1e-6)
    assert np.abs(rec[1, 1].numpy() - fresh[1].numpy()).max() > 1e-4     # env 1 carried its state
    with pytest.raises(ValueError):
        shard_range(10, 4, 0)
    st, lin, ang, stop = records_to_actions(rec[0])
    assert st.shape == (G,) and ang.abs().max() <= 1 and set(stop.tolist()) <= {0.0, 1.0}


def test_sharded_rollout_world_size_8():
    """The driver's 8-GPU layout on CPU: 8 gloo ranks x 2 environments, an uneven `done` pattern (one rank resets both of its
    environments, five ranks none), the per-rank cached-instruction path of rollout() -- the gathered (3, 16, 7) records must be
    the single-process rollout's, on every rank."""
    from robo_vln_amd.rollout import rollout
    cfg, hi_sd, lo_sd, obs_fn, done_fn, G = _setup(16, 2048)
    torch.set_num_threads(8)
    single = rollout(_OraclePolicy(cfg, hi_sd, lo_sd), obs_fn, done_fn, G, 3, cfg.num_recurrent_layers, cfg.hidden, "cpu").numpy()
    assert single.shape == (3, 16, 7)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    sharded = q.get(timeout=600)
    assert not isinstance(sharded, str), sharded
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    np.testing.assert_allclose(sharded, single, atol=2e-6, rtol=0)
