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

    def act(self, obs, hh, lh, prev, masks):
        return self.o.act(obs, hh, lh, masks.numpy())


def _setup():
    import hcm_pkg
    hcm_pkg.load()
    from robo_vln_amd import synth
    from robo_vln_amd.config import HCMConfig
    cfg = HCMConfig(rgb_hw=64, depth_hw=64, instr_len=8, bert_layers=1)
    hi_sd, lo_sd = synth.make_weights(cfg, seed=1)
    G = 4
    allobs = [synth.make_observations(cfg, G, step=t, seed=1) for t in range(3)]

    def obs_fn(t, lo, hi):
        return {k: v[lo:hi] for k, v in allobs[t].items()}

    def done_fn(t, lo, hi):
        d = torch.zeros(G, dtype=torch.bool)
        if t == 0:
            d[2] = True
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
    torch.set_num_threads(2)
    from robo_vln_amd.rollout import rollout
    cfg, hi_sd, lo_sd, obs_fn, done_fn, G = _setup()
    rec = rollout(_OraclePolicy(cfg, hi_sd, lo_sd), obs_fn, done_fn, G // world, 3, cfg.num_recurrent_layers, cfg.hidden,
                  "cpu", world, rank)
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
