"""hcm_act_gather with a REAL peer on a one-GPU box (round-4 review item 8): two processes, each with its own handle on cuda:0, whose library
communicators meet through tests/stub_rccl (a shared-memory stand-in for the six RCCL symbols csrc/comm.cpp resolves, loaded with HCM_RCCL_LIB).
Covered: a normal sharded rollout (every rank holds the whole record, rank-major, bit-equal to the rows each shard computes alone); a rank whose
step fails (it raises; the peer sees that rank's NaN rows and leaves rollout() at the SAME step, aborting its communicator); hcm_comm_abort
releasing a rank whose peer never joined; ranks that disagree with the handle on B are refused before any collective.
SCALE_rNN stays 'unmeasured' -- this is about not discovering a hang on the first 8-GPU node (reference counterpart: none, the reference steps one
environment in one process, hierarchical_trainer.py:1088-1107)."""
import os
import subprocess
import sys
import time

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_DIR = os.path.join(ROOT, "tests", "stub_rccl")
STUB = os.path.join(STUB_DIR, "libhcm_stub_rccl.so")

pytestmark = pytest.mark.gpu


def _build_stub():
    src = os.path.join(STUB_DIR, "stub_rccl.cpp")
    if not os.path.exists(STUB) or os.path.getmtime(STUB) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wl,-soname,libhcm_stub_rccl.so", "-o", STUB, src,
                        "-lrt", "-lpthread"], check=True)


def _worker(rank, world, port, scenario, q):
    try:
        q.put((rank, "ok", _body(rank, world, port, scenario)))
    except BaseException as e:  # noqa: BLE001  (the parent asserts on what each rank reports)
        q.put((rank, "raised", f"{type(e).__name__}: {e}"))


def _body(rank, world, port, scenario):
    os.environ["HCM_RCCL_LIB"] = STUB
    sys.path.insert(0, ROOT)
    import hcm_pkg
    hcm_pkg.load()
    import torch.distributed as dist
    from robo_vln_amd import synth, _lib
    from robo_vln_amd.config import HCMConfig
    from robo_vln_amd.policy import HCMEngine, Policy
    from robo_vln_amd.rollout import rollout
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=1).validate()
    Bl = 2
    G = Bl * world
    hi_sd, lo_sd = synth.make_weights(cfg, seed=4)
    eng = HCMEngine(cfg, hi_sd, lo_sd, max_batch=Bl, precision="fp16", graph=True)
    pol = Policy(eng)
    assert eng.comm_init() == world
    R = cfg.num_recurrent_layers
    T = 4

    def full_obs(t):
        return {k: torch.from_numpy(v).cuda() for k, v in synth.make_observations(cfg, G, step=t, seed=4).items()}

    def obs_fn(t, lo, hi):
        o = {k: v[lo:hi].contiguous() for k, v in full_obs(t).items()}
        if scenario == "fail" and rank == 1 and t == 2:
            o["instruction"] = torch.zeros(hi - lo, 600, dtype=torch.int64, device="cuda")      # L = 600 > the position table: this rank's step is refused
        return o

    def done_fn(t, lo, hi):
        return torch.zeros(hi - lo, dtype=torch.bool, device="cuda")

    out = None
    if scenario in ("normal", "fail"):
        t0 = time.time()
        try:
            recs = rollout(pol, obs_fn, done_fn, Bl, T, R, cfg.hidden, "cuda", world=world, rank=rank)
        except Exception as e:
            # both ranks must get here at step 2, promptly (nobody sits in the next step's collective until a timeout)
            assert scenario == "fail", repr(e)
            assert time.time() - t0 < 60
            torch.cuda.synchronize()
            dist.barrier()
            raise
        torch.cuda.synchronize()
        assert scenario == "normal"
        assert recs.shape == (T, G, 7) and bool(torch.isfinite(recs).all())
        # every shard computed alone (plain act(), no collective, fresh state) on this rank: the gathered record holds those rows, rank-major, bit for bit
        eng2 = HCMEngine(cfg, hi_sd, lo_sd, max_batch=Bl, precision="fp16", graph=True)
        for r in range(world):
            hh = torch.zeros(R, Bl, cfg.hidden, device="cuda"); lh = torch.zeros(R, Bl, cfg.hidden, device="cuda"); m = torch.zeros(Bl, device="cuda")
            for t in range(T):
                o = {k: v[r * Bl:(r + 1) * Bl].contiguous() for k, v in full_obs(t).items()}
                rec, hh, lh = eng2.act(o, hh, lh, m)
                hh, lh = hh.clone(), lh.clone()
                m = torch.ones(Bl, device="cuda")
                assert torch.equal(rec, recs[t, r * Bl:(r + 1) * Bl]), (r, t)
        eng2.close()
        out = recs.cpu().numpy().tobytes().hex()[:64]
        # ranks must pass a B the handle can run: refused on every rank BEFORE any collective (nobody hangs on mismatched counts)
        lib = _lib.lib()
        z = torch.zeros(R, 8, cfg.hidden, device="cuda")
        buf = torch.zeros(8 * world, 7, device="cuda")
        o = full_obs(0)
        rc = lib.hcm_act_gather(eng._h, o["rgb"].data_ptr(), _lib.HCM_F32, o["depth"].data_ptr(), o["instruction"].data_ptr(), _lib.HCM_I64, None, Bl + 1, 20,
                                z.data_ptr(), z.data_ptr(), z.data_ptr(), buf.data_ptr(), z.clone().data_ptr(), z.clone().data_ptr(), 0, buf.data_ptr(), None)
        assert rc == -1 and b"no collective was entered" in lib.hcm_last_error(eng._h)
        torch.cuda.synchronize()
        dist.barrier()
    elif scenario == "fail_lib":
        # the failure is INSIDE the library on rank 1 (hcm_act_ex refuses L = 513): the library itself joins with the poisoned record
        lib = _lib.lib()
        o = {k: v[rank * Bl:(rank + 1) * Bl].contiguous() for k, v in full_obs(0).items()}
        z = torch.zeros(R, Bl, cfg.hidden, device="cuda")
        m = torch.zeros(Bl, device="cuda")
        for _ in range(3):
            got, _, _ = eng.act(o, z, z, m, gather=True)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(got).all())
        if rank == 1:
            ids = torch.zeros(Bl, 513, dtype=torch.int64, device="cuda")
            loc = torch.zeros(Bl, 7, device="cuda"); gat = torch.zeros(G, 7, device="cuda")
            rc = lib.hcm_act_gather(eng._h, o["rgb"].data_ptr(), _lib.HCM_F32, o["depth"].data_ptr(), ids.data_ptr(), _lib.HCM_I64, None, Bl, 513,
                                    z.data_ptr(), z.data_ptr(), m.data_ptr(), loc.data_ptr(), z.clone().data_ptr(), z.clone().data_ptr(), 0, gat.data_ptr(), None)
            torch.cuda.synchronize()
            assert rc != 0 and b"NaN record" in lib.hcm_last_error(eng._h)
        else:
            gat, _, _ = eng.act(o, z, z, m, gather=True)
            torch.cuda.synchronize()
        assert bool(torch.isfinite(gat[:Bl]).all()) and bool(torch.isnan(gat[Bl:]).all())       # rank 0's rows are real, rank 1's are the poison
        out = "seen"
        dist.barrier()
    elif scenario == "abort":
        # rank 1 never steps; rank 0's step sits in the collective on its stream until rank 0's OWN hcm_comm_abort releases it
        if rank == 0:
            o = {k: v[:Bl].contiguous() for k, v in full_obs(0).items()}
            z = torch.zeros(R, Bl, cfg.hidden, device="cuda")
            rec, _, _ = eng.act(o, z, z, torch.zeros(Bl, device="cuda"), gather=True)
            time.sleep(1.0)
            t0 = time.time()
            eng.comm_abort()
            torch.cuda.synchronize()
            assert time.time() - t0 < 30 and eng.comm_world == 0
            # the handle still steps on its own afterwards
            rec2, _, _ = eng.act(o, z, z, torch.zeros(Bl, device="cuda"))
            torch.cuda.synchronize()
            assert bool(torch.isfinite(rec2).all())
            out = "released"
        dist.barrier()
        if rank == 1:
            eng.comm_abort()
            out = "idle"
    eng.close()
    dist.destroy_process_group()
    return out


def _run(scenario, port):
    _build_stub()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, scenario, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = {}
    try:
        for _ in ps:
            rank, status, val = q.get(timeout=600)
            res[rank] = (status, val)
    finally:
        for p in ps:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    return res


def test_two_rank_rollout_on_the_library_collective():
    res = _run("normal", 29611)
    assert res[0][0] == "ok" and res[1][0] == "ok", res
    assert res[0][1] == res[1][1]                 # both ranks hold the same gathered records


def test_peer_of_a_failed_rank_leaves_at_the_same_step():
    res = _run("fail", 29612)
    assert res[1][0] == "raised" and "instruction must be" in res[1][1], res          # the rank whose observation was malformed reports its own error
    assert res[0][0] == "raised" and "rank(s) [1] contributed a NaN action record" in res[0][1] and "step 2" in res[0][1], res


def test_comm_abort_releases_a_rank_whose_peer_never_joined():
    res = _run("abort", 29613)
    assert res[0] == ("ok", "released") and res[1] == ("ok", "idle"), res


def test_library_side_failure_reaches_the_peer_as_nan_rows():
    res = _run("fail_lib", 29614)
    assert res[0] == ("ok", "seen") and res[1] == ("ok", "seen"), res
