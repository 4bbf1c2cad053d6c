"""The driver's multi-GPU command line -- `python bench.py --gpus N --steps K --warmup W` -- must launch itself: it re-executes under
torch.distributed.run (one rank per GPU on 127.0.0.1), and rank 0 prints ONE JSON line carrying the weak AND the strong leg, the ranks seen
by the rank-tagged all-gather and the host enqueue time.  Run here at world size 2 on CPU with the stand-in engine (HCM_BENCH_STUB=1: gloo,
no GPU, no libhcm) -- the launch plumbing, rendezvous, legs and the output contract are the product code's own."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update({"HCM_BENCH_STUB": "1", "HCM_BENCH_STUB_BATCH": "2", "OMP_NUM_THREADS": "2"})
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    if env_extra:
        env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    return p


def test_gpus_2_spawns_its_own_ranks_and_prints_one_line():
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--prewarm", "1"])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["ranks_seen"] == [0, 1]
    assert out["scaling"] == "weak" and out["config"]["global_batch"] == 2 * out["config"]["per_gpu_batch"]
    assert out["weak"]["value"] == out["value"] and out["weak"]["per_gpu_batch"] == out["config"]["per_gpu_batch"]
    st = out["strong"]
    assert st["total_batch"] == 2 * st["per_gpu_batch"] and st["value"] > 0 and st["ranks_seen"] == [0, 1]
    assert out["host_us_per_step"] > 0 and out["value"] > 0 and out["higher_is_better"] is True
    assert "STUB" in out["metric"]           # a stub line can never be mistaken for a measurement


def test_gpus_1_stays_in_process():
    p = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--prewarm", "1"])
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.strip()][0])
    assert out["n_gpus"] == 1 and "strong" not in out and "ranks_seen" not in out


def test_world_size_mismatch_is_refused():
    p = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert p.returncode != 0 and "--gpus 4" in (p.stderr + p.stdout)
