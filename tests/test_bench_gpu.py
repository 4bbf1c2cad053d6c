"""bench.py's output contract on the GPU: one JSON line with the driver's keys plus `roofline` and `cpu_baseline`, launched the two ways the
driver launches it -- plain `python bench.py` (N = 1) and under `python -m torch.distributed.run` (here with ONE rank: the box has one GPU;
the RCCL process group, the per-step all-gather on the critical path, the rank-order self-check and the overlapped variant all run)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"}


def _run(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # exactly one line on stdout
    return json.loads(lines[0])


def test_bench_single_process_line():
    d = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2", "--sustain", "0"])
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["metric"].startswith("policy env-steps/sec") and d["unit"] == "env-steps/s" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and "model" not in d["config"] and d["config"]["workload"].startswith("BASELINE.json configs[1]")
    assert abs(d["value"] - d["config"]["global_batch"] / d["ms_per_step"] * 1e3) <= 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and 0 < c["value"] < d["value"] and c["unit"] == "env-steps/s" and "batch" in c["sample"]


def test_bench_under_torch_distributed_run_one_rank():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29541",
              "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2", "--sustain", "0", "--no-cpu-baseline", "--no-kernel-probe"])
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["global_batch"] == d["config"]["per_gpu_batch"]
