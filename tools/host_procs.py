"""Host side of an env-sharded node, measurable on ONE GPU (round-5 review item 7): N processes, each pinned to 2 cores and driving its own engine
(BASELINE configs[1] shapes at B environments) through the real graph-replay call, all sharing the one device.  Reported per process: the time spent INSIDE
act() (argument marshalling + hipGraphLaunch), with a synchronisation behind every call so that queue back-pressure does not leak into it.  What this bounds: whether 8
ranks' host threads, each on its own 2 cores of the node, can keep enqueueing steps at the single-process rate; what it cannot show: device time (the GPU is shared here).
usage: python tools/host_procs.py [N=8] [B=8] [seconds=3]      (parent; prints one JSON object)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(idx, B, seconds, cores):
    try:
        os.sched_setaffinity(0, set(cores))
    except Exception:
        pass
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    torch.set_num_threads(len(cores))
    import hcm_pkg; hcm_pkg.load()
    from robo_vln_amd import synth
    from robo_vln_amd.config import baseline_config
    from robo_vln_amd.policy import HCMEngine
    cfg = baseline_config(1)
    hi_sd, lo_sd = synth.make_weights(cfg, seed=0)
    eng = HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="fp16", graph=True, chain_graphs=False)
    o = synth.make_observations(cfg, B, step=idx, seed=0, rgb_uint8=True)
    obs = {k: torch.from_numpy(v).cuda() for k, v in o.items()}
    R = cfg.num_recurrent_layers
    hh = torch.zeros(R, B, cfg.hidden, device="cuda"); lh = torch.zeros(R, B, cfg.hidden, device="cuda")
    m = torch.ones(B, device="cuda")
    for _ in range(10):
        eng.act(obs, hh, lh, m)
    torch.cuda.synchronize()
    # rendezvous: every child is built before anybody measures
    open(os.path.join(os.environ["HP_DIR"], f"ready{idx}"), "w").close()
    n_all = int(os.environ["HP_N"])
    while sum(os.path.exists(os.path.join(os.environ["HP_DIR"], f"ready{i}")) for i in range(n_all)) < n_all:
        time.sleep(0.01)
    host, steps, t0 = 0.0, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        h0 = time.perf_counter()
        eng.act(obs, hh, lh, m)
        host += time.perf_counter() - h0
        torch.cuda.synchronize()
        steps += 1
    wall = time.perf_counter() - t0
    print(json.dumps({"proc": idx, "cores": list(cores), "steps": steps, "host_us_per_step": round(host / steps * 1e6, 1), "wall_ms_per_step": round(wall / steps * 1e3, 3)}))
    eng.close()


def run(N, B, seconds):
    import tempfile
    ncpu = os.cpu_count() or 2
    out = []
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, HP_DIR=d, HP_N=str(N))
        ps = []
        for i in range(N):
            cores = [(2 * i) % ncpu, (2 * i + 1) % ncpu]
            ps.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(i), str(B), str(seconds), ",".join(map(str, cores))],
                                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, cwd=ROOT))
        for p in ps:
            o, _ = p.communicate(timeout=600)
            for line in o.decode().splitlines():
                if line.startswith("{"):
                    out.append(json.loads(line))
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), [int(c) for c in sys.argv[5].split(",")])
        sys.exit(0)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    sec = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
    one = run(1, B, sec)
    many = run(N, B, sec)
    hs = [r["host_us_per_step"] for r in many]
    print(json.dumps({"processes": N, "cores_per_process": 2, "host_cores": os.cpu_count(), "per_process_batch": B,
                      "host_us_per_step": {"single_process": one[0]["host_us_per_step"] if one else None, "mean": round(sum(hs) / max(len(hs), 1), 1), "max": max(hs) if hs else None,
                                           "per_process": hs},
                      "wall_ms_per_step_shared_gpu": [r["wall_ms_per_step"] for r in many],
                      "note": "time inside act() (marshalling + forked-graph hipGraphLaunch) per process, a synchronisation behind every call; N processes pinned to 2 cores each "
                              "share ONE GPU, so wall time per step is the shared device, not a scaling number"}))
