"""How the three encoder chains of a step REALLY interleave: wall-clock stamps written by marker kernels between the launches of each chain
(`make DEV=1` library, HCM_MARKS=1; forward.cpp Fwd::mark), read after hipGraph-replayed steps at the bench configuration.  rocprofv3's
kernel trace serialises the streams, so this is the only in-step timeline there is.  Runs the step with all chains and -- for reference --
with single chains (HCM_SKIP drops the others: 1 RGB, 4 depth, 8 BERT), each in a fresh process.
usage: python tools/step_marks.py [B]      (prints a table per configuration: milestone, us since the step's first stamp, segment length)"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch, hcm_pkg
    hcm_pkg.load()
    from robo_vln_amd import synth, _lib
    from robo_vln_amd.config import baseline_config
    from robo_vln_amd.policy import HCMEngine
    B = int(sys.argv[2])
    cfg = baseline_config(1)
    hi_sd, lo_sd = synth.make_weights(cfg, seed=0)
    eng = HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision="fp16", graph=True)
    sets = []
    for k in range(2):
        o = synth.make_observations(cfg, B, step=k, seed=0, rgb_uint8=True)
        sets.append({kk: torch.from_numpy(v).cuda() for kk, v in o.items()})
    R = cfg.num_recurrent_layers
    hh = torch.zeros(R, B, cfg.hidden, device="cuda"); lh = torch.zeros(R, B, cfg.hidden, device="cuda")
    m = torch.ones(B, device="cuda")
    for i in range(30):
        rec, hh, lh = eng.act(sets[i & 1], hh, lh, m)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for i in range(40):
        rec, hh, lh = eng.act(sets[i & 1], hh, lh, m)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 40 * 1e3
    lib = _lib.lib()
    out = (C.c_uint64 * 256)()
    names = C.create_string_buffer(16384)
    n = lib.hcm_debug_marks(eng._h, out, names, 16384)
    nm = names.value.decode().split("\n")[:n]
    print(f"STEP_MS {ms:.3f}")
    for i in range(n):
        print(f"MARK {nm[i]} {out[i]}")
    eng.close()
    sys.exit(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def run(skip):
    env = dict(os.environ, HCM_DEV_LIB="1", HCM_MARKS="1")
    if skip: env["HCM_SKIP"] = str(skip)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(B)], capture_output=True, text=True, env=env, timeout=600)
    ms, marks = None, []
    for l in p.stdout.splitlines():
        if l.startswith("STEP_MS"): ms = float(l.split()[1])
        if l.startswith("MARK"): _, n, t = l.split(); marks.append((n, int(t)))
    if ms is None: print(p.stderr[-2000:])
    return ms, marks
for title, skip in (("all three chains (the step)", 0), ("RGB chain alone (+ tail)", 12), ("BERT chain alone (+ tail)", 5), ("depth chain alone (+ tail)", 9)):
    ms, marks = run(skip)
    if not marks: continue
    t0 = min(t for _, t in marks if t)
    print(f"\n## {title}: {ms:.3f} ms per step (wall, with the marker launches)\n")
    print("| chain | milestone | reached at (us) | segment (us) |\n|---|---|---|---|")
    chains = {}
    for n, t in marks:
        if not t: continue
        chains.setdefault(n.split(".")[0], []).append((t, n))
    for ch, lst in chains.items():
        lst.sort()
        prev = None
        for t, n in lst:
            us = (t - t0) / 100.0
            print(f"| {ch} | {n} | {us:8.1f} | {'' if prev is None else f'{us - prev:7.1f}'} |")
            prev = us
