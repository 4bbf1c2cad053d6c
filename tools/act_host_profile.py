"""Host-side cost of one act() at B = 1 (the reference's own eval loop steps ONE environment and needs the action before it can step the
simulator, so a step's latency is host enqueue + GPU execution): wall per step with a synchronise after every step, the host enqueue time
alone, and a cProfile of the enqueue path.  usage: python tools/act_host_profile.py [B]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import synth
from robo_vln_amd.config import HCMConfig
from robo_vln_amd.policy import HCMEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
REUSE = bool(int(os.environ.get("REUSE", "0")))      # REUSE=1: the instruction stream cached after the first step (HCM_ACT_REUSE_INSTRUCTION, opt-in)
cfg = HCMConfig().validate()
hi, lo = synth.make_weights(cfg, seed=0)
for graph, chain in ((True, False), (True, True), (False, False)):
    eng = HCMEngine(cfg, hi, lo, max_batch=B, precision="fp16", graph=graph, chain_graphs=chain)
    obs = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in synth.make_observations(cfg, B, step=0, seed=0, rgb_uint8=True).items()}
    R = cfg.num_recurrent_layers
    hh = torch.zeros(R, B, cfg.hidden, device="cuda"); lh = torch.zeros(R, B, cfg.hidden, device="cuda")
    m = torch.ones(B, device="cuda")
    eng._stepped = False
    for _ in range(10):
        rec, hh, lh = eng.act(obs, hh, lh, m, reuse_instruction=REUSE and eng._stepped)
        eng._stepped = True
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter(); host = 0.0
    for _ in range(n):
        h0 = time.perf_counter()
        rec, hh, lh = eng.act(obs, hh, lh, m, reuse_instruction=REUSE and eng._stepped)
        host += time.perf_counter() - h0
        torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        rec, hh, lh = eng.act(obs, hh, lh, m, reuse_instruction=REUSE and eng._stepped)
    torch.cuda.synchronize()
    pipe = (time.perf_counter() - t0) / n
    print(f"B={B} graph={graph} chain_graphs={chain}: synchronised per step {wall * 1e3:.3f} ms (host enqueue {host / n * 1e6:.0f} us), back-to-back {pipe * 1e3:.3f} ms per step")
    if graph and not chain:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(200):
            rec, hh, lh = eng.act(obs, hh, lh, m, reuse_instruction=REUSE and eng._stepped)
        pr.disable(); torch.cuda.synchronize()
        st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(18)
    eng.close()
