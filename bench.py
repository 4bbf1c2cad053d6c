"""bench.py -- policy env-steps/sec of the batched HCM act() on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config {0,1,3,4}]       (N > 1: re-executes itself under torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload (BASELINE.json configs[1]/[2], the configuration the metric is quoted on): per GPU 64 environments, 256x256
RGB-D frames, 80-token instruction, full HCM model (2x ResNet-50 RGB, 2x GN-ResNet-50 depth, BERT-base, cross-modal block,
2x LSTM + heads), 16-bit storage / fp32 accumulate, random-init weights, synthetic inputs resident in HBM (two observation sets
used alternately).  One step = one fused act() (hi -> argmax -> lo) over the rank's 64 environments, followed (N>1) by ONE RCCL
all-gather of the (64,7) action records ON THE CRITICAL PATH: a rollout cannot produce the next observation before it
has the record.  Weak scaling: global batch = 64 * N.  Prints ONE JSON line on rank 0.

The other BASELINE configs are parity-test shapes; `--config` times them for their per-config roofline:
  0  B=4, 128x128, L=20, N=2 (the reference's CPU-runnable plumbing case)            act()
  3  SimpleDepthCNN + 1-layer Visual_Ling_Attn, B=256 (memory-bound path)             robo-vln_amd/probe.py, HBM roofline
  4  high-level model, ResNet-50 RGB + N=6 decoder, L=160, B=128 (MFMA-bound path)    hcm_high_forward, MFMA roofline
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense 16-bit MFMA peak (MI355X_MICROARCH.md); bf16 and fp16 have the same rate
PEAK_HBM_TBPS = 8.0
# algorithmic work per env-step, SURVEY.md 8a table (2 x MAC of conv + matmul + attention + RNN gates)
GFLOP = {0: 9.34, 1: 36.9, 4: 42.9}
BATCH = {0: 4, 1: 64, 3: 256, 4: 128}
WORKLOAD = {
    0: "BASELINE.json configs[0]: full HCM act(), 128x128 RGB-D, L=20, VLA N=2, LSTM-512",
    1: "BASELINE.json configs[1]: full HCM act() (hi->argmax->lo), 256x256 RGB-D, L=80, VLA N=1, LSTM-512",
    3: "BASELINE.json configs[3]: SimpleDepthCNN(256x256 depth) -> one visual token -> Visual_Ling_Attn(N=1) over a pre-computed (B,80,768) instruction tensor",
    4: "BASELINE.json configs[4]: high-level model alone, ResNet-50 RGB + GN-ResNet-50 depth + BERT + 6-layer cross-modal decoder, 256x256, L=160",
}


def host_cores():
    """CPUs this process may actually use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return n


def cpu_baseline(cfg, hi_sd, lo_sd, batch=16, steps=8):
    """The CPU oracle (kind 'port': torch-CPU fp32 restatement validated against the imported reference) timed on this
    host's cores on a bounded sample of the same workload (8 steps at batch 16 instead of 64: 10-15 s of CPU work)."""
    import numpy as np
    import torch
    from oracle import hcm_oracle
    from robo_vln_amd import synth
    ncores = host_cores()
    torch.set_num_threads(ncores)
    ora = hcm_oracle.PolicyOracle(cfg, hi_sd, lo_sd)
    R = cfg.num_recurrent_layers
    hh = torch.zeros(R, batch, cfg.hidden)
    lh = torch.zeros(R, batch, cfg.hidden)
    obs = synth.make_observations(cfg, batch, step=0, seed=0)
    mask = np.zeros(batch, np.float32)
    _, hh, lh = ora.act(obs, hh, lh, mask)                 # warm-up
    mask[:] = 1
    t0 = time.time()
    for _ in range(steps):
        _, hh, lh = ora.act(obs, hh, lh, mask)
    dt = time.time() - t0
    return {"value": round(batch * steps / dt, 3), "unit": "env-steps/s", "cores": ncores, "kind": "port",
            "sample": f"{steps} act() steps at batch {batch} (the GPU line runs batch 64), 256x256 RGB-D, L=80, fp32, torch {torch.__version__} "
                      f"with {torch.get_num_threads()} threads"}


def cpu_baseline_protocol(cfg, hi_sd, lo_sd, budget_s=None):
    """BASELINE.md section 3's protocol: batches 4, 16 and 64, best of 5 timed steps each after a warm-up step.  budget_s (the
    default run: ~30 s): the timed steps of a batch size stop once the budget is spent -- a batch-64 step is ~5-8 s of CPU work, so it gets
    fewer than 5; `steps_timed` says how many each figure is the best of.  --cpu-batches runs the unbounded protocol."""
    import numpy as np
    import torch
    from oracle import hcm_oracle
    from robo_vln_amd import synth
    ncores = host_cores()
    torch.set_num_threads(ncores)
    ora = hcm_oracle.PolicyOracle(cfg, hi_sd, lo_sd)
    R = cfg.num_recurrent_layers
    out = {}
    t_all = time.time()
    for batch in (4, 16, 64):
        hh = torch.zeros(R, batch, cfg.hidden); lh = torch.zeros(R, batch, cfg.hidden)
        obs = synth.make_observations(cfg, batch, step=0, seed=0)
        mask = np.zeros(batch, np.float32)
        _, hh, lh = ora.act(obs, hh, lh, mask)
        mask[:] = 1
        best, n = None, 0
        for _ in range(5):
            t0 = time.time()
            _, hh, lh = ora.act(obs, hh, lh, mask)
            dt = time.time() - t0
            best = dt if best is None else min(best, dt)
            n += 1
            if budget_s is not None and time.time() - t_all + dt > budget_s * (0.1 if batch == 4 else 0.4 if batch == 16 else 1.0):
                break
        out[f"batch_{batch}"] = {"value": round(batch / best, 3), "unit": "env-steps/s", "ms_per_step": round(best * 1e3, 1), "steps_timed": n}
    out["cores"] = ncores
    out["seconds"] = round(time.time() - t_all, 1)
    out["protocol"] = ("best of up to 5 act() steps per batch size after one warm-up step (BASELINE.md section 3), CPU oracle, fp32" +
                       (f"; bounded to ~{budget_s:.0f} s of CPU work in the default run (steps_timed per batch)" if budget_s is not None else ""))
    return out


def _time_op(run, n=60, warm=30):
    """HIP events on torch's current stream, which is the stream the operator entry points are handed."""
    import torch
    for _ in range(warm):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n          # ms


def dominant_kernel_probe(batch, L=80, dtype="f16"):
    """The launches with the largest share of a step's kernel time are the GEMMs of the fp16 BERT encoder
    (profiles/r2b_kernel_trace_bench.md): 12 launches each of QKV (768 -> 2304) and FFN1 (768 -> 3072, GELU) on the 256 x 256-tile
    kernel `gemm256f_kernel` (one barrier per K tile; round 4), and of attention-output (768 -> 768) and FFN2 (3072 -> 768) on `igemm_dma_kernel` (at B = 64 their
    output is too narrow for 256-wide tiles), over M = batch * L token rows.  Each is timed live here through the library's operator
    entry point (same kernel, same tile choice as inside the step) and priced against the dense 16-bit MFMA peak: algorithmic FLOPs
    per launch = 2*M*N*K.  The FFN1 launch is the single most expensive one and is the `roofline` of the JSON line."""
    import ctypes as C
    import torch
    from robo_vln_amd import _lib
    lib = _lib.lib()
    M = batch * L
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tdt, hdt = (torch.bfloat16, _lib.HCM_BF16) if dtype == "bf16" else (torch.float16, _lib.HCM_F16)
    out = {}
    for name, N, K, act, res in (("ffn1", 3072, 768, _lib.ACT_GELU, False), ("qkv", 2304, 768, 0, False),
                                 ("ffn2", 768, 3072, 0, True), ("attn_out", 768, 768, 0, True)):
        x = (torch.randn(M, K, device="cuda") * 0.5).to(tdt)
        w = (torch.randn(N, K, device="cuda") * 0.03).to(tdt)
        b = torch.randn(N, device="cuda") * 0.1
        r = (torch.randn(M, N, device="cuda") * 0.5).to(tdt) if res else None
        y = torch.empty(M, N, device="cuda", dtype=tdt)

        def run():
            rc = lib.hcm_op_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if r is not None else None, y.data_ptr(),
                                   hdt, M, N, K, act, 0, st)
            assert rc == 0
        ms = _time_op(run)
        fl = 2.0 * M * N * K
        out[name] = {"shape": f"M={M} N={N} K={K}", "us_per_launch": round(ms * 1e3, 2), "gflop_per_launch": round(fl / 1e9, 2),
                     "tflops": round(fl / ms / 1e9, 1), "frac_of_peak": round(fl / ms / 1e9 / PEAK_BF16_TFLOPS, 4), "launches_per_step": 12}
    return out


def hbm_kernel_probe(batch):
    """Second, HBM-bound probe: the fused bottleneck kernel of the RGB ResNet-50 pair, `bneck231r_kernel` (3x3 conv 64->64 + ReLU, 1x1
    expansion 64->256 + identity + ReLU and the next block's 1x1 reduction in one launch), at its layer1 middle-block shape on the
    hi|lo pair workload (M = 2*B*4096 pixels).  Algorithmic bytes = 2 B/elem * M * (64 in + 256 identity + 256 out + 64 next)."""
    import ctypes as C
    import torch
    from robo_vln_amd import _lib
    lib = _lib.lib()
    B, H, W, C1, CN = 2 * batch, 64, 64, 64, 64
    C3 = 4 * C1
    bf = torch.float16                 # the storage type of the RGB trunks in 16-bit mode (round 2: fp16, range-calibrated)
    x = torch.randn(B, H, W, C1, device="cuda").to(bf)
    w2 = (torch.randn(C1, 3, 3, C1, device="cuda") * 0.05).to(bf)
    b2 = torch.randn(C1, device="cuda")
    w3 = (torch.randn(C3, 1, 1, C1, device="cuda") * 0.05).to(bf)
    b3 = torch.randn(C3, device="cuda")
    w1 = (torch.randn(CN, 1, 1, C3, device="cuda") * 0.05).to(bf)
    b1 = torch.randn(CN, device="cuda")
    idt = torch.randn(B, H, W, C3, device="cuda").to(bf)
    y = torch.empty_like(idt)
    o1 = torch.empty(B, H, W, CN, device="cuda", dtype=bf)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        rc = lib.hcm_op_bottleneck_tail_next(x.data_ptr(), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), idt.data_ptr(),
                                             y.data_ptr(), w1.data_ptr(), b1.data_ptr(), o1.data_ptr(), _lib.HCM_F16, B, H, W, C1, 1, CN, st)
        assert rc == 0
    ms = _time_op(run, 100, 100)
    M = B * H * W
    gbytes = 2.0 * M * (C1 + 2 * C3 + CN) / 1e9
    return {"kernel": "bneck231r_kernel<f16,128,64,64>: conv3x3 64->64 + conv1x1 64->256 + identity + next conv1x1 256->64 @64x64, hi|lo pair (M=2*B*4096)",
            "us_per_launch": round(ms * 1e3, 2), "bound": "hbm", "achieved_TBps": round(gbytes / ms, 3), "peak_TBps": PEAK_HBM_TBPS,
            "frac": round(gbytes / ms / PEAK_HBM_TBPS, 4)}


def pmc_traffic(config=1):
    """HBM bytes from the PMC counters: written by tools/pmc_traffic.py from separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`
    passes of this same command (x2 gfx950 correction on FETCH_SIZE, MI355X_MICROARCH.md); never a literal in this file."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json" if config == 1 else f"pmc_traffic_cfg{config}.json")
    try:
        return json.load(open(p))
    except Exception:
        return None


def dtype_string(args, eng):
    """What the 16-bit engine actually stores (the mode asked for and what the range calibration did to it)."""
    if args.precision == "fp32":
        return "fp32"
    if args.precision == "bf16":
        return "bf16 (bf16 storage and MFMA in BERT -- f32 residual stream -- and the cross-modal block; both trunk kinds on range-folded fp16 tiles; fp32 accumulate, fp32 recurrent state; DESIGN.md section 4)"
    fb = sorted(eng.fp16_fallback)
    fold = sorted(getattr(eng, "range_fold", ()))
    tail = ("; range fold: " + ", ".join(fold)) if fold else ""
    if not fb:
        return "fp16 (range-calibrated fp16 storage and MFMA in BERT, both trunks and the cross-modal block; fp32 accumulate, fp32 recurrent state; DESIGN.md section 5" + tail + ")"
    return "fp16+bf16 (fp16 storage / MFMA, fp32 accumulate; moved to bf16 by the range calibration: " + ", ".join(fb) + tail + ")"


# ------------------------------------------------------------------------------------------------------------ workloads
STUB = os.environ.get("HCM_BENCH_STUB", "0") not in ("", "0")     # tests/test_bench_cpu.py only: no GPU, no libhcm, gloo -- exercises the launch plumbing


def _device(local_rank):
    import torch
    return torch.device("cpu") if STUB else torch.device("cuda", local_rank)


def _sync():
    import torch
    if not STUB:
        torch.cuda.synchronize()


class _StubEngine:
    """NOT a measurement path: a deterministic stand-in for HCMEngine so that the exact multi-rank command line of the driver
    (`python bench.py --gpus N ...`: self-spawn, rendezvous, rank-tagged gather, weak + strong legs, one rank-0 JSON line) can be run on a
    CPU-only host under gloo.  Selected by HCM_BENCH_STUB=1 only; the result line says so in `data` and `metric`."""
    fp16_fallback = ()
    range_fold = ()

    def __init__(self, cfg, rank, dev):
        self.cfg, self.rank, self.dev, self.tick = cfg, rank, dev, 0

    def comm_init(self):
        raise RuntimeError("stub engine has no RCCL communicator")

    def act(self, obs, hh, lh, m, reuse_instruction=False, host_frames=False, gather=False):
        import torch
        B = obs["rgb"].shape[0]
        self.tick += 1
        base = obs["depth"].reshape(B, -1)[:, :7].float() + obs["rgb"].reshape(B, -1)[:, :7].float() / 255.0
        return base + 0.001 * self.rank + 1e-6 * self.tick, hh, lh

    def query(self, what):
        return 0

    def close(self):
        pass


def build_act_workload(args, cfg_idx, rank, world, local_rank, strong_batch=0):
    import torch
    from robo_vln_amd import synth
    from robo_vln_amd.config import baseline_config
    if not STUB:
        from robo_vln_amd.policy import HCMEngine
    cfg = baseline_config(cfg_idx)
    B = args.batch or BATCH[cfg_idx]
    if args.total_batch:                                      # strong scaling: the environments of ONE rollout split over the ranks
        if args.total_batch % world:
            raise SystemExit("--total-batch must be a multiple of the number of ranks")
        B = args.total_batch // world
    if STUB:
        B = int(os.environ.get("HCM_BENCH_STUB_BATCH", "4"))  # the plumbing test keeps its frames small; the command line stays the driver's
        strong_batch = min(strong_batch, 2 * B)
    hi_sd, lo_sd = (None, None) if STUB else synth.make_weights(cfg, seed=0)            # full replica per rank
    hi_only = cfg_idx == 4
    dev = _device(local_rank)
    if STUB:
        eng = _StubEngine(cfg, rank, dev)                     # CPU plumbing test of the multi-rank command line only (tests/test_bench_cpu.py)
    else:
        eng = HCMEngine(cfg, hi_sd, None if hi_only else lo_sd, max_batch=max(B, strong_batch), precision=args.precision,
                        graph=not args.no_graph and not hi_only, chain_graphs={"auto": "auto", "0": False, "1": True}[args.chain_graphs])
    steppers = {}

    def make_step(Bx, stager_ok=False):
        """A stepping closure over two resident observation sets at batch Bx (the engine is sized for the largest leg)."""
        if Bx not in steppers:
            steppers[Bx] = _act_stepper(args, cfg, eng, Bx, rank, local_rank, hi_only, dev, stager_ok)
        return steppers[Bx]
    step = make_step(B, stager_ok=True)
    return cfg, B, eng, step, (hi_sd, lo_sd), make_step


def _act_stepper(args, cfg, eng, B, rank, local_rank, hi_only, dev, stager_ok):
    import torch
    from robo_vln_amd import synth
    # two observation sets resident in HBM, used alternately (distinct frames per rank, same shapes)
    sets = []
    for k in range(2):
        o = synth.make_observations(cfg, B, step=2 * rank + k, seed=0, rgb_uint8=True)
        sets.append({"rgb": torch.from_numpy(o["rgb"]).to(dev), "depth": torch.from_numpy(o["depth"]).to(dev),
                     "instruction": torch.from_numpy(o["instruction"]).to(dev)})
    R = cfg.num_recurrent_layers
    state = {"hh": torch.zeros(R, B, cfg.hidden, device=dev), "lh": torch.zeros(R, B, cfg.hidden, device=dev), "tick": 0}
    mask1 = torch.ones(B, device=dev)
    stager = None
    if args.h2d and stager_ok:
        from robo_vln_amd.obs import ObsStager
        stager = ObsStager(B, cfg.rgb_hw, cfg.depth_hw, cfg.instr_len, device=torch.device("cuda", local_rank))
        stager.host["rgb"].copy_(sets[0]["rgb"].cpu())
        stager.host["depth"].copy_(sets[0]["depth"].cpu())
        stager.host["instruction"].copy_(sets[0]["instruction"].cpu().int())
        stager.dev["instruction"].copy_(stager.host["instruction"])       # ids change per episode, not per step: resident

    def step(mask=None, gather=False):
        obs = sets[state["tick"] & 1]
        state["tick"] += 1
        host_frames = False
        if stager is not None and args.h2d_prestage:
            # the frames are copied in front of the step on the caller's stream (what round 1 measured)
            for k in ("rgb", "depth"):
                stager.dev[k].copy_(stager.host[k], non_blocking=True)
            obs = stager.dev
        elif stager is not None:
            # HCM_ACT_HOST_FRAMES: the library reads the pinned host frames itself, one copy per encoder chain inside the (captured) step
            obs = {"rgb": stager.host["rgb"], "depth": stager.host["depth"], "instruction": stager.dev["instruction"]}
            host_frames = True
        m = mask1 if mask is None else mask
        if hi_only:
            logits, state["hh"] = eng.high_forward(obs, state["hh"], m)
            return logits
        r, state["hh"], state["lh"] = eng.act(obs, state["hh"], state["lh"], m,
                                              reuse_instruction=args.reuse_instruction and mask is None and state["tick"] > 3,
                                              host_frames=host_frames, gather=gather)
        return r
    return step


def build_probe_workload(args):
    """configs[3]: the depth-only SimpleCNN -> one token -> 1-layer cross-modal block composition (robo-vln_amd/probe.py)."""
    import torch
    from robo_vln_amd import synth
    from robo_vln_amd.config import HCMConfig
    from robo_vln_amd.probe import DepthCnnVlaProbe
    cfg = HCMConfig(vla_layers=1).validate()
    B, L = args.batch or BATCH[3], cfg.instr_len
    cnn_sd = synth.materialize(synth.simple_cnn_spec("", 1, cfg.depth_hw, 128), "probe_cnn", 0)
    vla_sd = synth.materialize(synth.vla_spec("", cfg, vis_in=128), "probe_vla", 0)
    prec = args.precision
    # hipGraph replay over the two input sets in place (a rollout stages observations into fixed device buffers): no input copies
    probe = DepthCnnVlaProbe(cnn_sd, vla_sd, depth_hw=256, instr_len=L, precision=prec, graph=not args.no_graph, frag_weights=not getattr(args, "probe_lds_ring", False))
    tdt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[prec]
    sets = []
    for k in range(2):
        depth = synth.uniform01(f"probe/depth{k}", B * 256 * 256, 0).reshape(B, 256, 256, 1)
        ins = synth.uniform01(f"probe/ins{k}", B * L * 768, 0).reshape(B, L, 768) * 2 - 1
        sets.append((torch.from_numpy(depth).cuda(), torch.from_numpy(ins).to(tdt).cuda()))
    state = {"tick": 0}

    def step(mask=None):
        d, i = sets[state["tick"] & 1]
        state["tick"] += 1
        return probe.forward(d, i)
    esz = 2 if prec == "fp16" else 4
    # depth f32 in + instruction tensor in + (B,L,256) out, weights once per batch
    alg_bytes = B * (256 * 256 * 4 + L * 768 * esz + L * 256 * esz) + (3.26e6 + 1.02e6) * esz
    return cfg, B, probe, step, alg_bytes, prec


def _respawn(args):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: become the launcher -- re-execute this script under
    torch.distributed.run with one rank per GPU on 127.0.0.1 and pass its exit code (and rank 0's ONE JSON line on stdout) through."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=1, choices=[0, 1, 3, 4], help="BASELINE.json configs index (1 = the headline workload)")
    ap.add_argument("--no-kernel-probe", action="store_true", help="skip the dominant-kernel timing (profiler runs: keeps the trace to the step's own launches)")
    ap.add_argument("--prewarm", type=int, default=40, help="untimed clock-ramp steps before the W warm-up steps (0 for profiler runs)")
    ap.add_argument("--sustain", type=float, default=5.0, help="seconds of the additional sustained measurement reported as `sustained` (0 = skip)")
    ap.add_argument("--batch", type=int, default=0, help="environments per GPU (default: the BASELINE size of the config)")
    ap.add_argument("--total-batch", type=int, default=0, help="strong scaling as the PRIMARY value: total environments, split evenly over the ranks; "
                                                               "default 0 = weak scaling at --batch per GPU (N > 1 then adds a `strong` leg at --strong-total)")
    ap.add_argument("--strong-total", type=int, default=512, help="N > 1: total environments of the additional strong-scaling leg reported as `strong` "
                                                                  "(BASELINE configs[2]: 512; 0 = skip)")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "bf16", "fp32"],
                    help="fp16 = the measured 16-bit mode (range-calibrated fp16 tiles); bf16 = bf16 tiles in BERT / RGB trunks / cross-modal block")
    ap.add_argument("--latency-leg", type=int, default=1, help="N = 1, configs[1]: also report the synchronous latency of a single-environment (B = 1) step, "
                    "the reference's own evaluation loop, for both graph replay forms (0 = skip)")
    ap.add_argument("--chain-graphs", choices=("auto", "0", "1"), default=None, help="engine option chain_graphs for the timed legs: one linear hipGraph per chain "
                    "(lower host cost and B = 1 latency, 2-4 %% less pipelined throughput); default 0 = the single forked graph, the measured configuration -- "
                    "except with --h2d, where the default is the engine's own choice (auto: the frame copies overlap BERT in that form)")
    ap.add_argument("--bf16-leg", type=float, default=2.0, help="N = 1, configs[1], --precision fp16: seconds of an additional run of the SAME workload on a "
                                                                 "`precision=\"bf16\"` engine, reported as `bf16_mode` (0 = skip)")
    ap.add_argument("--h2d-leg", type=float, default=2.0, help="N = 1, configs[1]: seconds of an additional PCIe-inclusive run of the SAME workload (pinned host "
                    "frames copied inside every step) reported as `h2d` (0 = skip)")
    ap.add_argument("--torch-gather", action="store_true", help="N > 1: the per-step all-gather through torch.distributed instead of the library's own "
                                                                "RCCL call behind the step (A/B of rounds 1-2 vs round 3)")
    ap.add_argument("--reuse-instruction", action="store_true",
                    help="NOT the headline configuration: steps after the first skip BERT (instructions unchanged; hcm_act_ex flag)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batches", action="store_true", help="additionally time the CPU oracle at batch 4, 16 and 64, best of 5 steps each "
                                                               "(BASELINE.md section 3's protocol; minutes of CPU work, not in the default run)")
    ap.add_argument("--h2d", action="store_true", help="include the per-step host->device staging of uint8 RGB + f32 depth "
                    "(pinned buffers) in the timed region: the PCIe-inclusive rate quoted in DESIGN.md, never the headline value")
    ap.add_argument("--h2d-prestage", action="store_true", help="with --h2d: copy the frames in front of the step on the caller's stream instead of "
                    "handing the pinned host frames to the library (HCM_ACT_HOST_FRAMES: one copy per encoder chain inside the step)")
    ap.add_argument("--no-graph", action="store_true", help="enqueue the kernels of a step eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--probe-lds-ring", action="store_true", help="configs[3] A/B aid: the cross-modal layer with its weights through the LDS ring (hcm_op_vla_layer) instead of fragment-order weights read into registers")
    ap.add_argument("--configs-leg", type=float, default=1.5, help="N = 1 default run: seconds per configuration of the bounded `configs` block (BASELINE configs[0], [3], [4] "
                    "at their own batch sizes: value, ms_per_step, roofline.frac); 0 = skip")
    ap.add_argument("--host-procs-leg", type=int, default=8, help="N = 1 default run: processes of the `host_8proc` leg (tools/host_procs.py: N processes pinned to 2 cores each, "
                    "every one driving its own engine through the real graph-replay call on the shared GPU: host time inside act()); 0 = skip")
    ap.add_argument("--gather-leg", type=float, default=1.0, help="N = 1 default run: seconds per side of `gather_world1` -- the same step with the library's collective "
                    "(hcm_act_gather through real librccl on a one-rank communicator) against the plain step; 0 = skip")
    args = ap.parse_args()
    if args.chain_graphs is None:
        args.chain_graphs = "auto" if args.h2d else "0"
    if args.gpus > 1 and "RANK" not in os.environ:
        if args.config != 1:
            raise SystemExit("--config 0/3/4 are single-GPU roofline lines; the multi-GPU workload is configs[1]/[2]")
        _respawn(args)
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner when its communicator
    # is created): keep the real stdout aside for the result line and point file descriptor 1 at stderr for everything else.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    if os.environ.get("HCM_SKIP"):
        # a development build of the library (make DEV=1) drops whole encoder chains under this variable: whatever it measures is
        # not the workload, so no result line is printed
        print("bench.py: HCM_SKIP is set -- a work-dropping profiling knob; refusing to produce a benchmark line", file=sys.stderr)
        raise SystemExit(3)

    import torch
    import torch.distributed as dist
    import hcm_pkg
    hcm_pkg.load()
    from robo_vln_amd.rollout import shard_range, gather_records

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): use `python bench.py --gpus N` (spawns its own ranks) or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if world > 1 and args.config != 1:
        raise SystemExit("--config 0/3/4 are single-GPU roofline lines; the multi-GPU workload is configs[1]/[2]")
    dev = _device(local_rank)
    if not STUB:
        torch.cuda.set_device(local_rank)
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ     # launched by torch.distributed.run (any N)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if STUB:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    alg_bytes = None
    make_step = None
    # N > 1 weak-scaling runs also time a strong-scaling leg (the same rollout of --strong-total environments split over the ranks; at N = 8 with the
    # default 512 that is the weak leg's 64 per GPU again = BASELINE configs[2])
    strong_B = 0
    if world > 1 and args.config == 1 and not args.total_batch and args.strong_total:
        if args.strong_total % world:
            raise SystemExit("--strong-total must be a multiple of the number of ranks")
        strong_B = args.strong_total // world
    if args.config == 3:
        cfg, B, eng, raw_step, alg_bytes, prec3 = build_probe_workload(args)
        weights = None
    else:
        cfg, B, eng, raw_step, weights, make_step = build_act_workload(args, args.config, rank, world, local_rank, strong_batch=strong_B)
        if STUB and strong_B:
            strong_B = min(strong_B, 2 * B)

    # N>1: the all-gather of step i can either sit on the critical path (what a rollout needs: the environments cannot produce
    # observation i+1 before they have record i) or be overlapped with the compute of step i+1 on a communication stream (an upper
    # bound that only a pipelined simulator could use).  `value` is the former; the latter is reported as an extra key.
    comm = torch.cuda.Stream() if use_dist and not STUB else None
    # the collective is enqueued by the LIBRARY behind the graph replay (hcm_act_gather, one ncclAllGather on the step's stream) unless
    # --torch-gather asks for the torch.distributed call per step of rounds 1-2 (A/B)
    lib_gather = use_dist and args.config == 1 and not args.torch_gather
    gather_note = None
    if lib_gather:
        try:
            eng.comm_init()             # raises on EVERY rank or on none (policy.py: status byte with the id, MIN-reduce of the init result)
        except RuntimeError as e:       # the library's own communicator is an optimisation: fall back to the torch.distributed collective
            lib_gather = False
            gather_note = f"hcm_comm_init failed ({e}); torch.distributed.all_gather_into_tensor per step instead"
            if rank == 0:
                print("bench.py: " + gather_note, file=sys.stderr)

    def run_leg(B, raw_step, steps, sustain=0.0, with_overlap=False):
        """One measured leg at B environments per rank: rank-tagged gather check, untimed warm-up, K timed steps between barriers (MAX over ranks),
        validation of what the step returned.  -> dict(dt, host_us, ranks_seen, sustained, overlapped)"""
        global_B = B * world
        lo_e, hi_e = shard_range(global_B, world, rank)     # this rank's contiguous block of environments e -> rank e // B
        all_rec = torch.empty(global_B, 7, device=dev) if use_dist else None
        last = {}
        pending = []

        def step(mask=None, overlap=False):
            if use_dist and overlap and len(pending) >= 2:
                torch.cuda.current_stream().wait_event(pending.pop(0))
            if lib_gather and not overlap:
                full = raw_step(mask, gather=True)          # ONE RCCL all-gather of the (B,7) records per step, in stream order, issued by libhcm
                last["full"] = full
                last["r"] = full[lo_e:hi_e]
                return
            r = raw_step(mask)
            last["r"] = r
            last["full"] = all_rec
            if not use_dist:
                return
            if not overlap:
                gather_records(r, all_rec)                  # ONE all-gather of the (B,7) records per step, in stream order
                return
            ready = torch.cuda.Event()
            ready.record()
            comm.wait_event(ready)
            with torch.cuda.stream(comm):
                gather_records(r, all_rec)
                done = torch.cuda.Event()
                done.record(comm)
            pending.append(done)

        def timed(n, overlap=False):
            if use_dist:
                dist.barrier()
            _sync()
            t0 = time.perf_counter()
            for _ in range(n):
                step(overlap=overlap)
            last["host_s"] = (time.perf_counter() - t0) / n          # host time to ENQUEUE a step (no synchronisation inside)
            _sync()
            if use_dist:
                dist.barrier()
            dt = time.perf_counter() - t0
            if use_dist:
                t = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            return dt

        ranks_seen = [0]
        if use_dist:
            # prove that the collective moves every rank's rows to every rank: rank-tagged records, checked on all ranks
            tag = (torch.arange(B * 7, device=dev, dtype=torch.float32).reshape(B, 7) + 1000.0 * rank)
            gather_records(tag, all_rec)
            _sync()
            want = torch.cat([torch.arange(B * 7, dtype=torch.float32).reshape(B, 7) + 1000.0 * r for r in range(world)]).to(dev)
            assert torch.equal(all_rec, want), f"rank {rank}: all-gather did not deliver every rank's rows in rank order"
            ranks_seen = sorted({int(v) for v in (all_rec[::B, 0] // 1000.0).tolist()})
            assert ranks_seen == list(range(world)), ranks_seen

        # two untimed steps before the W warm-up steps: the library runs a new (batch, pointer set) eagerly once and captures its
        # hipGraph on the second call -- neither belongs in anybody's timed region, whatever W is (x2: two observation sets)
        zero = torch.zeros(B, device=dev)
        step(zero)
        for _ in range(5):
            step()
        for _ in range(args.prewarm):                      # ~0.25 s of untimed steps so that the clocks have ramped (a fixed
            step()                                         # count: every rank must issue the same number of all-gathers)
        for _ in range(args.warmup):
            step()
        dt = timed(steps)
        res = {"B": B, "global_B": global_B, "dt": dt, "steps": steps, "host_us": last["host_s"] * 1e6, "ranks_seen": ranks_seen,
               "sustained": None, "overlapped": None}
        # what the step returned is checked, not only timed: finite, and (N>1) this rank's rows of the gathered records are its own
        r = last["r"]
        assert torch.isfinite(r.float()).all(), "non-finite outputs"
        if use_dist:
            full = last["full"]
            assert full.shape == (global_B, 7) and torch.equal(full[lo_e:hi_e], r), "gathered records do not contain this rank's rows"
            chk = full.double().sum().reshape(1)
            lo_c, hi_c = chk.clone(), chk.clone()
            dist.all_reduce(lo_c, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi_c, op=dist.ReduceOp.MAX)
            assert float(lo_c) == float(hi_c), "ranks disagree on the gathered records"
        if sustain > 0:
            n_s = max(steps, int(sustain / (dt / steps)) + 1)
            if use_dist:
                t = torch.tensor([n_s], device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                n_s = int(t.item())
            dts = timed(n_s)
            res["sustained"] = {"seconds": round(dts, 3), "steps": n_s, "value": round(global_B * n_s / dts, 2), "ms_per_step": round(dts / n_s * 1e3, 3)}
        if with_overlap and comm is not None:
            for _ in range(4):
                step(overlap=True)
            dto = timed(steps, overlap=True)
            torch.cuda.current_stream().wait_stream(comm)
            res["overlapped"] = {"value": round(global_B * steps / dto, 2), "ms_per_step": round(dto / steps * 1e3, 3),
                                 "note": "all-gather of step i on a communication stream, overlapping the compute of step i+1 (needs a pipelined simulator)"}
        return res

    leg = run_leg(B, raw_step, args.steps, sustain=args.sustain, with_overlap=use_dist)
    global_B, dt, host_us = leg["global_B"], leg["dt"], leg["host_us"]
    sustained, overlapped = leg["sustained"], leg["overlapped"]
    strong = None
    if strong_B and make_step is not None:
        if strong_B == B:
            strong = {"total_batch": global_B, "per_gpu_batch": B, "value": round(global_B * args.steps / dt, 2), "ms_per_step": round(dt / args.steps * 1e3, 3),
                      "note": "identical to the weak leg at this N (total / N = the weak per-GPU batch): the same timed region"}
        else:
            sl = run_leg(strong_B, make_step(strong_B), args.steps)
            strong = {"total_batch": sl["global_B"], "per_gpu_batch": strong_B, "value": round(sl["global_B"] * args.steps / sl["dt"], 2),
                      "ms_per_step": round(sl["dt"] / args.steps * 1e3, 3), "host_us_per_step": round(sl["host_us"], 1), "ranks_seen": sl["ranks_seen"]}
    # the "bf16" mode of the SAME workload on a second engine (N = 1 default run only): driver-visible throughput of the mode whose parity
    # tests/test_parity_gpu.py asserts at 1e-2
    bf16_mode = None
    if (world == 1 and args.config == 1 and args.precision == "fp16" and args.bf16_leg > 0 and not STUB and not args.h2d and not args.reuse_instruction
            and not args.batch and not args.total_batch):
        try:
            import copy
            a2 = copy.copy(args)
            a2.precision = "bf16"
            _, B2, eng2, step2, _, _ = build_act_workload(a2, 1, rank, world, local_rank)
            n2 = max(10, int(args.bf16_leg / (dt / args.steps)))
            l2 = run_leg(B2, step2, n2)
            bf16_mode = {"value": round(l2["global_B"] * n2 / l2["dt"], 2), "ms_per_step": round(l2["dt"] / n2 * 1e3, 3), "steps": n2,
                         "dtype": dtype_string(a2, eng2), "note": "same workload, same timed-region protocol, on an engine built with precision=\"bf16\""}
            eng2.close()
            if not args.no_kernel_probe:
                # the mode's own roofline block: the same dominant launch (BERT FFN1 on gemm256f_kernel) on bf16 tiles, timed live like the fp16 one
                dk2 = dominant_kernel_probe(B2, cfg.instr_len, dtype="bf16")
                v2 = bf16_mode["value"] * GFLOP[1] / 1e3
                bf16_mode["roofline"] = {"bound": "mfma", "achieved": dk2["ffn1"]["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                         "frac": dk2["ffn1"]["frac_of_peak"], "traffic": None,
                                         "scope": "dominant kernel: gemm256f_kernel<bf16> on BERT FFN1 (" + dk2["ffn1"]["shape"] + f"), {dk2['ffn1']['us_per_launch']} us per launch",
                                         "bert_gemms": dk2,
                                         "whole_step": {"achieved": round(v2, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(v2 / PEAK_BF16_TFLOPS, 4),
                                                        "basis": f"{GFLOP[1]} algorithmic GFLOP per env-step x env-steps/s of this mode"}}
        except Exception as e:           # never lose the headline number to the extra leg
            bf16_mode = {"error": str(e)}

    # PCIe-inclusive leg (N = 1 default run only; SURVEY 8d: "report separately with H->D of uint8 RGB + f32 depth included"): the SAME workload with the
    # frames handed over as pinned HOST tensors every step (HCM_ACT_HOST_FRAMES: the library copies each frame set at the head of the chain that
    # reads it, per-chain linear graphs).  Never `value`.
    h2d_leg = None
    if (world == 1 and args.config == 1 and args.precision == "fp16" and args.h2d_leg > 0 and not STUB and not args.h2d and not args.reuse_instruction
            and not args.batch and not args.total_batch and not args.no_graph):
        try:
            import copy
            a4 = copy.copy(args)
            a4.h2d, a4.chain_graphs = True, "auto"
            _, B4, eng4, step4, _, _ = build_act_workload(a4, 1, rank, world, local_rank)
            n4 = max(10, int(args.h2d_leg / (dt / args.steps)))
            l4 = run_leg(B4, step4, n4)
            per = B4 * (cfg.rgb_hw * cfg.rgb_hw * 3 + cfg.depth_hw * cfg.depth_hw * 4)
            h2d_leg = {"value": round(l4["global_B"] * n4 / l4["dt"], 2), "ms_per_step": round(l4["dt"] / n4 * 1e3, 3), "steps": n4, "bytes_per_step": per,
                       "pcie_GBps": round(per / (l4["dt"] / n4) / 1e9, 2),
                       "note": "same workload with uint8 RGB + f32 depth copied from pinned host memory inside every timed step (act(host_frames=True), "
                               "chain_graphs=auto); instruction ids resident (they change per episode, not per step)"}
            eng4.close()
        except Exception as e:           # never lose the headline number to the extra leg
            h2d_leg = {"error": str(e)}

    # single-environment latency (N = 1 default run only): the reference's evaluation loop (hierarchical_trainer.py:1088-1107) calls the policy once per
    # simulator step and needs the action before it can step the simulator, so what it sees is the SYNCHRONOUS latency of a B = 1 step, not a pipelined
    # rate.  Both replay forms of the library: one hipGraph captured across the forked streams, and one linear graph per chain (HCM_ACT_CHAIN_GRAPHS).
    single_env = None
    if (world == 1 and args.config == 1 and args.precision == "fp16" and args.latency_leg and not STUB and not args.h2d and not args.reuse_instruction
            and not args.batch and not args.total_batch and not args.no_graph):
        try:
            import copy
            import torch
            a3 = copy.copy(args)
            a3.batch = 1
            _, _, eng3, step3, _, _ = build_act_workload(a3, 1, rank, world, local_rank)
            single_env = {"batch": 1, "steps": 200, "protocol": "act() + synchronise per step; host_us = time inside act() (argument marshalling + graph replay)"}
            for name, mode in (("forked_graph", False), ("chain_graphs", True)):
                eng3._chain_graphs = mode
                for _ in range(20):
                    step3()
                _sync()
                host, t0 = 0.0, time.perf_counter()
                for _ in range(200):
                    h0 = time.perf_counter()
                    step3()
                    host += time.perf_counter() - h0
                    _sync()
                single_env[name] = {"ms_per_step": round((time.perf_counter() - t0) / 200 * 1e3, 3), "host_us_per_step": round(host / 200 * 1e6, 1)}
            eng3.close()
        except Exception as e:           # never lose the headline number to the extra leg
            single_env = {"error": str(e)}

    default_run = (world == 1 and args.config == 1 and args.precision == "fp16" and not STUB and not args.h2d and not args.reuse_instruction
                   and not args.batch and not args.total_batch and not args.no_graph)
    # What the step's collective costs, measurable on ONE GPU (SCALE_r* has never had a node): the same timed region with the record going through
    # hcm_act_gather -- real librccl, a one-rank communicator, ncclAllGather enqueued by the library behind the graph replay -- against the plain
    # hcm_act_ex step, interleaved A/B/A/B on the SAME engine.  A first multi-GPU run can lose to this (and to the host side, `host_us_per_step`).
    gather_w1 = None
    if default_run and args.gather_leg > 0:
        try:
            created = not dist.is_initialized()
            if created:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
            eng.comm_init()
            n5 = max(20, int(args.gather_leg / (dt / args.steps)))

            def span(gather):
                for _ in range(4):
                    raw_step(None, gather=gather)
                _sync()
                t0 = time.perf_counter()
                for _ in range(n5):
                    raw_step(None, gather=gather)
                host = (time.perf_counter() - t0) / n5
                _sync()
                return (time.perf_counter() - t0) / n5, host
            plain, gath = [], []
            for _ in range(3):
                plain.append(span(False)); gath.append(span(True))
            pm, gm = min(p[0] for p in plain), min(g[0] for g in gath)
            gather_w1 = {"value": round(B / gm, 2), "ms_per_step": round(gm * 1e3, 3), "plain_ms_per_step": round(pm * 1e3, 3),
                         "delta_us": round((gm - pm) * 1e6, 1), "host_us_per_step": round(min(g[1] for g in gath) * 1e6, 1),
                         "plain_host_us_per_step": round(min(p[1] for p in plain) * 1e6, 1), "steps_per_side": n5, "rounds": 3,
                         "note": "act(gather=True): hcm_act_gather = graph replay + ONE ncclAllGather of the (B,7) record through librccl on a one-rank communicator, "
                                 "against act() on the same engine, best of three interleaved rounds each; what the collective adds at world 1 (no wire time)"}
            eng.comm_abort()
            if created:
                dist.destroy_process_group()
        except Exception as e:           # never lose the headline number to the extra leg
            gather_w1 = {"error": str(e)}

    # The host side of an 8-rank node, as far as one GPU can show it: 8 processes on 2 cores each enqueueing real graph replays (tools/host_procs.py)
    host_procs = None
    if default_run and args.host_procs_leg > 0:
        try:
            import subprocess
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_procs.py"), str(args.host_procs_leg), "8", "2.0"], capture_output=True, timeout=240, cwd=ROOT)
            lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
            host_procs = json.loads(lines[-1]) if lines else {"error": "no output: " + r.stderr.decode()[-300:]}
        except Exception as e:           # never lose the headline number to the extra leg
            host_procs = {"error": str(e)}

    # The other BASELINE configurations in the driver-visible line (bounded: --configs-leg seconds each): configs[0] (the reference's own CPU-runnable
    # case), configs[3] (depth-only SimpleCNN + 1-layer block, B = 256), configs[4] (ResNet-50 + 6-layer decoder, L = 160, B = 128, high-level model).
    configs_block = None
    if default_run and args.configs_leg > 0:
        import copy
        configs_block = {}
        for idx in (0, 3, 4):
            try:
                a6 = copy.copy(args)
                a6.config, a6.batch = idx, 0
                if idx == 3:
                    _, B6, eng6, step6, alg6, _ = build_probe_workload(a6)
                else:
                    _, B6, eng6, step6, _, _ = build_act_workload(a6, idx, rank, world, local_rank)
                for _ in range(12):
                    step6()
                _sync()
                t0 = time.perf_counter()
                for _ in range(10):
                    step6()
                _sync()
                est = (time.perf_counter() - t0) / 10
                n6 = max(20, int(args.configs_leg / est))
                t0 = time.perf_counter()
                for _ in range(n6):
                    step6()
                _sync()
                ms6 = (time.perf_counter() - t0) / n6 * 1e3
                ent = {"workload": WORKLOAD[idx], "per_gpu_batch": B6, "steps": n6, "value": round(B6 / ms6 * 1e3, 2), "unit": "env-steps/s", "ms_per_step": round(ms6, 4)}
                if idx == 3:
                    tf6 = 0.2422 * B6 / ms6
                    ent["roofline"] = {"bound": "mfma", "achieved": round(tf6, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf6 / PEAK_BF16_TFLOPS, 4),
                                       "hbm_view": {"achieved_TBps": round(alg6 / 1e9 / ms6, 4), "peak_TBps": PEAK_HBM_TBPS, "frac": round(alg6 / 1e9 / ms6 / PEAK_HBM_TBPS, 4)}}
                else:
                    tf6 = B6 / ms6 * 1e3 * GFLOP[idx] / 1e3
                    ent["roofline"] = {"bound": "mfma", "achieved": round(tf6, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf6 / PEAK_BF16_TFLOPS, 4),
                                       "basis": f"{GFLOP[idx]} algorithmic GFLOP per env-step x env-steps/s (whole step)"}
                configs_block[f"cfg{idx}"] = ent
                if hasattr(eng6, "close"):
                    eng6.close()
                del eng6, step6
                torch.cuda.empty_cache()
            except Exception as e:       # never lose the headline number to the extra legs
                configs_block[f"cfg{idx}"] = {"error": str(e)}

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = global_B * args.steps / dt
        tr = pmc_traffic(args.config) if B == BATCH[args.config] and not args.total_batch and not STUB else None
        out = {
            "metric": "policy env-steps/sec (batched act()) at 256x256 RGB-D, 80-tok instr" if args.config == 1 else
                      f"policy env-steps/sec, BASELINE.json configs[{args.config}]",
            "value": round(value, 2), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong" if args.total_batch else "weak", "vs_baseline": None,
            "dtype": dtype_string(args, eng) if args.config != 3 else (prec3 + " storage / MFMA, fp32 accumulate"),
            "data": "synthetic (random-init weights, random RGB-D frames and token ids, two observation sets resident in HBM used alternately)",
            "h2d_in_timed_region": bool(args.h2d),
            "config": {"workload": WORKLOAD[args.config], "per_gpu_batch": B, "global_batch": global_B,
                       "parallelism": (f"env-sharded data parallel x{world}, one all-gather of (B,7) records per step on the critical path"
                                       if world > 1 else "single GPU")},
            "host_us_per_step": round(host_us, 1),
        }
        if STUB:
            out["metric"] = "STUB ENGINE -- launch-plumbing test on CPU, not a measurement"
            out["data"] = "HCM_BENCH_STUB=1: stand-in engine, gloo, no GPU"
        if use_dist:
            out["ranks_seen"] = leg["ranks_seen"]
            out["config"]["all_gather"] = ("ncclAllGather enqueued by libhcm on the step's stream behind the hipGraph replay (hcm_act_gather)" if lib_gather
                                           else gather_note or "torch.distributed.all_gather_into_tensor per step (--torch-gather)")
            out["weak"] = {"per_gpu_batch": B, "global_batch": global_B, "value": round(value, 2), "ms_per_step": round(ms, 3)} if not args.total_batch else None
        if strong:
            out["strong"] = strong
        if sustained:
            out["sustained"] = sustained
        if overlapped:
            out["overlapped_all_gather"] = overlapped
        if bf16_mode:
            out["bf16_mode"] = bf16_mode
        if h2d_leg:
            out["h2d"] = h2d_leg
        if single_env is not None:
            out["single_env_latency"] = single_env
        if gather_w1 is not None:
            out["gather_world1"] = gather_w1
        if host_procs is not None:
            out["host_8proc"] = host_procs
        if configs_block is not None:
            out["configs"] = configs_block
        if args.config == 3:
            gb = alg_bytes / 1e9
            # BASELINE.json calls this configuration memory-bound; at 16-bit MFMA rates it is not (0.25 GFLOP and 0.46 MB per sample = 540 FLOP/B,
            # above the 312 FLOP/B ridge): the line is judged against the MATRIX peak, the HBM view rides along as a sub-key
            gflop3 = 0.2422 * B                                   # conv0 + conv1 + conv2 + FC + one cross-modal layer per sample
            tf = gflop3 / ms                                      # GFLOP / ms = TFLOP/s
            out["roofline"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4),
                               "traffic": (tr or {}).get("step_GB"), "traffic_unit": "GB per step (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/pmc_traffic_cfg3.json)" if tr else None,
                               "basis": f"{gflop3:.1f} algorithmic GFLOP per step / step time; the step is the whole probe (all its launches)",
                               "hbm_view": {"achieved_TBps": round(gb / ms, 4), "peak_TBps": PEAK_HBM_TBPS, "frac": round(gb / ms / PEAK_HBM_TBPS, 4),
                                            "basis": f"{gb * 1e3:.1f} MB algorithmic bytes per step (depth f32 in + (B,80,768) instruction tensor in + (B,80,256) out + "
                                                     "weights once, SURVEY 8d)"}}
        else:
            gf = GFLOP[args.config]
            achieved = value * gf / 1e3                           # TFLOP/s, algorithmic
            whole = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS * world, "unit": "TFLOP/s",
                     "frac": round(achieved / (PEAK_BF16_TFLOPS * world), 4),
                     "basis": f"{gf} algorithmic GFLOP per env-step (SURVEY 8a) x env-steps/s"}
            if tr:
                whole["traffic_GB_per_step"] = tr.get("step_GB")
                whole["hbm_view"] = {"achieved_TBps": round(tr["step_GB"] / ms, 3), "peak_TBps": PEAK_HBM_TBPS * world,
                                     "frac": round(tr["step_GB"] / ms / PEAK_HBM_TBPS, 4)}
            if args.reuse_instruction:
                out["metric"] += " [instruction stream cached: BERT skipped, 23.1 instead of 36.9 GFLOP per env-step executed]"
                whole["achieved"] = round(value * (gf - 13.83) / 1e3, 2)
                whole["frac"] = round(whole["achieved"] / (PEAK_BF16_TFLOPS * world), 4)
                whole["basis"] = "cached-instruction variant: 36.9 - 13.83 (BERT) GFLOP per env-step x env-steps/s"
            roof = dict(whole)
            roof["scope"] = "whole step"
            if args.precision != "fp32" and not args.no_kernel_probe and args.config in (1, 4) and not STUB:
                try:
                    L = cfg.instr_len
                    dk = dominant_kernel_probe(B, L)
                    top = dk["ffn1"]
                    k_tr = (tr or {}).get("dominant_kernel_GB_per_launch")
                    # the JSON line's `roofline` is the DOMINANT KERNEL (gemm256f_kernel on the BERT FFN1 shape), live HIP-event timing;
                    # the whole-step view and the HBM-bound probe ride along as labelled sub-objects
                    roof = {"bound": "mfma", "achieved": top["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": top["frac_of_peak"],
                            "traffic": k_tr, "traffic_unit": "GB per launch, averaged over this kernel's launches of the step (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/pmc_traffic*.json)" if k_tr else None,
                            "scope": "dominant kernel: gemm256f_kernel<f16> on BERT FFN1 (" + top["shape"] + f", GELU epilogue), {top['us_per_launch']} us per launch, "
                                     f"{top['gflop_per_launch']} algorithmic GFLOP per launch, 12 launches per step",
                            "bert_gemms": dk, "whole_step": whole}
                    roof["hbm_probe"] = hbm_kernel_probe(B)
                except Exception as e:       # never lose the headline number to the probe
                    roof["dominant_kernel_error"] = str(e)
            out["roofline"] = roof
        if args.config == 3:
            out["config"]["hipgraph"] = {"enabled": not args.no_graph, "inputs": "read in place from two alternating device buffer sets"}
        if hasattr(eng, "query"):
            out["config"]["hipgraph"] = {"enabled": not args.no_graph and args.config in (0, 1), "chain_graphs": args.chain_graphs, "graph_steps": eng.query(7), "eager_steps": eng.query(8)}
        if not args.no_cpu_baseline and world == 1 and args.config == 1 and not STUB:          # reported on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(cfg, *weights)
            out["cpu_baseline"]["batches"] = cpu_baseline_protocol(cfg, *weights, budget_s=None if args.cpu_batches else 30.0)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if hasattr(eng, "close"):
        eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
