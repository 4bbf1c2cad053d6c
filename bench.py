"""bench.py -- policy env-steps/sec of the batched HCM act() on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]/[2]): per GPU 64 environments, 256x256 RGB-D frames, 80-token
instruction, full HCM model (2x ResNet-50 RGB, 2x GN-ResNet-50 depth, BERT-base, cross-modal block,
2x LSTM + heads), bf16 storage / fp32 accumulate, random-init weights, synthetic inputs resident in HBM.
One step = one fused act() (hi -> argmax -> lo) over the rank's 64 environments, followed (N>1) by ONE
RCCL all-gather of the (64,7) action records.  Weak scaling: global batch = 64 * N.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_STEP = 36.9          # SURVEY.md 8a / BASELINE.md section 2: config 2/3, 2xMAC conv+matmul+attention+RNN
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PER_GPU_BATCH = 64


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


def cpu_baseline(cfg, hi_sd, lo_sd, batch=16, steps=5):
    """The CPU oracle (kind 'port': torch-CPU fp32 restatement validated against the imported reference)
    timed on this host's cores on a bounded sample of the same workload."""
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
            "sample": f"{steps} act() steps at batch {batch}, 256x256 RGB-D, L=80, fp32, torch {torch.__version__} "
                      f"with {torch.get_num_threads()} threads"}


def dominant_kernel_probe(batch):
    """The launch type with the largest share of the step is the fused bottleneck kernel of the RGB ResNet-50 pair, `bneck231_kernel`
    (3x3 conv 64->64 + ReLU, 1x1 expansion 64->256 + identity + ReLU, and the next block's 1x1 reduction 256->64 + ReLU in one
    launch; six launches per step in four shapes).  Time its layer1 middle-block shape live with HIP events (torch events on the
    launch stream).  In the step that layer is ONE launch over the hi|lo pair (2 groups x B images); the probe runs the same amount
    of work as a single group over 2*B images (same tile count, same per-tile work), so its duration is comparable with the
    bneck231_kernel<bf16,128,64,64,0> row of profiles/r1_kernel_trace_bench.md.  The launch is HBM-bound: algorithmic bytes =
    2 B/elem * M * (64 in + 256 identity + 256 out + 64 next-reduction out); algorithmic FLOPs = 2 * M * (576*64 + 64*256 + 256*64)."""
    import ctypes as C
    import torch
    from robo_vln_amd import _lib
    lib = _lib.lib()
    B, H, W, C1, CN = 2 * batch, 64, 64, 64, 64
    C3 = 4 * C1
    bf = torch.bfloat16
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
                                             y.data_ptr(), w1.data_ptr(), b1.data_ptr(), o1.data_ptr(), _lib.HCM_BF16, B, H, W, C1, 1, CN, st)
        assert rc == 0
    for _ in range(100):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 100
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    M = B * H * W
    flops = 2.0 * M * (9 * C1 * C1 + C1 * C3 + C3 * CN)
    gbytes = 2.0 * M * (C1 + 2 * C3 + CN) / 1e9
    return {"kernel": "bneck231_kernel<bf16,128,64,64>: conv3x3 64->64 + conv1x1 64->256 + identity + next conv1x1 256->64 @64x64, hi|lo pair workload (M=2*B*4096)",
            "us_per_launch": round(ms * 1e3, 2), "bound": "hbm", "achieved_TBps": round(gbytes / ms, 3), "peak_TBps": 8.0,
            "frac": round(gbytes / ms / 8.0, 4), "tflops": round(flops / ms / 1e9, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-kernel-probe", action="store_true", help="skip the dominant-kernel timing (profiler runs: keeps the trace to the step's own launches)")
    ap.add_argument("--prewarm", type=int, default=40, help="untimed clock-ramp steps before the W warm-up steps (0 for profiler runs)")
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="environments per GPU")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--reuse-instruction", action="store_true",
                    help="NOT the headline configuration: steps after the first skip BERT (instructions unchanged; hcm_act_ex flag)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--h2d", action="store_true", help="include the per-step host->device staging of uint8 RGB + f32 depth "
                    "(pinned buffers) in the timed region: the PCIe-inclusive rate quoted in DESIGN.md, never the headline value")
    ap.add_argument("--no-graph", action="store_true", help="enqueue the ~700 kernels of a step eagerly instead of replaying the captured hipGraph")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import hcm_pkg
    hcm_pkg.load()
    from robo_vln_amd import synth
    from robo_vln_amd.config import baseline_config
    from robo_vln_amd.policy import HCMEngine
    from robo_vln_amd.rollout import shard_range, gather_records

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    torch.cuda.set_device(local_rank)
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ     # launched by torch.distributed.run (any N)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    cfg = baseline_config(1)
    B = args.batch
    global_B = B * world
    hi_sd, lo_sd = synth.make_weights(cfg, seed=0)            # full replica per rank
    eng = HCMEngine(cfg, hi_sd, lo_sd, max_batch=B, precision=args.precision, graph=not args.no_graph)
    # this rank's contiguous block of environments e -> rank e // B
    lo_e, hi_e = shard_range(global_B, world, rank)
    obs_np = synth.make_observations(cfg, B, step=rank, seed=0)    # distinct frames per rank, same shapes
    obs = {"rgb": torch.from_numpy(obs_np["rgb"]).cuda(), "depth": torch.from_numpy(obs_np["depth"]).cuda(),
           "instruction": torch.from_numpy(obs_np["instruction"]).cuda()}
    R = cfg.num_recurrent_layers
    hh = torch.zeros(R, B, cfg.hidden, device="cuda")
    lh = torch.zeros(R, B, cfg.hidden, device="cuda")
    mask0 = torch.zeros(B, device="cuda")
    mask1 = torch.ones(B, device="cuda")
    rec = torch.empty(B, 7, device="cuda")
    all_rec = torch.empty(global_B, 7, device="cuda") if use_dist else rec

    stager = None
    if args.h2d:
        from robo_vln_amd.obs import ObsStager
        stager = ObsStager(B, cfg.rgb_hw, cfg.depth_hw, cfg.instr_len, device=torch.device("cuda", local_rank))
        stager.host["rgb"].copy_(torch.from_numpy(obs_np["rgb"].astype("uint8")))
        stager.host["depth"].copy_(torch.from_numpy(obs_np["depth"]))
        stager.host["instruction"].copy_(torch.from_numpy(obs_np["instruction"].astype("int32")))

    # The all-gather of step i is enqueued on a communication stream and overlaps the compute of step i+1 (the gathered
    # records are the step's OUTPUT towards the environments; nothing in the next policy step reads them).  The engine's
    # record buffers ping-pong, so step i+2 waits for gather i before overwriting its source.
    comm = torch.cuda.Stream() if use_dist else None
    pending = []

    def step(mask):
        nonlocal hh, lh, obs
        if use_dist and len(pending) >= 2:
            torch.cuda.current_stream().wait_event(pending.pop(0))
        if stager is not None:
            for k in ("rgb", "depth"):
                stager.dev[k].copy_(stager.host[k], non_blocking=True)
            obs = stager.dev
        r, hh, lh = eng.act(obs, hh, lh, mask, reuse_instruction=args.reuse_instruction and mask is mask1)
        if use_dist:
            ready = torch.cuda.Event()
            ready.record()
            comm.wait_event(ready)
            with torch.cuda.stream(comm):
                gather_records(r, all_rec)              # ONE RCCL all-gather of the (B,7) records per step
                done = torch.cuda.Event()
                done.record(comm)
            pending.append(done)
        else:
            rec.copy_(r)

    # two untimed steps before the W warm-up steps: the library runs a new (batch, pointer set) eagerly once and captures its
    # hipGraph on the second call -- neither belongs in anybody's timed region, whatever W is
    step(mask0)
    step(mask1)
    for _ in range(args.prewarm):                      # ... and ~0.25 s of untimed steps so that the clocks have ramped (a fixed
        step(mask1)                                    # count: every rank must issue the same number of all-gathers)
    for _ in range(args.warmup):
        step(mask1)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(mask1)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert os.environ.get("HCM_SKIP") or torch.isfinite(all_rec).all()

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = global_B * args.steps / dt
        achieved = value * GFLOP_PER_STEP / 1e3               # TFLOP/s, algorithmic
        out = {
            "metric": "policy env-steps/sec (batched act()) at 256x256 RGB-D, 80-tok instr",
            "value": round(value, 2), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16+fp16 (16-bit MFMA, fp32 accumulate; DESIGN.md section 5)" if args.precision == "bf16" else "fp32", "data": "synthetic (random-init weights, random RGB-D frames and token ids, resident in HBM)",
            "h2d_in_timed_region": bool(args.h2d),
            "config": {"workload": "BASELINE.json configs[1]: full HCM act() (hi->argmax->lo), 256x256 RGB-D, L=80, VLA N=1, LSTM-512",
                       "per_gpu_batch": B, "global_batch": global_B,
                       "parallelism": f"env-sharded data parallel x{world}, one all-gather of (B,7) records per step" if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS * world, "unit": "TFLOP/s",
                         "frac": round(achieved / (PEAK_BF16_TFLOPS * world), 4),
                         # HBM-side bytes per act() step at B=64 from rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) +
                         # WRITE_SIZE, separate passes of this same command: profiles/r1_pmc_traffic_bench.md
                         "traffic": 15.07 if B == 64 else None, "traffic_unit": "GB per act() step at B=64 (whole step, like achieved)",
                         "traffic_source": "profiles/r1_pmc_traffic_bench.md",
                         "basis": f"{GFLOP_PER_STEP} algorithmic GFLOP per env-step (SURVEY 8a) x env-steps/s",
                         # the same step seen from the memory side: measured HBM bytes per step / step time vs the 8 TB/s peak
                         "hbm_view": ({"achieved_TBps": round(15.07 / ms, 3), "peak_TBps": 8.0 * world, "frac": round(15.07 / ms / 8.0, 4)}
                                      if B == 64 else None)},
        }
        if args.reuse_instruction:
            out["metric"] += " [instruction stream cached: BERT skipped, 23.1 instead of 36.9 GFLOP per env-step executed]"
            out["roofline"]["achieved"] = round(value * (GFLOP_PER_STEP - 13.83) / 1e3, 2)
            out["roofline"]["frac"] = round(out["roofline"]["achieved"] / (PEAK_BF16_TFLOPS * world), 4)
            out["roofline"]["basis"] = "cached-instruction variant: 36.9 - 13.83 (BERT) GFLOP per env-step x env-steps/s"
        out["config"]["hipgraph"] = {"enabled": not args.no_graph, "graph_steps": eng.query(7), "eager_steps": eng.query(8)}
        if args.precision == "bf16" and not args.no_kernel_probe:
            try:
                out["roofline"]["dominant_kernel"] = dominant_kernel_probe(B)
            except Exception as e:       # never lose the headline number to the probe
                out["roofline"]["dominant_kernel"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1:          # reported on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(cfg, hi_sd, lo_sd)
        print(json.dumps(out), flush=True)
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
