"""In-step tile / staging sweep for single GEMM shapes (development build): for every (shape, choice) pair, one default bench.py run with
HCM_IGEMM_SHAPE_FORCE="M,N,K:choice" -- the step is the judge, not the stand-alone launch (a launch tuned alone at one workgroup per CU returns
little in the time-shared step).  A base run is interleaved every `--base-every` candidates; prints a table per shape.
usage (GPU box): HCM_DEV_LIB=1 python tools/shape_sweep_instep.py --shapes 65536,512,256 16384,1024,512 --choices 6 12 24 30 36 --out gpurun_out/x.json"""
import argparse, json, os, subprocess, sys

A = "--steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --bf16-leg 0 --latency-leg 0 --h2d-leg 0 --no-kernel-probe --configs-leg 0 --gather-leg 0 --host-procs-leg 0".split()


def run(env_extra):
    env = dict(os.environ, HCM_DEV_LIB="1", **env_extra)
    try:
        out = subprocess.run([sys.executable, "bench.py"] + A, env=env, capture_output=True, text=True, timeout=300)
        return json.loads(out.stdout.strip().splitlines()[-1])["value"]
    except Exception as e:                                   # a choice the shape cannot take (tile too big, no instantiation): reported, skipped
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="+", required=True)
    ap.add_argument("--choices", nargs="+", type=int, required=True)
    ap.add_argument("--base-every", type=int, default=5)
    ap.add_argument("--out", default="gpurun_out/shape_sweep_instep.json")
    a = ap.parse_args()
    res = {"base": [], "shapes": {}}
    n = 0
    for sh in a.shapes:
        res["shapes"][sh] = {}
        for c in a.choices:
            if n % a.base_every == 0:
                res["base"].append(run({}))
            n += 1
            res["shapes"][sh][str(c)] = run({"HCM_IGEMM_SHAPE_FORCE": f"{sh}:{c}"})
        json.dump(res, open(a.out, "w"), indent=1)
    res["base"].append(run({}))
    json.dump(res, open(a.out, "w"), indent=1)
    base = [b for b in res["base"] if b]
    b0 = sum(base) / len(base)
    print(f"base: {min(base):.0f} .. {max(base):.0f} (mean {b0:.0f}, {len(base)} runs)")
    for sh, d in res["shapes"].items():
        print(sh, " ".join(f"{c}:{(v / b0 - 1) * 100:+.2f}%" if v else f"{c}:--" for c, v in d.items()))


if __name__ == "__main__":
    main()
