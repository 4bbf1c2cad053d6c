"""Sum FETCH_SIZE / WRITE_SIZE (KB) over all kernels of a rocprofv3 --pmc csv and report per-step HBM-side traffic.
usage: pmc_traffic.py fetch.csv write.csv n_steps [out.json] > profiles/rN_pmc_traffic_bench.md
FETCH_SIZE on gfx950 reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): the
corrected read figure doubles it; WRITE_SIZE is taken as reported.  out.json (profiles/pmc_traffic.json) is what bench.py
reads for `roofline.traffic`: the whole-step figure and the per-launch figure of the dominant kernel (gemm256_kernel<f16>: its FFN1 and
QKV launches cannot be told apart in a counter dump, so the per-launch figure is the instantiation's average over both)."""
import csv, sys, collections, json
fetch, write, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
out_json = sys.argv[4] if len(sys.argv) > 4 else None
def tot(path, name):
    per = collections.Counter(); cnt = collections.Counter(); t = 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            v = float(r["Counter_Value"]); t += v
            k = r["Kernel_Name"].split("(")[0][:60]
            per[k] += v; cnt[k] += 1
    return t, per, cnt
f, pf, cf = tot(fetch, "FETCH_SIZE"); w, pw, cw = tot(write, "WRITE_SIZE")
print(f"# HBM-side traffic per step of the profiled bench.py configuration, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, {steps} steps in the run\n")
print(f"FETCH_SIZE sum {f/1e6:.3f} GB raw -> {2*f/1e6/steps:.3f} GB/step corrected (x2);  WRITE_SIZE sum {w/1e6:.3f} GB -> {w/1e6/steps:.3f} GB/step")
print(f"total corrected traffic {(2*f+w)/1e6/steps:.3f} GB/step\n")
print("| kernel | launches/step | fetch GB/step (x2 corrected) | write GB/step | MB/launch |\n|---|---|---|---|---|")
keys = sorted(set(pf)|set(pw), key=lambda k: -(2*pf[k]+pw[k]))
for k in keys[:16]:
    n = max(cf[k], cw[k], 1)
    print(f"| `{k}` | {n/steps:.0f} | {2*pf[k]/1e6/steps:.3f} | {pw[k]/1e6/steps:.3f} | {(2*pf[k]+pw[k])/1e3/n:.1f} |")
if out_json:
    dom = [k for k in keys if "gemm256" in k and "hcm::f16" in k] or [k for k in keys if "igemm" in k and "hcm::f16" in k]
    d = dom[0] if dom else None
    json.dump({"step_GB": round((2*f+w)/1e6/steps, 3), "fetch_GB": round(2*f/1e6/steps, 3), "write_GB": round(w/1e6/steps, 3),
               "dominant_kernel": d, "dominant_kernel_GB_per_launch": round((2*pf[d]+pw[d])/1e6/max(cf[d], cw[d], 1), 4) if d else None,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `python bench.py --steps 4 --warmup 1 --prewarm 0 --sustain 0 "
                         "--no-cpu-baseline --no-kernel-probe`; FETCH_SIZE x2 (gfx950)", "steps_in_run": steps}, open(out_json, "w"), indent=1)
