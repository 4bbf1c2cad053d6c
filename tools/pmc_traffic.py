"""Sum FETCH_SIZE / WRITE_SIZE (KB) over all kernels of a rocprofv3 --pmc csv and report per-step HBM-side traffic.
usage: pmc_traffic.py fetch.csv write.csv n_steps
FETCH_SIZE on gfx950 reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): the
corrected read figure doubles it; WRITE_SIZE is taken as reported."""
import csv, sys, collections
fetch, write, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
def tot(path, name):
    per = collections.Counter(); t = 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            v = float(r["Counter_Value"]); t += v
            per[r["Kernel_Name"].split("(")[0][:60]] += v
    return t, per
f, pf = tot(fetch, "FETCH_SIZE"); w, pw = tot(write, "WRITE_SIZE")
print(f"# HBM-side traffic per act() step (B=64), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, {steps} steps in the run\n")
print(f"FETCH_SIZE sum {f/1e6:.3f} GB raw -> {2*f/1e6/steps:.3f} GB/step corrected (x2);  WRITE_SIZE sum {w/1e6:.3f} GB -> {w/1e6/steps:.3f} GB/step")
print(f"total corrected traffic {(2*f+w)/1e6/steps:.3f} GB/step\n")
print("| kernel | fetch GB/step (x2 corrected) | write GB/step |\n|---|---|---|")
for k in sorted(set(pf)|set(pw), key=lambda k: -(2*pf[k]+pw[k]))[:14]:
    print(f"| `{k}` | {2*pf[k]/1e6/steps:.3f} | {pw[k]/1e6/steps:.3f} |")
