"""Launch-by-launch listing of ONE step of one encoder chain from a rocprofv3 --kernel-trace CSV: name, workgroups, duration, gap to the
previous kernel's end.  Meant for traces of tools/chain_step.py (dev library, HCM_SKIP mask), where a single chain runs serially.
usage: python tools/chain_seq.py p_kernel_trace.csv [marker_substring=pack_frame]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void ", "").replace("hcm::", "")
    g = lambda k: int(r.get(k + "_X", r.get(k, 1))) * int(r.get(k + "_Y", 1)) * int(r.get(k + "_Z", 1))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, g("Grid_Size") // max(1, g("Workgroup_Size"))))
rows.sort()
mark = sys.argv[2] if len(sys.argv) > 2 else "pack_frame"
idx = [i for i, r in enumerate(rows) if mark in r[2]]
a, b = idx[-2], idx[-1]
step = rows[a:b]
t0 = step[0][0]
print(f"{len(step)} launches, window {(step[-1][1] - t0) / 1e3:.1f} us, kernel time {sum(e - s for s, e, _, _ in step) / 1e3:.1f} us")
prev = t0
for s, e, n, wg in step:
    print(f"+{(s - t0) / 1e3:8.1f} gap {(s - prev) / 1e3:6.1f}  {(e - s) / 1e3:7.1f} us  wg {wg:6d}  {n[:110]}")
    prev = e
