"""Summarise a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): per-kernel launches / total / average, as markdown.
usage: python tools/ktrace_summary.py p_kernel_trace.csv STEPS > profiles/xxx.md"""
import collections
import csv
import sys

steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void ", "").replace("hcm::", "")
    if "copyBuffer" in n or n.startswith("at::") or "elementwise_kernel" in n:
        n = "(torch / runtime helper kernels: weight upload copies, synthetic-input generation)"
    if n.startswith("absmax_kernel"):
        n = "absmax_kernel (fp16 range calibration at engine construction: not part of a step)"
    t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    e = d.setdefault(n, [0, 0, 1 << 60, 0])
    e[0] += 1; e[1] += t; e[2] = min(e[2], t); e[3] = max(e[3], t)
rows = sorted(d.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(v[0] for _, v in rows)} launches; per step ({steps} steps) {tot / 1e6 / steps:.3f} ms\n")
print("| kernel | launches | total ms | % | avg us | min us | max us | ms/step |")
print("|---|---|---|---|---|---|---|---|")
for n, (c, t, mn, mx) in rows[:45]:
    if len(n) > 95:
        n = n[:95] + "..."
    print(f"| `{n}` | {c} | {t / 1e6:.3f} | {100 * t / tot:.1f} | {t / c / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {t / 1e6 / steps:.3f} |")
