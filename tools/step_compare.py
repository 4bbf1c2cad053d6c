"""In-step vs alone: two rocprofv3 --kernel-trace CSVs of the same bench.py command -- one of the three-chain step (hipGraph, streams) and one
with HCM_SERIAL=1 HCM_GRAPH=0 (every kernel alone on the chip) -- joined per kernel name over the LAST `steps` steps' launches:
launches per step, average duration alone, average duration inside the step, the slowdown, and the per-step totals.  Shows which kernels pay for
running beside the other chains, and what the chains' overlap is worth (sum alone / sum in step / wall).
usage: step_compare.py <concurrent_kernel_trace.csv> <serial_kernel_trace.csv> <steps in each run> [wall_ms]"""
import collections, csv, sys
def load(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].replace("void ", "").replace("hcm::", "")
        if "copyBuffer" in n or n.startswith("at::") or "elementwise_kernel" in n or n.startswith("absmax_kernel") or "fillBuffer" in n:
            continue
        n = n.split("(")[0]
        t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        e = d.setdefault(n, [0, 0])
        e[0] += 1; e[1] += t
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
steps = int(sys.argv[3])
rows = []
for n in a:
    ca, ta = a[n]
    cb, tb = b.get(n, (0, 0))
    rows.append((n, ca / steps, (tb / cb / 1e3) if cb else float("nan"), ta / ca / 1e3, ta / 1e6 / steps, (tb / 1e6 / steps) if cb else float("nan")))
rows.sort(key=lambda r: -r[4])
sa = sum(r[4] for r in rows); sb = sum(r[5] for r in rows if r[5] == r[5])
print(f"per step: kernel time inside the three-chain step {sa:.3f} ms, the same launches alone {sb:.3f} ms" + (f", wall {sys.argv[4]} ms" if len(sys.argv) > 4 else "") + "\n")
print("| kernel | launches/step | us alone | us in step | slowdown | ms/step in step | ms/step alone |")
print("|---|---|---|---|---|---|---|")
for n, c, ub, ua, ma, mb in rows[:40]:
    nn = n if len(n) < 90 else n[:90] + "..."
    print(f"| `{nn}` | {c:.0f} | {ub:.1f} | {ua:.1f} | {ua / ub if ub == ub and ub > 0 else float('nan'):.2f} | {ma:.3f} | {mb:.3f} |")
for n in b:
    if n not in a:
        print(f"| `{n[:90]}` (serial run only) | {b[n][0] / steps:.0f} | {b[n][1] / b[n][0] / 1e3:.1f} | - | - | - | {b[n][1] / 1e6 / steps:.3f} |")
