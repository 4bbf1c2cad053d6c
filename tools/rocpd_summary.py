"""Summarise a rocprofv3 rocpd .db (kernel trace): per-kernel count / total / average, plus per-launch-shape rows
for the implicit-GEMM kernel.  Usage: python tools/rocpd_summary.py results.db [steps] > profiles/xxx.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace summary ({sys.argv[1].split('/')[-1]}); all launches in the process (warm-up + timed), {steps} steps assumed for the per-step column")
print(f"\ntotal kernel time {tot/1e6:.2f} ms over {sum(r[1] for r in rows)} launches; per step {tot/1e6/steps:.3f} ms\n")
print("| kernel | launches | total ms | % | avg us | min us | max us | ms/step |")
print("|---|---|---|---|---|---|---|---|")
for name, n, s, a, mn, mx in rows[:40]:
    short = name.replace("hcm::", "").replace("void ", "")
    if len(short) > 90:
        short = short[:90] + "..."
    print(f"| `{short}` | {n} | {s/1e6:.3f} | {100*s/tot:.1f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {s/1e6/steps:.3f} |")
