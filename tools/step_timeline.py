"""Per-queue timeline of ONE step from a rocprofv3 --kernel-trace rocpd database: when each HIP queue (stream) is busy,
its first / last kernel, and the serial tail after the encoder chains join.
usage: python tools/step_timeline.py <results.db> [step_index_from_end=2]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
namecol = "display_name" if "display_name" in cols else "kernel_name"
names = dict(cur.execute(f"select id, {namecol} from {ks}"))
rows = list(cur.execute(f"select start, end, queue_id, kernel_id from {kd} order by start"))
# steps are delimited by the stem/pack kernel of the RGB chain (one per step)
marks = [i for i, r in enumerate(rows) if "pack_frame" in names[r[3]]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = marks[-k - 1], marks[-k]
# the step starts a little before the pack kernel (BERT / depth chains are enqueued first): take kernels from the end of the previous tail
step = rows[a - 40:b - 40] if a >= 40 else rows[a:b]
t0 = min(r[0] for r in step)
t1 = max(r[1] for r in step)
print(f"step window {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels")
qs = {}
for s, e, q, kid in step:
    qs.setdefault(q, []).append((s, e, names[kid]))
for q, lst in sorted(qs.items(), key=lambda kv: kv[1][0][0]):
    busy = sum(e - s for s, e, _ in lst)
    print(f"queue {q}: {len(lst):4d} kernels, first +{(lst[0][0] - t0) / 1e3:8.1f} us, last end +{(max(e for _, e, _ in lst) - t0) / 1e3:8.1f} us, busy {busy / 1e3:8.1f} us")
    print(f"     first: {lst[0][2][:70]}\n     last : {lst[-1][2][:70]}")
# concurrency profile: number of kernels in flight, sampled
ev = sorted([(s, 1) for s, e, q, k in step] + [(e, -1) for s, e, q, k in step])
cur_n, last_t, hist = 0, t0, {}
for t, d in ev:
    hist[cur_n] = hist.get(cur_n, 0) + (t - last_t)
    cur_n += d
    last_t = t
tot = sum(hist.values())
print("kernels in flight: " + ", ".join(f"{n}: {100 * v / tot:.1f}%" for n, v in sorted(hist.items())))
