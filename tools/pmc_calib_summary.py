"""Per-kernel averages of every counter in a rocprofv3 counter_collection CSV (one row per dispatch and counter).
usage: pmc_calib_summary.py <counter_collection.csv> [min_us]"""
import csv, sys, collections
disp = {}
for r in csv.DictReader(open(sys.argv[1])):
    d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                           "grid": r.get("Grid_Size", ""), "wg": r.get("Workgroup_Size", "")})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.OrderedDict()
for d in disp.values():
    n = d["name"].replace("void hcm::", "").replace("hcm::", "").split("(")[0][:70] + f" grid={d['grid']}"
    a = agg.setdefault(n, collections.defaultdict(float))
    a["_n"] += 1; a["_t"] += d["t"]
    for k, v in d.items():
        if k not in ("name", "t", "grid", "wg"): a[k] += v
names = sorted({k for a in agg.values() for k in a if not k.startswith("_")})
print("| kernel | launches | us/launch (under counters) | " + " | ".join(names) + " |")
print("|---|---|---|" + "---|" * len(names))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
for n, a in agg.items():
    if a["_t"] / a["_n"] < min_us: continue
    print(f"| `{n}` | {int(a['_n'])} | {a['_t'] / a['_n']:.1f} | " + " | ".join(f"{a[k] / a['_n']:.4g}" for k in names) + " |")
