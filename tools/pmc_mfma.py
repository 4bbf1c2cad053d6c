"""Per-kernel MFMA utilisation from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace` CSV:
util = MFMA busy cycles / (GUI-active cycles summed over the 8 XCDs / 8 x 1024 SIMDs).  usage: pmc_mfma.py <counter_collection.csv> <steps>"""
import csv, sys, collections
rows = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0])
disp = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Dispatch_Id"])
    d = disp.setdefault(k, {"name": r["Kernel_Name"], "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for d in disp.values():
    n = d["name"].replace("void hcm::", "").replace("hcm::", "")
    n = n.split("(")[0][:64]
    a = rows[n]
    a[0] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a[1] += d.get("GRBM_GUI_ACTIVE", 0.0); a[2] += 1; a[3] += d["t"]
print("| kernel | launches/step | us/launch (under counters) | MFMA busy % of SIMD-cycles |\n|---|---|---|---|")
tot_b = tot_c = 0.0
for n, (b, c, cnt, t) in sorted(rows.items(), key=lambda kv: -kv[1][3]):
    if b == 0 and t / cnt < 20: continue
    util = b / (c / 8 * 1024) * 100 if c else 0.0
    tot_b += b; tot_c += c
    print(f"| `{n}` | {cnt / steps:.0f} | {t / cnt:.1f} | {util:.1f} |")
print(f"\nall kernels: MFMA busy {tot_b / (tot_c / 8 * 1024) * 100:.1f} % of SIMD-cycles while a kernel is resident")
