"""Per-kernel MFMA utilisation from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace` CSV.

util = MFMA busy cycles / (SIMD-cycles the kernel was resident for) = busy / (active_cycles x 1024 SIMDs), where
active_cycles = GRBM_GUI_ACTIVE / 8 XCDs MINUS the per-dispatch idle offset of this collection mode.

Why the offset (round-4 review, item 9): under per-dispatch counter collection GRBM_GUI_ACTIVE also counts the time the profiler holds the
queue around the kernel (counter start / stop, ~10 us), so `busy / GUI` under-read every SHORT kernel (gemm256f: 23 % reported, 32 % from the
exact MFMA count over the kernel's own duration) while long kernels were right.  The offset is not a constant of the chip, so it is FITTED from
the file itself: GUI/8 = clock x duration + offset, the clock from the long dispatches and the offset from the short ones (medians, iterated); every dispatch then
uses its OWN clock, (GUI/8 - offset) / duration, so DVFS differences between kernels stay in.  The fit is printed; `--selfcheck` asserts that a
`mfma_only<8>` dispatch of tools/mfma_calib.py in the same file (a kernel of nothing but back-to-back MFMAs: 16 busy cycles per 16.8-cycle issue
slot = 95 %) reads 90-100 %.

usage: pmc_mfma.py <counter_collection.csv> <steps> [--selfcheck]"""
import collections
import csv
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
selfcheck = "--selfcheck" in sys.argv
disp = {}
for r in csv.DictReader(open(args[0])):
    d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
steps = int(args[1]) if len(args) > 1 else 4
pts = [(d["t"], d["GRBM_GUI_ACTIVE"] / 8.0) for d in disp.values() if d.get("GRBM_GUI_ACTIVE", 0) > 0 and d["t"] > 0]


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if v else None


# clock from the LONG dispatches (>= 100 us: the hold is a few per cent of them), offset from the SHORT ones (< 30 us: mostly hold), iterated; a plain
# least-squares line through everything is pulled around by the few long kernels of a calibration run
longs = [(x, y) for x, y in pts if x >= 100.0] or pts
shorts = [(x, y) for x, y in pts if x < 30.0]
slope, off = 2000.0, 0.0
for _ in range(4):
    slope = median([(y - off) / x for x, y in longs]) or slope
    off = max(median([y - slope * x for x, y in shorts]) or 0.0, 0.0)
rows = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0, 0.0])
for d in disp.values():
    n = d["name"].replace("void ", "").replace("hcm::", "").replace("(anonymous namespace)::", "").split("(")[0][:64]
    gui = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    act = max(gui - off, 0.5 * slope * d["t"])            # never below half the fitted clock x duration (a dispatch the fit does not describe)
    a = rows[n]
    a[0] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a[1] += act; a[2] += 1; a[3] += d["t"]; a[4] += gui
print(f"fit over {len(pts)} dispatches ({len(longs)} long, {len(shorts)} short): GRBM_GUI_ACTIVE/8 = {slope:.0f} cycles/us x duration + {off:.0f} cycles "
      f"(= {off / slope:.1f} us of profiler hold per dispatch at {slope / 1e3:.2f} GHz)\n")
print("| kernel | launches/step | us/launch (under counters) | MFMA busy % of SIMD-cycles (offset-corrected) | uncorrected busy / GUI % | clock GHz |\n|---|---|---|---|---|---|")
tot_b = tot_c = 0.0
check = None
for n, (b, c, cnt, t, gui) in sorted(rows.items(), key=lambda kv: -kv[1][3]):
    tot_b += b; tot_c += c
    util = b / (c * 1024) * 100 if c else 0.0
    if n.startswith("mfma_only<8>"):
        check = util
    if b == 0 and t / cnt < 20:
        continue
    raw = b / (gui * 1024) * 100 if gui else 0.0
    print(f"| `{n}` | {cnt / steps:.1f} | {t / cnt:.1f} | {util:.1f} | {raw:.1f} | {c / t / 1e3:.2f} |")
print(f"\nall kernels: MFMA busy {tot_b / (tot_c * 1024) * 100:.1f} % of the SIMD-cycles in which a kernel was resident (offset-corrected)")
if selfcheck:
    assert check is not None, "--selfcheck: no mfma_only<8> dispatch in this file (run tools/mfma_calib.py under the same rocprofv3 pass)"
    print(f"self-check: mfma_only<8> reads {check:.1f} % (expected 95 %: 16 busy cycles per 16.8-cycle issue slot)")
    assert 90.0 <= check <= 100.5, check
