"""Per-kernel roofline table of one collected workload: duration from the kernel trace, HBM bytes from the FETCH_SIZE / WRITE_SIZE passes
(corrected with the calibration factors of profiles/<tag>_pmc_calibration.md: FETCH_SIZE counts 0.5 per byte of narrow coalesced reads on
gfx950, WRITE_SIZE 1.0), VALU wave-instructions from the SQ_INSTS_VALU pass.
usage: python tools/kernel_rooflines.py r03_b config4 > profiles/r03_b_config4_kernel_rooflines.md"""
import collections
import csv
import os
import re
import sqlite3
import sys

tag, wl = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", "prof_" + tag)
HBM, VALU_PEAK = 8000.0, 1024 * 2.4e9 / 2.0


def short(name):
    m = re.search(r"(k_\w+)", name)
    return m.group(1) if m else name.split("(")[0]


def counter(c):
    acc, n = collections.defaultdict(float), collections.Counter()
    path = os.path.join(src, "pmc_%s_%s" % (wl, c), "p_counter_collection.csv")
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == c]
    # full-size launches only (the largest grid of each kernel)
    gmax = collections.defaultdict(int)
    for r in rows:
        gmax[short(r["Kernel_Name"])] = max(gmax[short(r["Kernel_Name"])], int(r["Grid_Size"]))
    for r in rows:
        k = short(r["Kernel_Name"])
        if int(r["Grid_Size"]) == gmax[k]:
            acc[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: acc[k] / n[k] for k in acc}


fetch, write, valu = counter("FETCH_SIZE"), counter("WRITE_SIZE"), counter("SQ_INSTS_VALU")
db = sqlite3.connect(os.path.join(src, "trace_" + wl, "t_results.db"))
rows = list(db.execute("select name, total_calls, total_duration, average from top_kernels"))
tot = sum(r[2] for r in rows)
print("# %s %s: every kernel against the HBM roofline and the VALU issue peak\n" % (tag, wl))
print("Durations: `rocprofv3 --kernel-trace --stats` (average over the calls of the trace, microseconds).  HBM bytes per launch: separate\n"
      "`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, full-size launches only, FETCH_SIZE / 0.5 + WRITE_SIZE / 1.0 (KiB; calibration in\n"
      "profiles/%s_pmc_calibration.md).  VALU wave-instructions per launch: `--pmc SQ_INSTS_VALU`.  Peaks: HBM 8 TB/s; VALU issue\n"
      "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction = 1.23e12 / s (MI355X_MICROARCH.md).  A fraction near or above 1 (k_ovl_stops)\n"
      "means the counters also see requests the Infinity Cache served: the kernel re-reads what the kernel before it just wrote.\n" % tag)
print("| kernel | calls | avg us | %% of kernel time | HBM MB / launch | GB/s | frac of HBM | VALU M-instr / launch | frac of VALU issue |")
print("|---|---|---|---|---|---|---|---|---|")
for name, calls, total, avg in sorted(rows, key=lambda r: -r[2]):
    k = short(name)
    if not k.startswith("k_") or total / tot < 0.004:
        continue
    b = (fetch.get(k, 0.0) / 0.5 + write.get(k, 0.0)) * 1024.0
    gbs = b / (avg * 1e-6) / 1e9 if avg > 0 else 0.0
    vi = valu.get(k, 0.0)
    print("| %s | %d | %.1f | %.1f | %.1f | %.0f | %.3f | %.1f | %.3f |" % (k, calls, avg, 100.0 * total / tot, b / 1e6, gbs, gbs / HBM, vi / 1e6,
                                                                   vi / (avg * 1e-6) / VALU_PEAK if avg > 0 else 0.0))
