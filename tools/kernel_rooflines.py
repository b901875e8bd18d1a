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
    """per CALL (one pga_find_genes = one k_digitize launch; bench.py issues full-size calls only, warm-up included): every launch of a
    kernel counts -- a kernel that runs once per translation-table group and once more for the winners' re-score is the sum of its
    launches (round 5 took 'the largest grid of each kernel', which for k_score_starts was not its main launch)"""
    acc = collections.defaultdict(float)
    calls = 0
    path = os.path.join(src, "pmc_%s_%s" % (wl, c), "p_counter_collection.csv")
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != c:
            continue
        k = short(r["Kernel_Name"])
        acc[k] += float(r["Counter_Value"])
        calls += k == "k_digitize"
    return {k: v / max(calls, 1) for k, v in acc.items()}


fetch, write, valu = counter("FETCH_SIZE"), counter("WRITE_SIZE"), counter("SQ_INSTS_VALU")
db = sqlite3.connect(os.path.join(src, "trace_" + wl, "t_results.db"))
rows = list(db.execute("select name, total_calls, total_duration, average from top_kernels"))
n_calls = max(1, sum(r[1] for r in rows if short(r[0]) == "k_digitize"))
per = collections.OrderedDict()
for name, calls, total, avg in rows:
    e = per.setdefault(short(name), [0, 0.0])
    e[0] += calls; e[1] += total
tot = sum(v[1] for v in per.values())
print("# %s %s: every kernel against the HBM roofline and the VALU issue peak, per device call\n" % (tag, wl))
print("Per CALL (one pga_find_genes; %d calls in the trace): a kernel's launches of a call are summed.  Durations: `rocprofv3 --kernel-trace\n"
      "--stats` (microseconds).  HBM bytes: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, FETCH_SIZE / 0.5 + WRITE_SIZE / 1.0 (KiB;\n"
      "calibration in profiles/%s_pmc_calibration.md).  VALU wave-instructions: `--pmc SQ_INSTS_VALU`.  Peaks: HBM 8 TB/s; VALU issue\n"
      "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction = 1.23e12 / s (MI355X_MICROARCH.md).  A fraction near or above 1 (k_ovl_stops)\n"
      "means the counters also see requests the Infinity Cache served: the kernel re-reads what the kernel before it just wrote.\n" % (n_calls, tag))
print("| kernel | launches / call | us / call | %% of kernel time | HBM MB / call | GB/s | frac of HBM | VALU M-instr / call | frac of VALU issue |")
print("|---|---|---|---|---|---|---|---|---|")
for k, (calls, total) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    if not k.startswith("k_") or total / tot < 0.004:
        continue
    us = total / n_calls
    b = (fetch.get(k, 0.0) / 0.5 + write.get(k, 0.0)) * 1024.0
    gbs = b / (us * 1e-6) / 1e9 if us > 0 else 0.0
    vi = valu.get(k, 0.0)
    print("| %s | %.1f | %.1f | %.1f | %.1f | %.0f | %.3f | %.1f | %.3f |" % (k, calls / n_calls, us, 100.0 * total / tot, b / 1e6, gbs, gbs / HBM, vi / 1e6,
                                                                     vi / (us * 1e-6) / VALU_PEAK if us > 0 else 0.0))
