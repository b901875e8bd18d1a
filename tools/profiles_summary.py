"""Turn the output of tools/collect_profiles.sh into the committed summaries under profiles/.
usage: python tools/profiles_summary.py r03_a [commit]
Writes profiles/<tag>_bench_<workload>_kernel_stats.md, profiles/<tag>_bench_<workload>.json,
profiles/pmc/<tag>_<workload>_<COUNTER>.csv (connection-scoring kernel rows only), profiles/<tag>_pmc_calibration.md and
refreshes profiles/r06_pmc_traffic.json, which bench.py reads for roofline.traffic / roofline.valu_issue_frac."""
import csv
import json
import os
import re
import subprocess
import sys

tag = sys.argv[1]
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs("profiles/pmc", exist_ok=True)
GIB = float(1 << 30)

# ---- calibration: counter value (KiB) per known GiB, by kernel
cal = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    path = os.path.join(src, "cal_" + c, "p_counter_collection.csv")
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != c:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"])
        key = (c, name, r.get("Grid_Size", ""))
        cal.setdefault((c, r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
lines = ["# FETCH_SIZE / WRITE_SIZE against known byte counts (tools/pmc/pmc_calibrate.hip, 1 GiB streamed per kernel)\n",
         "Counters are in KiB.  `ratio` = counter bytes / bytes the kernel asked for.\n",
         "| counter | kernel | launches | counter KiB (avg) | ratio |", "|---|---|---|---|---|"]
factors = {}
for (c, name), vals in sorted(cal.items()):
    avg = sum(vals) / len(vals)
    short = re.sub(r"^void ", "", name)
    asked = GIB
    ratio = avg * 1024.0 / asked
    lines.append("| %s | %s | %d | %.0f | %.3f |" % (c, short, len(vals), avg, ratio))
    factors[(c, short)] = ratio
open("profiles/%s_pmc_calibration.md" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

def factor(counter, pattern, default):
    vals = [v for (c, n), v in factors.items() if c == counter and re.search(pattern, n)]
    return sum(vals) / len(vals) if vals else default

# the connection-scoring kernel reads 4- and 8-byte values per lane (coalesced) and 64-byte records of stop nodes, and writes
# 1- to 8-byte values per lane: the calibration of the narrow accesses applies
f_read = factor("FETCH_SIZE", r"k_read<(unsigned int|HIP_vector_type<unsigned int, 2u?>)>", 0.5)
f_write = factor("WRITE_SIZE", r"k_write<(unsigned int|HIP_vector_type<unsigned int, 2u?>)>", 1.0)
traffic = {
    "_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (profiles/pmc/*.csv), average over the launches of the "
             "connection-scoring kernel; counters are in KiB.  Each counter is divided by what it reports per requested byte on "
             "known-byte-count kernels with the same access widths (profiles/%s_pmc_calibration.md: 4- and 8-byte coalesced loads / "
             "stores; MI355X_MICROARCH.md describes the same effect for 16-byte loads)." % tag,
    "_collected": tag, "collected_at_commit": commit, "_fetch_counter_per_byte": round(f_read, 4), "_write_counter_per_byte": round(f_write, 4),
}
for wl in ("config4", "config3", "config2", "config5"):
    bpath = os.path.join(src, "bench_%s.json" % wl)
    if not os.path.exists(bpath):
        continue
    bench = json.loads(open(bpath).read().strip().splitlines()[-1])
    json.dump(bench, open("profiles/%s_bench_%s.json" % (tag, wl), "w"), indent=1)
    roof = bench["roofline"]
    title = "Round 6 (%s) -- bench.py --workload %s --contexts 1: %s, %.1f Mbp/s, %d chains per launch" % (
        tag.split("_")[-1], wl, bench["config"]["workload"], bench["value"], roof["chains_per_launch"])
    md = subprocess.run([sys.executable, "tools/rocpd_stats.py", os.path.join(src, "trace_" + wl, "t_results.db"), title],
                        capture_output=True, text=True, check=True).stdout
    group = roof.get("kernels") or [roof["kernel"]]
    pat = re.compile(r"(?<![A-Za-z0-9_])(" + "|".join(re.escape(k) for k in group) + r")(?![A-Za-z0-9_])")
    once = "k_seg_gather" if len(group) > 1 else group[0]
    tot_us, launches = 0.0, 0
    for line in md.splitlines():
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) >= 4 and pat.fullmatch(cells[0] or "-"):
            tot_us += float(cells[2])
            if cells[0] == once:
                launches = int(cells[1])
    if launches:
        md += "\nConnection scoring (%s): %.1f us per launch over %d launches in the trace; bench.py's HIP events: %.1f us per launch; " \
              "%d node-passes per launch x 64 B = %.1f GB/s = %.4f of 8 TB/s.\n" % (
                  " + ".join(group), tot_us / launches, launches, 1e3 * roof["kernel_ms_per_launch"], roof["node_passes_per_launch"],
                  64.0 * roof["node_passes_per_launch"] / (tot_us / launches * 1e-6) / 1e9,
                  64.0 * roof["node_passes_per_launch"] / (tot_us / launches * 1e-6) / 8e12)
    open("profiles/%s_bench_%s_kernel_stats.md" % (tag, wl), "w").write(md)
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        path = os.path.join(src, "pmc_%s_%s" % (wl, c), "p_counter_collection.csv")
        if not os.path.exists(path):
            continue
        rows = list(csv.DictReader(open(path)))
        if not rows:
            continue
        keep = [r for r in rows if pat.search(r["Kernel_Name"]) and r["Counter_Name"] == c]
        with open("profiles/pmc/%s_%s_%s.csv" % (tag, wl, c), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
        is_once = lambda r: re.search(r"(?<![A-Za-z0-9_])" + re.escape(once) + r"(?![A-Za-z0-9_])", r["Kernel_Name"]) is not None
        if len(group) == 1:
            # bench.py warms the clocks up with quarter-size calls before its steps: only the full-size launches count
            gmax = max(int(r["Grid_Size"]) for r in keep if is_once(r))
            keep = [r for r in keep if int(r["Grid_Size"]) == gmax]
        n = sum(1 for r in keep if is_once(r))
        vals[c] = sum(float(r["Counter_Value"]) for r in keep) / max(1, n)
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        hbm = (vals["FETCH_SIZE"] / f_read + vals["WRITE_SIZE"] / f_write) * 1024
        traffic[bench["config"]["workload"]] = {
            "valu_insts_per_launch": int(vals["SQ_INSTS_VALU"]) if "SQ_INSTS_VALU" in vals else None,
            "kernel": roof["kernel"], "FETCH_SIZE_KiB": round(vals["FETCH_SIZE"], 1), "WRITE_SIZE_KiB": round(vals["WRITE_SIZE"], 1),
            "hbm_bytes_per_launch": int(round(hbm)), "algorithmic_bytes_per_launch": int(64 * roof["node_passes_per_launch"]),
            "ratio_to_algorithmic": round(hbm / (64.0 * roof["node_passes_per_launch"]), 3),
        }
# entries of workloads that were not part of this collection stay (a later collection of one workload refreshes that one only)
try:
    merged = json.load(open("profiles/r06_pmc_traffic.json"))
except (OSError, ValueError):
    merged = {}
for k, v in traffic.items():
    if isinstance(v, dict):
        v = dict(v, collected=tag, collected_at_commit=commit)
        # what other collections added to the entry (the SQ pipe counters of tools/sq_counters.py) stays
        if isinstance(merged.get(k), dict):
            v = dict({kk: vv for kk, vv in merged[k].items() if kk not in v}, **v)
    merged[k] = v
traffic = merged
json.dump(traffic, open("profiles/r06_pmc_traffic.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))
