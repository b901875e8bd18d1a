"""Turn the output of tools/collect_profiles.sh into the committed summaries under profiles/.
usage: python tools/profiles_summary.py r01_c
Writes profiles/<tag>_bench_<workload>_kernel_stats.md, profiles/<tag>_bench_<workload>.json,
profiles/pmc/<tag>_<workload>_<COUNTER>.csv (DP kernel rows only) and refreshes profiles/r01_pmc_traffic.json,
which bench.py reads for roofline.traffic."""
import csv
import json
import os
import re
import subprocess
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs("profiles/pmc", exist_ok=True)
traffic = {
    "_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (profiles/pmc/*.csv), average over the "
             "launches of the DP kernel; counters are in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports "
             "half of a wide coalesced read). WRITE_SIZE matches the kernel's own store count, so it is used as is.",
    "_collected": tag,
}
for wl in ("config2", "config3"):
    bench = json.loads(open(os.path.join(src, "bench_%s.json" % wl)).read().strip().splitlines()[-1])
    json.dump(bench, open("profiles/%s_bench_%s.json" % (tag, wl), "w"), indent=1)
    title = "Round 1 (%s) -- bench.py --workload %s: %s, %.1f Mbp/s, %d DP chains" % (
        tag.split("_")[-1], wl, bench["config"]["workload"], bench["value"], bench["roofline"]["chains"])
    md = subprocess.run([sys.executable, "tools/rocpd_stats.py", os.path.join(src, "trace_" + wl, "t_results.db"), title],
                        capture_output=True, text=True, check=True).stdout
    open("profiles/%s_bench_%s_kernel_stats.md" % (tag, wl), "w").write(md)
    # the connection scoring is one kernel, or (segmented) a group of kernels launched once per step
    group = bench["roofline"].get("kernels") or [bench["roofline"]["kernel"]]
    kern = bench["roofline"]["kernel"]
    pat = re.compile(r"(?<![A-Za-z0-9_])(" + "|".join(re.escape(k) for k in group) + r")(?![A-Za-z0-9_])")
    once = "k_seg_gather" if len(group) > 1 else group[0]            # launched exactly once per step
    # per-step duration of the group from the kernel trace, next to bench.py's own HIP-event figure
    tot_us, steps = 0.0, 0
    for line in md.splitlines():
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) >= 4 and pat.fullmatch(cells[0] or "-"):
            tot_us += float(cells[2])
            if cells[0] == once:
                steps = int(cells[1])
    if steps:
        md += "\nConnection scoring (%s): %.1f us per step over %d steps in the trace; bench.py's HIP events: %.1f us per step.\n" % (
            " + ".join(group), tot_us / steps, steps, 1e3 * bench["roofline"]["kernel_ms_per_step"])
        open("profiles/%s_bench_%s_kernel_stats.md" % (tag, wl), "w").write(md)
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = list(csv.DictReader(open(os.path.join(src, "pmc_%s_%s" % (wl, c), "p_counter_collection.csv"))))
        keep = [r for r in rows if pat.search(r["Kernel_Name"]) and r["Counter_Name"] == c]
        with open("profiles/pmc/%s_%s_%s.csv" % (tag, wl, c), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
        launches = sum(1 for r in keep if re.search(r"(?<![A-Za-z0-9_])" + re.escape(once) + r"(?![A-Za-z0-9_])", r["Kernel_Name"]))
        vals[c] = sum(float(r["Counter_Value"]) for r in keep) / max(1, launches)
    traffic[bench["config"]["workload"]] = {
        "kernel": kern, "FETCH_SIZE_KiB": round(vals["FETCH_SIZE"], 1), "WRITE_SIZE_KiB": round(vals["WRITE_SIZE"], 1),
        "hbm_bytes_per_launch": int(round((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)),
    }
json.dump(traffic, open("profiles/r01_pmc_traffic.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))
