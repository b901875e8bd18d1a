"""Whole-pipeline figures of one device call ("whole-pipeline rocprof HBM bytes / bp"), from what
tools/collect_profiles.sh collected for one workload: every kernel's time per call from the kernel trace and its HBM bytes per call
from the FETCH_SIZE / WRITE_SIZE passes (corrected as in tools/kernel_rooflines.py: FETCH_SIZE counts 0.5 per byte of narrow coalesced
reads on gfx950, WRITE_SIZE 1.0; both in KiB).  A "call" is one pga_find_genes: the launches of a run divided by its k_digitize launches
(bench.py issues full-size calls only, warm-up included).

usage: python tools/pipeline_traffic.py r04_a config4 [commit]
Writes profiles/<tag>_<workload>_pipeline.json and puts the same under "pipeline:<workload name>" in profiles/r06_pmc_traffic.json,
where bench.py finds it (`pipeline` in its JSON line)."""
import collections
import csv
import json
import os
import re
import sqlite3
import subprocess
import sys

tag, wl = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
src = os.path.join("gpurun_out", "prof_" + tag)
F_READ, F_WRITE = 0.5, 1.0


def short(name):
    m = re.search(r"(k_\w+|__amd_rocclr_\w+)", name)
    return m.group(1) if m else name.split("(")[0]


def counter(c):
    acc = collections.defaultdict(float)
    calls = 0
    for r in csv.DictReader(open(os.path.join(src, "pmc_%s_%s" % (wl, c), "p_counter_collection.csv"))):
        if r["Counter_Name"] != c:
            continue
        k = short(r["Kernel_Name"])
        acc[k] += float(r["Counter_Value"])
        calls += k == "k_digitize"
    return {k: v / max(calls, 1) for k, v in acc.items()}, calls


bench = json.loads(open(os.path.join(src, "bench_%s.json" % wl)).read().strip().splitlines()[-1])
bases_per_call = bench["config"]["bases"] / max(1, bench["config"]["device_calls_per_step_rank0"])
fetch, nf = counter("FETCH_SIZE")
write, nw = counter("WRITE_SIZE")
db = sqlite3.connect(os.path.join(src, "trace_" + wl, "t_results.db"))
rows = [(short(n), calls, tot) for n, calls, tot in db.execute("select name, total_calls, total_duration from top_kernels")]
n_calls = sum(c for k, c, _ in rows if k == "k_digitize")
kern = collections.OrderedDict()
for k, calls, tot in sorted(rows, key=lambda r: -r[2]):
    e = kern.setdefault(k, {"launches_per_call": 0.0, "ms_per_call": 0.0})
    e["launches_per_call"] += calls / n_calls
    e["ms_per_call"] += tot / n_calls / 1e3
total_bytes = 0.0
for k, e in kern.items():
    b = (fetch.get(k, 0.0) / F_READ + write.get(k, 0.0) / F_WRITE) * 1024.0
    e["hbm_MB_per_call"] = round(b / 1e6, 2)
    e["launches_per_call"] = round(e["launches_per_call"], 2)
    e["ms_per_call"] = round(e["ms_per_call"], 4)
    total_bytes += b
kernel_ms = sum(e["ms_per_call"] for e in kern.values())
out = {
    "workload": bench["config"]["workload"], "collected": tag, "collected_at_commit": commit,
    "bases_per_call": int(bases_per_call), "calls_in_trace": n_calls, "calls_in_counter_passes": [nf, nw],
    "pipeline_hbm_bytes_per_bp": round(total_bytes / bases_per_call, 2),
    "pipeline_hbm_GB_per_call": round(total_bytes / 1e9, 3),
    "kernel_ms_per_call": round(kernel_ms, 3),
    "hbm_GBps_over_kernel_time": round(total_bytes / (kernel_ms * 1e-3) / 1e9, 1),
    "method": "rocprofv3 --kernel-trace --stats for the times, separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes for the bytes "
              "(KiB; FETCH_SIZE / 0.5, WRITE_SIZE / 1.0: profiles/%s_pmc_calibration.md), every launch of the run divided by its "
              "pga_find_genes calls; one context, so a kernel's time is its own; requests the Infinity Cache served are counted" % tag,
    "kernels": {k: e for k, e in kern.items() if e["ms_per_call"] >= 0.004 or e["hbm_MB_per_call"] >= 5.0},
}
path = "profiles/%s_%s_pipeline.json" % (tag, wl)
json.dump(out, open(path, "w"), indent=1)
try:
    merged = json.load(open("profiles/r06_pmc_traffic.json"))
except (OSError, ValueError):
    merged = {}
merged["pipeline:" + out["workload"]] = out
json.dump(merged, open("profiles/r06_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
for k, e in list(out["kernels"].items())[:16]:
    print("%-26s %7.3f ms  %8.1f MB  x%.0f" % (k, e["ms_per_call"], e["hbm_MB_per_call"], e["launches_per_call"]))
