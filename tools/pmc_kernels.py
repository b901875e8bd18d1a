"""Per-kernel sums of the counters of a rocprofv3 --pmc run (rocpd database or counter_collection csv).
usage: python tools/pmc_kernels.py <dir> [kernel-substring ...]"""
import csv, glob, os, sqlite3, sys, collections
d = sys.argv[1]; want = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
csvs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if csvs:
    for f in csvs:
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r.get("Dispatch_Id"))
            if key not in seen: seen.add(key); calls[k] += 1
else:
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        for name, cn, v in db.execute("select k.name, c.counter_name, c.value from counters_collection c join kernels k on k.dispatch_id = c.dispatch_id"):
            acc[name.replace("(anonymous namespace)::", "").split("(")[0]][cn] += v
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    if want and not any(w in k for w in want): continue
    print(k, f"calls={calls[k]}")
    for cn, v in sorted(acc[k].items()): print(f"    {cn:28s} {v:.4g}")
