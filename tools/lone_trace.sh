#!/bin/bash
# kernels of a lone 20 kbp find_genes call, in order:  gpurun -- 'bash tools/lone_trace.sh'
REPO=$(pwd); OUT=$REPO/gpurun_out/lone_trace; mkdir -p "$OUT"; export TMPDIR=/tmp
( cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python "$REPO/tools/lone_timing.py" > "$OUT/log.txt" 2>&1 )
python - <<PY
import csv, glob, re
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [(re.search(r"(k_\w+|__amd\w+)", r["Kernel_Name"]) or [None, r["Kernel_Name"][:30]])[1] for r in rows]
last = max(i for i, n in enumerate(names) if n == "k_digitize")
t0 = int(rows[last]["Start_Timestamp"]); tot = 0
for r, n in zip(rows[last:], names[last:]):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += d
    print("%8.1f us  +%8.1f  %-28s grid %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e3, n, r.get("Grid_Size_X")))
print("kernel sum %.1f us, span %.1f us, %d launches" % (tot, (int(rows[-1]["End_Timestamp"]) - t0) / 1e3, len(rows) - last))
PY
