#!/bin/bash
# every kernel of ONE device call of the config 4 job (6 250 contigs, one context), in launch order with its duration:
#   gpurun -- 'bash tools/step_kernels.sh tag [config4|config2|config3|config5]'   ->  gpurun_out/sk_<tag>/step.txt
TAG=${1:-q}; WL=${2:-config4}; REPO=$(pwd); OUT=$REPO/gpurun_out/sk_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python "$REPO/bench.py" --workload $WL --contigs 6250 \
    --no-cpu-baseline --no-secondary --contexts 1 --gen-procs 1 --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/log.txt" )
python - <<PY
import csv, glob, re
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [re.search(r"(k_\w+|__amd\w+)", r["Kernel_Name"]) for r in rows]
names = [m.group(1) if m else r["Kernel_Name"][:30] for m, r in zip(names, rows)]
# the last device call: from the last k_digitize on
last = max(i for i, n in enumerate(names) if n == "k_digitize")
t0 = int(rows[last]["Start_Timestamp"]); tot = 0
with open("$OUT/step.txt", "w") as o:
    for r, n in zip(rows[last:], names[last:]):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += d
        o.write("%9.1f us  +%8.1f  %-28s grid %s\n" % (d, (int(r["Start_Timestamp"]) - t0) / 1e3, n, r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
    o.write("kernel sum %.1f us, span %.1f us\n" % (tot, (int(rows[-1]["End_Timestamp"]) - t0) / 1e3))
print(open("$OUT/step.txt").read())
PY
