#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of k_dp_wave for one config-4 device call (separate --pmc passes); usage: [ENV=..] bash tools/r06_traffic.sh <tag>
export TMPDIR=/tmp
REPO=$(pwd); T=${1:-r06_tr}; OUT=$REPO/gpurun_out/$T; mkdir -p "$OUT"
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -k 5 200 rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o p -- \
    python "$REPO/bench.py" --workload config4 --contigs 6250 --no-cpu-baseline --no-secondary --contexts 1 --gen-procs 1 --steps 2 --warmup 1 > /dev/null 2> "$OUT/$C.log" )
done
python - <<PY
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    fs=glob.glob("$OUT/%s/**/*counter_collection.csv"%c,recursive=True)
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        for k in ("${KERNELS:-k_dp_wave}").split(","):
            if k in r['Kernel_Name'] and r['Counter_Name']==c and int(r['Grid_Size'])>${MINGRID:-1000000}:
                agg[k].append(float(r['Counter_Value']))
    for k,v in agg.items(): print("%-16s %-12s %.1f MB (x%s)"%(k,c,sum(v)/len(v)*1024*(2 if c=="FETCH_SIZE" else 1)/1e6, 2 if c=="FETCH_SIZE" else 1))
PY
