#!/bin/bash
# HBM bytes and time of the connection-scoring launch of one config-4 device call (one context), e.g. under PGA_DP_XCD=0 / 1:
#   gpurun -- 'bash tools/dp_traffic.sh tag'
TAG=${1:-q}; REPO=$(pwd); OUT=$REPO/gpurun_out/dpt_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS="--workload config4 --contigs 6250 --no-cpu-baseline --no-secondary --contexts 1 --gen-procs 1 --steps 3 --warmup 1"
python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dp ms per launch', d['roofline']['kernel_ms_per_launch'])"
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -k 5 200 rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o p -- python "$REPO/bench.py" $ARGS > /dev/null 2> "$OUT/$C.log" )
  python - <<PY
import csv,glob
f=glob.glob("$OUT/$C/**/*counter_collection.csv",recursive=True)[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_dp_wave" in r["Kernel_Name"] and r["Counter_Name"]=="$C"]
g=max(int(r["Grid_Size"]) for r in csv.DictReader(open(f)) if "k_dp_wave" in r["Kernel_Name"])
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_dp_wave" in r["Kernel_Name"] and r["Counter_Name"]=="$C" and int(r["Grid_Size"])==g]
print("$C KiB per launch %.0f  -> MB %.1f" % (sum(v)/len(v), sum(v)/len(v)*1024/(0.5 if "$C"=="FETCH_SIZE" else 1.0)/1e6))
PY
done
