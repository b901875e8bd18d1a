#!/bin/bash
# instruction counters of k_dp_wave for one config-4 device call (passes a + b of tools/collect_sq_counters.sh), summarised
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_sq
mkdir -p "$OUT"
pass() {
  local name=$1; shift
  ( cd /tmp && timeout -k 5 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- \
    python "$REPO/bench.py" --workload config4 --contigs 6250 --no-cpu-baseline --no-secondary --contexts 1 --gen-procs 1 --steps 2 --warmup 1 > /dev/null 2> "$OUT/$name.log" )
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS
python - <<PY
import csv,glob,collections
for name in "ab":
    fs=glob.glob("$OUT/%s/**/*counter_collection.csv"%name,recursive=True)
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        for k in ("${KERNELS:-k_dp_wave}").split(","):
            if k in r['Kernel_Name'] and int(r['Grid_Size'])>${MINGRID:-1000000}:
                agg[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print("%-14s %-24s %.4g"%(k[0],k[1],sum(v)/len(v)))
PY
