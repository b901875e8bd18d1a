#!/bin/bash
# SQ pipe counters of one config-4 device call, three --pmc passes of eight counters (never together with a trace):
#   gpurun --timeout 900 -- 'bash tools/collect_sq_counters.sh'   ->  gpurun_out/prof_sq/{a,b,c}/p_counter_collection.csv
# tools/sq_counters.py turns them (and a kernel trace of the same build) into profiles/<tag>_config4_sq_counters.md.
set -u
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_sq
mkdir -p "$OUT"
pass() {
  local name=$1; shift
  ( cd /tmp && timeout -k 5 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- \
    python "$REPO/bench.py" --workload config4 --contigs 6250 --no-cpu-baseline --no-secondary --contexts 1 --gen-procs 1 --steps 2 --warmup 1 > /dev/null 2> "$OUT/$name.log" )
  python - <<PY
import csv,glob,collections,re
fs=glob.glob("$OUT/$name/**/*counter_collection.csv",recursive=True)
if not fs: print("$name: no csv"); print(open("$OUT/$name.log").read()[-600:]); raise SystemExit
agg=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    m=re.search(r'(k_\w+)', r['Kernel_Name']); nm=m.group(1) if m else r['Kernel_Name'][:20]
    agg[(nm,int(r['Grid_Size']),r['Counter_Name'])].append(float(r['Counter_Value']))
for k in sorted(agg):
    if k[0] in ('k_dp_wave','k_dpw_sched','k_dpw_dyn','k_score_starts','k_coding_score_quads','k_extract_tile') and len(agg[k])>=2:
        v=agg[k]; print("%-22s grid %9d %-26s n=%2d avg %.4g"%(k[0],k[1],k[2],len(v),sum(v)/len(v)))
PY
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS
pass c SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_WAVES
