#!/bin/bash
# LDS counters of one config-4 call (bank conflicts of the table walks)
export TMPDIR=/tmp; R=$(pwd); O=$R/gpurun_out/${1:-r05_lds}; mkdir -p $O
( cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_LDS[A-Z_]*" | sort -u | tr '\n' ' ' ) > $O/avail.txt; cat $O/avail.txt; echo
( cd /tmp && timeout -k 5 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $O/p -o p -- \
    python $R/bench.py --workload config4 --contigs 6250 --no-cpu-baseline --no-secondary --contexts 1 --gen-procs 1 --steps 2 --warmup 1 > /dev/null 2> $O/p.log )
python - <<PY
import csv,glob,collections,re
fs=glob.glob("$O/p/**/*counter_collection.csv",recursive=True)
if not fs: print(open("$O/p.log").read()[-1500:]); raise SystemExit
agg=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    m=re.search(r'(k_\w+)', r['Kernel_Name']); nm=m.group(1) if m else r['Kernel_Name'][:20]
    agg[(nm,int(r['Grid_Size']),r['Counter_Name'])].append(float(r['Counter_Value']))
for k in sorted(agg):
    if k[0] in ('k_coding_score_quads','k_score_starts','k_extract_tile','k_dp_wave'): print("%-22s grid %9d %-26s avg %.4g"%(k[0],k[1],k[2],sum(agg[k])/len(agg[k])))
PY
