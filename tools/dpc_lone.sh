#!/bin/bash
# a handful of contig-per-wavefront waves alone on the device (one contig, metagenomic models): wave cycles and waits per node
#   gpurun --timeout 300 -- 'bash tools/dpc_lone.sh [length]'
set -u
export TMPDIR=/tmp PGA_DP_KERNEL=contig
REPO=$(pwd); OUT=$REPO/gpurun_out/lone; mkdir -p "$OUT"; L=${1:-60000}
cat > /tmp/lone.py <<PY
import sys; sys.path.insert(0, "$REPO")
from pyrodigal_amd import _cabi, benchdata
seq = benchdata.synthetic_contig($L, 0.5, 77)
ctx = _cabi.Context(0)
ctx.set_models([b for _, b in benchdata.load_model_set()])
for _ in range(4):
    res = ctx.find_genes_batch([seq], meta=True)
print(res.nodes, res.n_chains)
PY
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" ; do
( cd /tmp && rm -rf $OUT/a && timeout -k 5 200 rocprofv3 --pmc $set --output-format csv -d "$OUT/a" -o p -- python /tmp/lone.py > $OUT/a.log 2>&1 )
python - <<PY
import csv,glob,collections,re
fs=glob.glob("$OUT/a/**/*counter_collection.csv",recursive=True)
if not fs: print(open("$OUT/a.log").read()[-800:]); raise SystemExit
agg=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'k_dp_contig' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print("%-20s n=%d last %.5g"%(k,len(v),v[-1]))
w=agg['SQ_WAVES'][-1]; print("cycles per wave %.0f  wait %.0f"%(agg['SQ_WAVE_CYCLES'][-1]*4/w, agg['SQ_WAIT_ANY'][-1]*4/w))
PY
done
