#!/bin/bash
# kernel times of 6 250-contig calls on one context under environment settings, on ONE box: bash tools/r06_envstat.sh KERNEL "A=1" "B=2 C=3" ...
export TMPDIR=/tmp; REPO=$(pwd); K=$1; shift
i=0
for E in "$@"; do
  i=$((i + 1)); OUT=$REPO/gpurun_out/envstat_$i; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && env $E rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python "$REPO/bench.py" --contigs 6250 --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary --steps 6 --warmup 2 > "$OUT/c1.json" 2> "$OUT/c1.err" )
  echo "$E: $(python tools/rocpd_stats.py "$OUT/trace/t_results.db" | grep "$K" | head -1)  parity $(python -c "import json;d=json.loads(open('$OUT/c1.json').read().strip().splitlines()[-1]);print(d.get('parity',{}).get('tuples_identical'), d['config'].get('resident_ms_per_step'))")"
done
