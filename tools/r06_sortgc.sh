#!/bin/bash
# (historical: PGA_BENCH_SORT_GC was an experiment of the fourth session of round 6 in bench.py, measured and removed -- profiles/r06_f_experiments.md)
# experiment: the contigs of a call in GC order (neighbours share their models) against the job's order; kernel times of 6 250-contig calls
REPO=$(pwd); export TMPDIR=/tmp
for S in 0 1; do
  OUT=$REPO/gpurun_out/sortgc_$S; mkdir -p "$OUT"
  ( cd /tmp && PGA_BENCH_SORT_GC=$S rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python "$REPO/bench.py" --contigs 6250 --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary --steps 6 --warmup 2 > "$OUT/c1.json" 2> "$OUT/c1.err" )
  echo "== sort $S"; python tools/rocpd_stats.py "$OUT/trace/t_results.db" | head -14
done
for S in 0 1; do
  PGA_BENCH_SORT_GC=$S python bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 2 > gpurun_out/sortgc_$S/full.json 2> gpurun_out/sortgc_$S/full.err
  python -c "
import json;d=json.loads(open('gpurun_out/sortgc_$S/full.json').read().strip().splitlines()[-1]);print('sort $S value',d['value'],'ms',d['ms_per_step'],'resident',d['config']['resident_Mbp_s'],'b2b',d['config']['host_to_host_back_to_back_Mbp_s'],'parity',d.get('parity',{}).get('tuples_identical'))"
done
