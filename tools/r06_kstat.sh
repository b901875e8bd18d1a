#!/bin/bash
# kernel times of 6 250-contig calls on one context (rocprofv3 --kernel-trace --stats), after the tests named in $2
T=${1:-kstat}; K=${2:-stages}
REPO=$(pwd); OUT=$REPO/gpurun_out/$T; mkdir -p "$OUT"; export TMPDIR=/tmp
if [ "$K" != "none" ]; then timeout 900 python -m pytest tests -q -m gpu -x -k "$K" 2>&1 | tail -2; fi
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python "$REPO/bench.py" --contigs 6250 --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary --steps 6 --warmup 2 > "$OUT/c1.json" 2> "$OUT/c1.err" )
python tools/rocpd_stats.py "$OUT/trace/t_results.db" | head -${3:-16}
