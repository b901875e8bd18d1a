#!/bin/bash
# Kernel trace of one bench workload, summary on stdout:  gpurun -- 'bash tools/quick_trace.sh config2 tag'
WL=${1:-config2}; TAG=${2:-q}
REPO=$(pwd); OUT=$REPO/gpurun_out/qt_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python "$REPO/bench.py" --workload $WL --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/log.txt" )
cut -c1-160 "$OUT/bench.json"
python tools/rocpd_stats.py "$OUT/trace/t_results.db" | head -${3:-24}
