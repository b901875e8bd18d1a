#!/bin/bash
# the kernels of one 6 250-contig call on one context in launch order, with the gaps between them (host phases of the call)
REPO=$(pwd); OUT=$REPO/gpurun_out/onecall; mkdir -p "$OUT"; export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d "$OUT/trace" -o t -- python "$REPO/bench.py" --contigs 6250 --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/log.txt" )
python tools/call_timeline.py "$OUT/trace/t_results.db" > "$OUT/timeline.txt"; tail -3 "$OUT/timeline.txt"
