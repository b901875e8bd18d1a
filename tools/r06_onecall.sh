#!/bin/bash
# the kernels of one call of a workload on one context in launch order, with the gaps between them (host phases of the call)
# usage: bash tools/r06_onecall.sh [workload] [extra bench args]
WL=${1:-config4}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/onecall_$WL; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS="--workload $WL --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary --steps 3 --warmup 1"
if [ "$WL" = config4 ]; then ARGS="$ARGS --contigs 6250"; fi
( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d "$OUT/trace" -o t -- python "$REPO/bench.py" $ARGS "$@" > "$OUT/bench.json" 2> "$OUT/log.txt" )
python tools/call_timeline.py "$OUT/trace/t_results.db" > "$OUT/timeline.txt"
python - "$OUT/timeline.txt" <<'PY'
import sys
prev=None
for l in open(sys.argv[1]):
    p=l.split()
    if len(p)<8 or p[2]!='us' or p[3]!='grid':
        print(l.rstrip()); continue
    dur=float(p[1]); at=float(p[-2]); gap=at-prev if prev is not None else 0.0
    print("%-28s %8.1f us  grid %9s  at %8.1f  gap %7.1f" % (p[0][:28], dur, p[4], at, gap))
    prev=at+dur
PY
