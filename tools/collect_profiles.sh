#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh r01_c'
# Kernel trace + stats and the HBM counters are separate passes (one counter per pass).
# Results land in gpurun_out/prof_<tag>/; tools/profiles_summary.py turns them into profiles/<tag>_*.
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for WL in config2 config3; do
    ( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/trace_$WL" -o t -- \
        python "$REPO/bench.py" --workload $WL --no-cpu-baseline > "$OUT/bench_$WL.json" 2> "$OUT/trace_$WL.log" )
    for C in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${WL}_$C" -o p -- \
            python "$REPO/bench.py" --workload $WL --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_${WL}_$C.log" )
    done
done
find "$OUT" -name "*.db" -o -name "*counter_collection.csv" | head -20
