#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh r02_a'
# Kernel trace + stats and the HBM counters are separate passes (one counter per pass, never together with a trace).
# Workloads: one device call of the config 4 job (6 250 x 20 kbp, bench.py's default sub-batch), config3 (1000 x 50 kbp), config2 (1 x 5 Mbp)
# and config5 (1 x 200 Mbp, single mode), one context each so that a kernel's duration is its own.  Results land in gpurun_out/prof_<tag>/; tools/profiles_summary.py turns them into
# profiles/<tag>_*.
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-secondary --contexts 1 --gen-procs 1"
run() { local name=$1; shift; ( cd /tmp && timeout -k 5 240 "$@" ) ; }
# known-byte-count kernels: what FETCH_SIZE / WRITE_SIZE report for 4 / 8 / 16 bytes per lane and for 64-byte records
for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout -k 5 120 rocprofv3 --pmc $C --output-format csv -d "$OUT/cal_$C" -o p -- "$REPO/tools/pmc/pmc_calibrate" > "$OUT/cal_$C.log" 2>&1 )
done
for WL in ${WORKLOADS:-config4 config3 config2 config5}; do
    ARGS="--workload $WL $COMMON"
    [ $WL = config4 ] && ARGS="--workload config4 --contigs 6250 $COMMON"
    ( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_$WL" -o t -- \
        python "$REPO/bench.py" $ARGS --steps 4 --warmup 2 > "$OUT/bench_$WL.json" 2> "$OUT/trace_$WL.log" )
    # HBM bytes and VALU wave-instructions: one counter per pass, never together with a trace
    for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
        ( cd /tmp && timeout -k 5 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${WL}_$C" -o p -- \
            python "$REPO/bench.py" $ARGS --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_${WL}_$C.log" )
    done
done
find "$OUT" -name "*.db" -o -name "*counter_collection.csv" | head -20
