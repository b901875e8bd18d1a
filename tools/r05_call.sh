#!/bin/bash
# One GPU session of round 5: the GPU test suite, the default bench line, the config-4 (or $WORKLOADS) profile collection and the
# SQ counters, all at the same build.  gpurun --timeout 1200 -- 'bash tools/r05_call.sh r05_a'
set -u
TAG=${1:-r05_a}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
( timeout -k 5 400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 ); tail -2 $O/pytest_gpu.log
( timeout -k 5 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); tail -c 600 $O/bench_default.json
WORKLOADS="${WORKLOADS:-config4}" bash tools/collect_profiles.sh $TAG > $O/collect.log 2>&1
bash tools/collect_sq_counters.sh > $O/sq.log 2>&1
tail -30 $O/sq.log
