#!/bin/bash
# kernel trace of one-context config-4 calls: per-kernel average durations (us) and calls
T=${1:-r06_t}
mkdir -p gpurun_out/$T
R=$(pwd)
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$T/trace -o t -- python $R/bench.py --no-cpu-baseline --no-secondary --contigs 6250 --contexts 1 --gen-procs 1 --steps 4 --warmup 2 > $R/gpurun_out/$T/c1.json 2> $R/gpurun_out/$T/c1.err
cd $R
python tools/rocpd_stats.py $(find gpurun_out/$T/trace -name "*.db" | head -1) 2>/dev/null | head -${2:-30}
