#!/bin/bash
# one secondary workload alone: bash tools/r06_cfg.sh config3 [trace]
W=${1:-config3}; T=gpurun_out/r06_cfg_$W; mkdir -p $T; R=$(pwd)
timeout 300 python bench.py --workload $W --no-cpu-baseline --no-secondary --contexts 1 --gen-procs 1 --steps 10 --warmup 2 > $T/b.json 2> $T/b.err
python -c "
import json;d=json.load(open('$T/b.json'));print('$W value',d['value'],'ms/step',d['ms_per_step'],'resident ms',d['config']['resident_ms_per_step'],'dp ms',d['roofline']['kernel_ms_per_launch'],'frac',d['roofline']['frac'])"
if [ "${2:-}" = "trace" ]; then
  cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$T/trace -o t -- python $R/bench.py --workload $W --no-cpu-baseline --no-secondary --contexts 1 --gen-procs 1 --steps 4 --warmup 2 > $R/$T/c1.json 2> $R/$T/c1.err
  cd $R; python tools/rocpd_stats.py $(find $T/trace -name "*.db" | head -1) 2>/dev/null | head -16
fi
