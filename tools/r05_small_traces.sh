#!/bin/bash
# traces of the few-chain workloads: one config-2 call, one config-5 call, a lone 20 kbp call (kernels in launch order)
R=$(pwd); O=$R/gpurun_out/${1:-r05_small}; mkdir -p $O; export TMPDIR=/tmp
bash tools/lone_trace.sh > $O/lone.txt 2>&1; tail -75 $O/lone.txt
for WL in config2 config5; do
  ( cd /tmp && timeout 250 rocprofv3 --kernel-trace --stats -d $O/trace_$WL -o t -- python $R/bench.py --workload $WL --no-cpu-baseline --no-secondary --contexts 1 --gen-procs 1 --steps 3 --warmup 1 > $O/$WL.json 2> $O/$WL.err )
  python tools/call_timeline.py $(find $O/trace_$WL -name "*.db" | head -1) > $O/$WL.timeline.txt 2>&1
  tail -3 $O/$WL.timeline.txt
  python -c "
import json;d=json.load(open('$O/$WL.json'));print('$WL',d['value'],d['ms_per_step'],d['config'].get('resident_ms_per_step'))"
done
