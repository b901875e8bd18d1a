#!/bin/bash
# ms per step of the other workloads under environment settings, on ONE box: bash tools/r06_wl.sh "A=1" "B=2" ...
for E in "$@"; do
  for W in config3 config2 config5; do
    env $E python bench.py --workload $W --contexts 1 --no-cpu-baseline --no-secondary --steps 10 --warmup 3 > gpurun_out/wl.json 2> gpurun_out/wl.err
    python -c "
import json;d=json.loads(open('gpurun_out/wl.json').read().strip().splitlines()[-1]);print('$E', '$W', 'ms/step', d['ms_per_step'], 'resident', d['config'].get('resident_ms_per_step'))"
  done
done
