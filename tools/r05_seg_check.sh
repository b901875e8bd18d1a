#!/bin/bash
# segmented path: its tests, then config 2 / config 5 steps
T=${1:-r05_seg}; mkdir -p gpurun_out/$T
timeout 500 python -m pytest tests/test_dp_segments_gpu.py tests/test_fullsize_properties_gpu.py -x -q -m gpu -k "seg or config2 or config5" > gpurun_out/$T/pytest.log 2>&1; tail -3 gpurun_out/$T/pytest.log
for WL in config2 config5; do
  timeout 250 python bench.py --workload $WL --no-cpu-baseline --no-secondary --contexts 1 --gen-procs 1 --steps 10 --warmup 2 > gpurun_out/$T/$WL.json 2> gpurun_out/$T/$WL.err
  python -c "
import json;d=json.load(open('gpurun_out/$T/$WL.json'));print('$WL',d['value'],d['ms_per_step'],d['config'].get('resident_ms_per_step'),d['roofline']['kernel_ms_per_launch'])"
done
