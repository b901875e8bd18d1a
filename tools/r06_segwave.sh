#!/bin/bash
# segments walked by the wave-batch kernel: tests, then configs 2 and 5 with and without
O=gpurun_out/${1:-r06_segwave}; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_segments_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for wl in config2 config5; do
  for w in 0 1; do
    PGA_DP_SEG_WAVE=$w PGA_DP_SEG_DEBUG=1 timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > $O/$wl.w$w.json 2> $O/$wl.w$w.err
    python -c "
import json;d=json.load(open('$O/$wl.w$w.json'));r=d['roofline'];print('$wl wave=$w ms/step',d['ms_per_step'],'resident',d['config'].get('resident_ms_per_step'),'dp ms',r['kernel_ms_per_launch'],'segments',r.get('segments'),'rejected',r.get('rejected_by_verification'),'serial',r.get('chains_walked_serially'), d.get('parity'))"
    grep "dp-seg" $O/$wl.w$w.err | tail -1
  done
done
