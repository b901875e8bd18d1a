#!/bin/bash
# call plans of the default line: first / last calls of a pass smaller (PGA_BENCH_RAMP)
O=gpurun_out/r06_ramp; mkdir -p $O
for r in none "0.25,0.5,0.75,1" "0.2,0.4,0.6,0.8" "0.125,0.25,0.5,1" "0.1,0.35,0.6,0.85"; do
  t=$(echo $r | tr ',.' '__')
  if [ $r = none ]; then unset PGA_BENCH_RAMP; else export PGA_BENCH_RAMP=$r; fi
  timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/$t.json 2> $O/$t.err
  python -c "
import json;d=json.load(open('$O/$t.json'));c=d['config'];print('$r','value',d['value'],'ms',d['ms_per_step'],'calls',c['device_calls_per_step_rank0'],'resident',c['resident_Mbp_s'],'b2b',c['host_to_host_back_to_back_Mbp_s'],'dp',d['roofline']['kernel_ms_per_launch'],d['roofline']['frac'])"
done
