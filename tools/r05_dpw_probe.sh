#!/bin/bash
# k_dp_wave probes: cycles per batch phase (PGA_DP_PROFILE) and the occupancy sweep, one 6 250-contig call on one context
T=${1:-r05_probe}
mkdir -p gpurun_out/$T
B="python bench.py --contigs 6250 --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary"
PGA_DP_PROFILE=1 timeout 200 $B --steps 1 --warmup 0 > gpurun_out/$T/prof.json 2> gpurun_out/$T/prof.err; grep "dp profile" gpurun_out/$T/prof.err | tail -3
for occ in 4 5 6; do
  PGA_DPW_OCC=$occ timeout 200 $B --steps 4 --warmup 2 > gpurun_out/$T/occ$occ.json 2> gpurun_out/$T/occ$occ.err
  python -c "
import json;d=json.load(open('gpurun_out/$T/occ$occ.json'));print('occ $occ resident ms',d['config']['resident_ms_per_step'],'dp ms',d['roofline']['kernel_ms_per_launch'],'frac',d['roofline']['frac'])"
done
