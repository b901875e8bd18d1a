#!/bin/bash
T=${1:-r06_occ}
mkdir -p gpurun_out/$T
B="python bench.py --contigs 6250 --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary"
for occ in 4 5 6; do
  PGA_DPW_OCC=$occ timeout 200 $B --steps 4 --warmup 2 > gpurun_out/$T/occ$occ.json 2> gpurun_out/$T/occ$occ.err
  python -c "
import json;d=json.load(open('gpurun_out/$T/occ$occ.json'));print('occ $occ dp ms',d['roofline']['kernel_ms_per_launch'],'frac',d['roofline']['frac'])"
done
