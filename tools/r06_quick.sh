#!/bin/bash
# quick GPU check of the wave kernel (round 6): wave / stress tests, phase cycles (PGA_DP_PROFILE), one-call bench, instruction counters
# usage: bash tools/r06_quick.sh <tag> [notests] [nocounters]
T=${1:-r06_q}
mkdir -p gpurun_out/$T
B="python bench.py --contigs 6250 --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary"
if [ "${2:-}" != "notests" ]; then
  timeout 600 python -m pytest tests/test_dp_gpu.py tests/test_finder_gpu.py tests/test_stress_gpu.py -x -q -m gpu -k "wave or stress or random" > gpurun_out/$T/pytest_wave.log 2>&1; tail -3 gpurun_out/$T/pytest_wave.log
fi
PGA_DP_PROFILE=1 timeout 200 $B --steps 1 --warmup 0 > gpurun_out/$T/prof.json 2> gpurun_out/$T/prof.err; grep "dp profile" gpurun_out/$T/prof.err | tail -3
timeout 300 $B --steps 4 --warmup 2 > gpurun_out/$T/bench_c1.json 2> gpurun_out/$T/bench_c1.err
python -c "
import json;d=json.load(open('gpurun_out/$T/bench_c1.json'));print('value',d['value'],'resident ms',d['config']['resident_ms_per_step'],'dp ms',d['roofline']['kernel_ms_per_launch'],'frac',d['roofline']['frac']);print({k:v for k,v in d.get('pipeline',{}).items()} if 0 else '')"
if [ "${3:-}" != "nocounters" ]; then
  bash tools/collect_sq_counters.sh 2>&1 | grep -i "dp_wave" | grep "INSTS_VALU\|INSTS_SALU\|INSTS_BRANCH\|INSTS_LDS\|INSTS_SMEM\|ACTIVE_INST_VALU\|ACTIVE_INST_SCA\|WAVE_CYCLES \|WAIT_ANY\|WAIT_INST_ANY\|SQ_WAVES"
fi
