#!/bin/bash
# quick GPU check of the wave kernel: tests, one-call bench, VALU/SALU counters
T=$1
mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_dp_gpu.py tests/test_finder_gpu.py tests/test_stress_gpu.py -x -q -m gpu -k "wave or stress or random" > gpurun_out/$T/pytest_wave.log 2>&1; tail -3 gpurun_out/$T/pytest_wave.log
timeout 300 python bench.py --contigs 6250 --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary --steps 4 --warmup 2 > gpurun_out/$T/bench_c1.json 2> gpurun_out/$T/bench_c1.err
python -c "
import json;d=json.load(open('gpurun_out/$T/bench_c1.json'));print('value',d['value'],'resident ms',d['config']['resident_ms_per_step'],'dp ms',d['roofline']['kernel_ms_per_launch'],'frac',d['roofline']['frac'])"
bash tools/collect_sq_counters.sh 2>&1 | grep -i "dp_wave" | grep "INSTS_VALU\|INSTS_SALU\|INSTS_BRANCH\|ACTIVE_INST_VALU\|ACTIVE_INST_SCA\|WAVE_CYCLES \|WAIT_ANY\|WAIT_INST_ANY\|SQ_WAVES"
