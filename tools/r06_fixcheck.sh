#!/bin/bash
# after the schedule fix: the DP / finder / stress tests, the sweeps over the seed range that found it, one bench call
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_fix; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_finder_gpu.py tests/test_stress_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
( timeout -k 5 200 python tools/stress_variants.py 830000 890000 150 2>&1 | tail -3 ) > $O/variants.log 2>&1 &
P1=$!
( timeout -k 5 200 python tools/stress_nodes.py 930000 990000 150 2>&1 | tail -4 ) > $O/nodes.log 2>&1 &
P2=$!
( timeout -k 5 200 python tools/stress_score.py 150 2>&1 | tail -4 ) > $O/score.log 2>&1 &
P3=$!
wait $P1 $P2 $P3
for f in variants nodes score; do echo "== $f"; cat $O/$f.log; done
timeout 300 python bench.py --contigs 6250 --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary --steps 4 --warmup 2 > $O/bench_c1.json 2> $O/bench_c1.err
python -c "
import json;d=json.load(open('$O/bench_c1.json'));print('value',d['value'],'resident ms',d['config']['resident_ms_per_step'],'dp ms',d['roofline']['kernel_ms_per_launch'],'frac',d['roofline']['frac'], d.get('parity'))"
