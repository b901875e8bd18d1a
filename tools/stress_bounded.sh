#!/bin/bash
# The three randomised sweeps that touch node scoring, 130 s each, side by side (gpurun --timeout 600 -- 'bash tools/stress_bounded.sh');
# results in gpurun_out/stress_r4/*.log, appended by hand to profiles/r04_stress_sweeps.log.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/stress_r4
( timeout -k 5 170 python tools/stress_nodes.py 900000 990000 130 2>&1 | tail -4 ) > gpurun_out/stress_r4/nodes.log 2>&1 &
P1=$!
( timeout -k 5 170 python tools/stress_score.py 130 2>&1 | tail -4 ) > gpurun_out/stress_r4/score.log 2>&1 &
P2=$!
( timeout -k 5 170 python tools/stress_variants.py 800000 890000 130 2>&1 | tail -4 ) > gpurun_out/stress_r4/variants.log 2>&1 &
P3=$!
wait $P1 $P2 $P3
for f in nodes score variants; do echo "== $f"; cat gpurun_out/stress_r4/$f.log; done
