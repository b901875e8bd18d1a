#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/stress_r4b
( timeout -k 5 340 python tools/stress_nodes.py 930000 999000 300 2>&1 | tail -4 ) > gpurun_out/stress_r4b/nodes.log 2>&1 &
P1=$!
( timeout -k 5 340 python tools/stress_variants.py 830000 899000 300 2>&1 | tail -4 ) > gpurun_out/stress_r4b/variants.log 2>&1 &
P2=$!
( timeout -k 5 340 python tools/stress_extract.py 300 2>&1 | tail -3 ) > gpurun_out/stress_r4b/extract.log 2>&1 &
P3=$!
wait $P1 $P2 $P3
for f in nodes variants extract; do echo "== $f"; cat gpurun_out/stress_r4b/$f.log; done
