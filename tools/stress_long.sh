#!/bin/bash
# the three randomised sweeps of tools/stress_bounded.sh over fresh seed ranges, T seconds each (default 1500), side by side
# usage: gpurun --timeout 2400 -- 'bash tools/stress_long.sh [T] [nodes seed] [variants seed]'
set -u
export TMPDIR=/tmp
T=${1:-1500}; S1=${2:-1000000}; S2=${3:-1200000}
O=gpurun_out/stress_long; mkdir -p $O
( timeout -k 5 $((T + 60)) python tools/stress_nodes.py $S1 $((S1 + 190000)) $T 2>&1 | tail -4 ) > $O/nodes.log 2>&1 &
P1=$!
( timeout -k 5 $((T + 60)) python tools/stress_score.py $T 2>&1 | tail -4 ) > $O/score.log 2>&1 &
P2=$!
( timeout -k 5 $((T + 60)) python tools/stress_variants.py $S2 $((S2 + 190000)) $T 2>&1 | tail -4 ) > $O/variants.log 2>&1 &
P3=$!
wait $P1 $P2 $P3
for f in nodes score variants; do echo "== $f"; cat $O/$f.log; done
