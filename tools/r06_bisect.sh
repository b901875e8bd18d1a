#!/bin/bash
# which build / switch makes the randomised sweeps disagree: every library under variants_so/ with and without PGA_SS_FULL_STOPS
set -u
export TMPDIR=/tmp
O=gpurun_out/r06_bisect; mkdir -p $O
for so in variants_so/*.so; do
  cp $so pyrodigal_amd/libpyrodigal_amd.so
  for fs in 0 1; do
    tag=$(basename $so .so)_fs$fs
    if [ $fs = 1 ]; then export PGA_SS_FULL_STOPS=1; else unset PGA_SS_FULL_STOPS; fi
    ( timeout -k 5 100 python tools/stress_variants.py 830022 830024 80 2>&1 | tail -3 ) > $O/$tag.variants.log 2>&1 &
    P1=$!
    ( timeout -k 5 100 python tools/stress_nodes.py 900000 990000 60 2>&1 | tail -4 ) > $O/$tag.nodes.log 2>&1 &
    P2=$!
    wait $P1 $P2
    echo "== $tag"; cat $O/$tag.variants.log $O/$tag.nodes.log
  done
done
