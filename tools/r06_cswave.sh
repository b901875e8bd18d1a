#!/bin/bash
# k_coding_score_quads over the threshold from which an ORF takes a whole wavefront (PGA_CS_WAVE; default 2048 codons)
export TMPDIR=/tmp; REPO=$(pwd)
for w in ${WS:-2048 1024 512 256}; do
  OUT=$REPO/gpurun_out/cswave_$w; mkdir -p $OUT
  ( cd /tmp && PGA_CS_WAVE=$w rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python "$REPO/bench.py" --contigs 6250 --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary --steps 4 --warmup 2 > "$OUT/c1.json" 2> "$OUT/c1.err" )
  echo "PGA_CS_WAVE=$w: $(python tools/rocpd_stats.py "$OUT/trace/t_results.db" | grep k_coding_score_quads)"
done
