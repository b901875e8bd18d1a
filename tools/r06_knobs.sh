#!/bin/bash
# one-context kernel times over a few launch-shape knobs (rocprofv3 --kernel-trace --stats; the kernel named by $2)
export TMPDIR=/tmp; REPO=$(pwd)
run() { # tag kernel env...
  local tag=$1 k=$2; shift 2
  OUT=$REPO/gpurun_out/knob_$tag; mkdir -p $OUT
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python "$REPO/bench.py" --contigs 6250 --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary --steps 4 --warmup 2 > "$OUT/c1.json" 2> "$OUT/c1.err" )
  echo "$tag: $(python tools/rocpd_stats.py "$OUT/trace/t_results.db" | grep "$k" | head -1)"
}
run base k_coding_score_quads A=1
run tn3072 k_coding_score_quads PGA_CS_TASK_NODES=3072
run tn5120 k_coding_score_quads PGA_CS_TASK_NODES=5120
run tn6144 k_coding_score_quads PGA_CS_TASK_NODES=6144
run xcd0 k_dp_wave PGA_DP_XCD=0
run base2 k_dp_wave A=1
run noorder k_dp_wave PGA_DP_NO_ORDER=1
