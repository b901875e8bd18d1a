#!/bin/bash
# (historical: PGA_CS_TASK_MIX was an experiment of the fourth session of round 6 in pga_cs_tasks, measured and removed -- profiles/r06_f_experiments.md; the PGA_CS_TASK_NODES rows still run)
# experiment: task sizes of k_coding_score_quads mixed over the launch (big first, small last)
export TMPDIR=/tmp; REPO=$(pwd)
run() { local tag=$1; shift
  OUT=$REPO/gpurun_out/mix_$tag; mkdir -p $OUT
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python "$REPO/bench.py" --contigs 6250 --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary --steps 4 --warmup 2 > "$OUT/c1.json" 2> "$OUT/c1.err" )
  echo "$tag: $(python tools/rocpd_stats.py "$OUT/trace/t_results.db" | grep k_coding_score_quads | head -1)"
}
run base A=1
run t8192 PGA_CS_TASK_NODES=8192
run t2048 PGA_CS_TASK_NODES=2048
run m8_2_30 PGA_CS_TASK_MIX=8192,2048,30
run m8_2_50 PGA_CS_TASK_MIX=8192,2048,50
run m8_4_40 PGA_CS_TASK_MIX=8192,4096,40
run m6_2_40 PGA_CS_TASK_MIX=6144,2048,40
run m8_1_25 PGA_CS_TASK_MIX=8192,1024,25
