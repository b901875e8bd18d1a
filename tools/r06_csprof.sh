#!/bin/bash
# phase cycles of the coding score over task sizes (PGA_CS_PROFILE, a synchronising debug aid), one 6250-contig call
O=gpurun_out/csprof; mkdir -p $O
B="python bench.py --contigs 6250 --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary --steps 1 --warmup 0"
for tn in 4096 8192 2048; do
  PGA_CS_TASK_NODES=$tn PGA_CS_PROFILE=1 timeout 200 $B > $O/cs$tn.json 2> $O/cs$tn.err; echo "task nodes $tn"; grep "cs-profile" $O/cs$tn.err | tail -2
done
