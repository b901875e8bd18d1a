#!/bin/bash
# phase cycles of the start scorer and the coding score (synchronising debug aids), one 6250-contig call
O=gpurun_out/${1:-r06_ssprof}; mkdir -p $O
B="python bench.py --contigs 6250 --contexts 1 --gen-procs 1 --no-cpu-baseline --no-secondary --steps 1 --warmup 0"
PGA_SS_PROFILE=1 timeout 200 $B > $O/ss.json 2> $O/ss.err; grep "ss-profile" $O/ss.err | tail -3
PGA_CS_PROFILE=1 timeout 200 $B > $O/cs.json 2> $O/cs.err; grep "cs-profile" $O/cs.err | tail -2
