#!/bin/bash
# host-side stage timing of one-context config-4 calls (PGA_TIMING): the last call's stages
T=${1:-r06_tm}
mkdir -p gpurun_out/$T
PGA_TIMING=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary --contigs 6250 --contexts 1 --gen-procs 1 --steps 3 --warmup 2 > gpurun_out/$T/c1.json 2> gpurun_out/$T/c1.err
grep "pga timing" gpurun_out/$T/c1.err | tail -14 | cut -c1-300
