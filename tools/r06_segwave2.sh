#!/bin/bash
O=gpurun_out/${1:-r06_segwave2}; mkdir -p $O
for ws in 2048 4096 6144; do
    PGA_DP_SEG_WSLOTS=$ws PGA_DP_SEG_WAVE=1 PGA_DP_SEG_DEBUG=1 timeout 600 python bench.py --workload config5 --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > $O/c5.$ws.json 2> $O/c5.$ws.err
    python -c "
import json;d=json.load(open('$O/c5.$ws.json'));r=d['roofline'];print('config5 wslots=$ws ms/step',d['ms_per_step'],'resident',d['config'].get('resident_ms_per_step'),'dp ms',r['kernel_ms_per_launch'],'segments',r.get('segments'),'rejected',r.get('rejected_by_verification'),'serial',r.get('chains_walked_serially'))"
done
bash tools/quick_trace.sh config5 r6c5w 60 | tail -62
