#!/bin/bash
# A/B of the default bench line under an environment setting: bash tools/r06_ab.sh TAG VAR=VALUE [VAR=VALUE ..]  (runs base, setting, base, setting)
T=$1; shift; O=gpurun_out/ab_$T; mkdir -p $O
for r in 1 2; do
  for side in base set; do
    if [ $side = set ]; then E="$@"; else E="PGA_AB_NONE=1"; fi
    env $E python bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 2 > $O/$side$r.json 2> $O/$side$r.err
    python -c "
import json;d=json.loads(open('$O/$side$r.json').read().strip().splitlines()[-1]);print('$side','$E' if '$side'=='set' else '','h2h',d['value'],'ms',d['ms_per_step'],'resident',d['config']['resident_Mbp_s'],'b2b',d['config']['host_to_host_back_to_back_Mbp_s'],'parity',d.get('parity',{}).get('tuples_identical'))"
  done
done
