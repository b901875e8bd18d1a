#!/bin/bash
# headline job (100 000 x 20 kbp) over context counts and call sizes: host to host / resident / back to back Gbp/s
mkdir -p gpurun_out/r06_sweep
for spec in "4 6250" "3 6250" "5 6250" "6 6250" "4 3125" "4 4167" "8 3125"; do
  set -- $spec
  timeout 400 python bench.py --no-cpu-baseline --no-secondary --contexts $1 --sub-batch $2 --steps 10 --warmup 2 > gpurun_out/r06_sweep/c$1_s$2.json 2> gpurun_out/r06_sweep/c$1_s$2.err
  python -c "
import json;d=json.load(open('gpurun_out/r06_sweep/c$1_s$2.json'));c=d['config'];print('contexts $1 sub $2: h2h',d['value'],'resident',c['resident_Mbp_s'],'b2b',c['host_to_host_back_to_back_Mbp_s'],'hbm GB',c.get('hbm_in_use_GB'))"
done
