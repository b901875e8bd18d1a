#!/bin/bash
# quick GPU check: selected tests, one-context bench (per-kernel HIP-event figures) and the eight-context line without the extras
# usage: bash tools/r05_quick.sh <tag> ["pytest args"]
T=${1:-r05_q}
TESTS=${2:-tests/test_stages_gpu.py tests/test_dp_gpu.py tests/test_finder_gpu.py}
mkdir -p gpurun_out/$T
timeout 400 python -m pytest $TESTS -x -q -m gpu > gpurun_out/$T/pytest.log 2>&1; tail -3 gpurun_out/$T/pytest.log
R=$(pwd)
B="python $R/bench.py --no-cpu-baseline --no-secondary"
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$T/trace -o t -- $B --contigs 6250 --contexts 1 --gen-procs 1 --steps 4 --warmup 2 > $R/gpurun_out/$T/c1.json 2> $R/gpurun_out/$T/c1.err
cd $R
python tools/rocpd_stats.py $(find gpurun_out/$T/trace -name "*.db" | head -1) 2>/dev/null | head -24
timeout 300 $B > gpurun_out/$T/d8.json 2> gpurun_out/$T/d8.err
python - <<PY
import json
for f in ("c1","d8"):
    try:
        d=json.load(open("gpurun_out/$T/%s.json"%f))
        print(f,"value",d["value"],"ms/step",d["ms_per_step"],"resident",d["config"]["resident_Mbp_s"],"b2b",d["config"]["host_to_host_back_to_back_Mbp_s"],"gather",d["config"]["gather_ms_per_step_rank0"],"dp ms",d["roofline"]["kernel_ms_per_launch"],"frac",d["roofline"]["frac"],d["roofline"].get("frac_in_timed_region"))
    except Exception as e: print(f,"failed",e)
PY
