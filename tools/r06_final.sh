#!/bin/bash
# everything profiles/r06_c_* comes from, at one commit: the collects, the SQ counters, the GPU suite, the sweeps, the default line
TAG=${1:-r06_c}
bash tools/collect_profiles.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
bash tools/collect_sq_counters.sh > gpurun_out/collect_sq_$TAG.log 2>&1
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -2 gpurun_out/${TAG}_pytest_gpu.log
bash tools/stress_bounded.sh > gpurun_out/${TAG}_stress.log 2>&1; tail -12 gpurun_out/${TAG}_stress.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; cut -c1-300 gpurun_out/${TAG}_bench_default.json
