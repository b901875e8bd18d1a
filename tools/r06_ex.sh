#!/bin/bash
# extraction by tile lists: stage tests, the extraction sweep, kernel trace of one-context config-4 calls
O=gpurun_out/${1:-r06_ex}; mkdir -p $O
timeout 900 python -m pytest tests/test_stages_gpu.py tests/test_finder_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/stress_extract.py > $O/extract.log 2>&1; tail -3 $O/extract.log
bash tools/r06_trace.sh ${1:-r06_ex} 12
