#!/bin/bash
# after a change of the call's launch order: the suite's finder / stage / distributed tests, the host-side timing of a call, the default line
timeout 900 python -m pytest tests -q -m gpu -x -k "finder or stages or lib_api or fullsize or stress or train" 2>&1 | tail -3
bash tools/r06_timing.sh r06_tm3 | tail -2
timeout 600 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/early.json 2> gpurun_out/early.err
python - <<PY
import json
d=json.loads(open("gpurun_out/early.json").read().strip().splitlines()[-1]); c=d["config"]
print("value %.0f ms/step %.2f b2b %.0f resident %.0f frac %.4f" % (d["value"], d["ms_per_step"], c.get("host_to_host_back_to_back_Mbp_s",0), c.get("resident_Mbp_s",0), d["roofline"]["frac"]))
PY
