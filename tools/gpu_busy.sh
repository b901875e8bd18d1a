#!/bin/bash
# How busy the device is while eight contexts run the job: union of the kernel intervals over the span of the traced run's last
# second (kernel trace of the default bench, few steps).   gpurun -- 'bash tools/gpu_busy.sh'
REPO=$(pwd); OUT=$REPO/gpurun_out/busy; mkdir -p "$OUT"; export TMPDIR=/tmp
( cd /tmp && timeout -k 5 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -o t -- python "$REPO/bench.py" --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/log.txt" )
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)))
end = iv[-1][1]
for win_ms in (150, 300):
    lo = end - int(win_ms * 1e6)
    cur_s = cur_e = None; busy = 0; ksum = 0
    for s, e in iv:
        if e <= lo: continue
        s = max(s, lo); ksum += e - s
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("last %d ms: device busy (union of kernels) %.1f %%, sum of kernel durations %.2f x the window" % (win_ms, 100.0 * busy / (win_ms * 1e6), ksum / (win_ms * 1e6)))
PY
python -c "
import json; d=json.load(open('$OUT/bench.json')); print('under the profiler: h2h', d['value'], 'resident', d['config']['resident_Mbp_s'])"
