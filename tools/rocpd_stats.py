"""Dump the per-kernel summary of a rocprofv3 (rocpd sqlite) result as a markdown table.
usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db "title" > profiles/NAME.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("# %s\n" % title)
print("Source: `rocprofv3 --kernel-trace --stats` (durations in microseconds).\n")
print("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
for name, calls, tot, avg, pct in rows:
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    short = m.group(1) if m else name.split("(")[0]          # template arguments dropped: k_dp_wave<6> is k_dp_wave
    print("| %s | %d | %.1f | %.1f | %.2f |" % (short, calls, tot, avg, pct))
