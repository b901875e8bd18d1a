"""The kernels of the LAST device call of a traced run in launch order: duration, grid, start offset (rocprofv3 --kernel-trace db).
usage: python tools/call_timeline.py <t_results.db> [k]   (k: which call, counted from the first; default: the last)"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select s.kernel_name, d.start, d.end, d.grid_size_x * d.grid_size_y from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
def short(n):
    m = re.search(r"(k_\w+?)(I|E|\b)", n) or re.search(r"(__amd\w+)", n)
    return m.group(1) if m else n[:30]
names = [(short(r[0]), (r[2] - r[1]) / 1e3, r[3], r[1], r[2]) for r in rows]
starts = [i for i, n in enumerate(names) if n[0].startswith("k_digitize")]
idx = starts[int(sys.argv[2])] if len(sys.argv) > 2 else starts[-1]
end = starts[starts.index(idx) + 1] if starts.index(idx) + 1 < len(starts) else len(names)
t0 = names[idx][3]; tot = 0.0
for n in names[idx:end]:
    print("%-30s %8.1f us  grid %9d  at %8.1f us" % (n[0], n[1], n[2], (n[3] - t0) / 1e3)); tot += n[1]
print("sum %.1f us over %d launches; span %.1f us; %d calls in the trace" % (tot, end - idx, (names[end - 1][4] - t0) / 1e3, len(starts)))
