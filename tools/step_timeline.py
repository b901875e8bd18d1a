"""Per-call kernel timeline from a rocprofv3 rocpd database: the kernels of the last find_genes call (from its k_digitize on).
usage: python tools/step_timeline.py <results.db> [n_rows]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,grid_x,workgroup_x,vgpr_count,lds_size from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'k_digitize' in r[0]]
s = idx[-1]; t0 = rows[s][1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
busy = 0
for r in rows[s:s + n]:
    name = r[0].replace('(anonymous namespace)::', '').split('(')[0]
    d = (r[2] - r[1]) / 1e3; busy += d
    if d >= 50:
        print(f"{(r[1]-t0)/1e3:9.1f} {d:8.1f} {name[:34]:34s} grid={r[3]} wg={r[4]} vgpr={r[5]} lds={r[6]}")
last = rows[min(len(rows), s + n) - 1]
print(f"kernels busy {busy/1e3:.2f} ms of {(last[2]-t0)/1e6:.2f} ms wall, {min(len(rows), s+n)-s} launches")
