#!/usr/bin/env python3
"""Per-kernel pipe occupancy from rocprofv3 --pmc passes of SQ counters + a kernel trace of the same build.

  python tools/sq_counters.py <dir with */p_counter_collection.csv> <trace t_results.db> <title>  > profiles/<tag>_config4_sq_counters.md

The passes are tools/collect_sq_counters.sh's (three passes of eight SQ counters; never together with a trace).  SQ_ACTIVE_INST_* and
SQ_WAVE_CYCLES / SQ_WAIT_* count in units of four cycles, summed over the chip's SIMDs, so
  busy fraction of a pipe = 4 x SQ_ACTIVE_INST_<pipe> / (1024 SIMDs x kernel cycles at 2.4 GHz),
  share of a wave's resident cycles spent waiting = SQ_WAIT_* / SQ_WAVE_CYCLES,
  waves per SIMD = 4 x SQ_WAVE_CYCLES / (1024 x kernel cycles).
"""
import collections, csv, glob, re, sqlite3, sys


def main():
    src, dbpath, title = sys.argv[1], sys.argv[2], sys.argv[3]
    agg = collections.defaultdict(list)
    for f in glob.glob(src + "/*/p_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_\w+)", r["Kernel_Name"])
            agg[(m.group(1) if m else r["Kernel_Name"][:20], int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    cur = sqlite3.connect(dbpath).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    dur = {}
    for r in cur.execute(f"select s.kernel_name, d.grid_size_x * d.grid_size_y * d.grid_size_z, avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name, d.grid_size_x * d.grid_size_y * d.grid_size_z"):
        m = re.search(r"(k_\w+?)(I|E)", r[0])
        dur[(m.group(1) if m else r[0][:30], r[1])] = r[2]
    rows = []
    for nm, grid in sorted({(k[0], k[1]) for k in agg}):
        us = dur.get((nm, grid), 0.0)
        if us < 100:
            continue
        cyc = us * 1e-6 * 2.4e9 * 1024

        def g(c):
            v = agg.get((nm, grid, c))
            return sum(v) / len(v) if v else float("nan")
        rows.append((nm, grid, us, g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_BRANCH"), g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR"),
                     g("SQ_INSTS_LDS"), 4 * g("SQ_ACTIVE_INST_VALU") / cyc, 4 * g("SQ_ACTIVE_INST_SCA") / cyc,
                     g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 4 * g("SQ_WAVE_CYCLES") / cyc))
    rows.sort(key=lambda r: -r[2])
    print("# %s\n" % title)
    print(__doc__.split("\n\n", 2)[2].strip() + "\n")
    print("Wave-instructions per launch in millions; durations: `rocprofv3 --kernel-trace --stats` of the same build (average, microseconds).\n")
    print("| kernel | grid | avg us | VALU M | SALU M | branch M | VMEM M | LDS M | VALU busy | scalar busy | waiting on s_waitcnt | waiting to issue | waves per SIMD |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %d | %.0f | %.1f | %.1f | %.1f | %.1f | %.1f | %.2f | %.2f | %.2f | %.2f | %.1f |" % (r[0], r[1], r[2], r[3] / 1e6, r[4] / 1e6, r[5] / 1e6,
                                                                                                            r[6] / 1e6, r[7] / 1e6, r[8], r[9], r[10], r[11], r[12]))


if __name__ == "__main__":
    main()
