"""The SQ pipe counters of the connection-scoring kernel (tools/collect_sq_counters.sh passes) into profiles/r06_pmc_traffic.json, next to
its HBM bytes: bench.py turns them into roofline.valu_busy_frac / scalar_busy_frac / salu_insts_per_node_pass / branch_insts_per_node_pass.
usage: python tools/sq_to_traffic.py gpurun_out/prof_sq "6250x20kbp_gc30-70_meta" [kernel]"""
import collections, csv, glob, json, re, sys
src, key = sys.argv[1], sys.argv[2]
kernel = sys.argv[3] if len(sys.argv) > 3 else "k_dp_wave"
agg = collections.defaultdict(list)
for f in glob.glob(src + "/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        if m and m.group(1) == kernel:
            agg[(int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
gmax = max(g for g, _ in agg)
val = {c: sum(v) / len(v) for (g, c), v in agg.items() if g == gmax}
d = json.load(open("profiles/r06_pmc_traffic.json"))
e = d.setdefault(key, {})
e.update(valu_busy_simd_cycles_per_launch=int(4 * val["SQ_ACTIVE_INST_VALU"]), scalar_busy_simd_cycles_per_launch=int(4 * val["SQ_ACTIVE_INST_SCA"]),
         salu_insts_per_launch=int(val["SQ_INSTS_SALU"]), branch_insts_per_launch=int(val["SQ_INSTS_BRANCH"]),
         sq_counters="tools/collect_sq_counters.sh: SQ_ACTIVE_INST_VALU / _SCA count in units of four cycles summed over the SIMDs")
json.dump(d, open("profiles/r06_pmc_traffic.json", "w"), indent=1)
print(key, {k: e[k] for k in ("valu_busy_simd_cycles_per_launch", "scalar_busy_simd_cycles_per_launch", "salu_insts_per_launch", "branch_insts_per_launch")})
