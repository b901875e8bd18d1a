"""Where k_dp_wave leaves the oracle on contig K of stress_variants seed SEED (single mode, mask, closed): python tools/r06_dpdiag.py SEED K"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(sys.path[0])
import importlib.util
from tests.util import synthetic_contig
from pyrodigal_amd import _cabi, benchdata
from oracle import oracle as orc
spec = importlib.util.spec_from_file_location("mm", "tests/golden/make_models.py"); mm = importlib.util.module_from_spec(spec); spec.loader.exec_module(mm)
models = [b for _, b in benchdata.load_model_set()]
seed = int(sys.argv[1]); want = int(sys.argv[2])
rng = np.random.default_rng(seed)
seqs = []
for k in range(300):
    L = int(rng.choice([300, 900, 2500, 7000, 20000, 60000], p=[0.1, 0.15, 0.25, 0.25, 0.2, 0.05]))
    gc = float(rng.uniform(0.22, 0.78))
    s = bytearray(mm.planted_genome(L, gc, seed * 1000 + k) if rng.random() < 0.7 else synthetic_contig(L, gc, seed * 1000 + k))
    if rng.random() < 0.2 and L > 1000:
        for _ in range(int(rng.integers(1, 5))):
            at = int(rng.integers(0, L - 200)); n = int(rng.choice([1, 3, 49, 50, 200]))
            s[at:at + n] = b"N" * n
    seqs.append(bytes(s))
tinf = orc.Training(models[int(rng.integers(0, 16))])
o = orc.Oracle(seqs[want], mask=True, mask_size=50)
o.extract(tinf.trans_table, orc.Params(closed=True)); o.sort(); o.reset_scores()
o.score_nodes(tinf, True, False)
o.overlapping_starts(tinf, 1, 60)
o.dprog_raw(tinf, True)
ref = o.nodes(); n = len(ref)
ctx = _cabi.Context(0)
KINDS = {(1, False): "Fstart", (1, True): "Fstop", (-1, False): "Rstart", (-1, True): "Rstop"}
def kind(i): return KINDS[(int(ref["strand"][i]), int(ref["type"][i]) == 3)]
for env in ({"PGA_DP_KERNEL": "wave"}, {"PGA_DP_KERNEL": "wave", "PGA_DPW_SCHED": "0"}, {"PGA_DP_KERNEL": "wave", "PGA_DPW_OCC": "5"}, {"PGA_DP_KERNEL": "wave", "PGA_DPW_OCC": "4"},
            {"PGA_DP_KERNEL": "wave", "PGA_DPW_TOPO_LDS": "0"}, {"PGA_DP_KERNEL": "wave", "PGA_DPW_TOPO_WALK": "1"}, {"PGA_DP_KERNEL": "tree1"}):
    for k in ("PGA_DP_KERNEL", "PGA_DPW_SCHED", "PGA_DPW_OCC", "PGA_DPW_TOPO_LDS", "PGA_DPW_TOPO_WALK"): os.environ.pop(k, None)
    os.environ.update(env)
    score, traceb, ov, mi, ms = ctx.score_connections(ref["ndx"], ref["stop_val"], ref["type"], ref["strand"], ref["cscore"], ref["sscore"],
                                                      ref["rscore"], ref["uscore"], ref["star_ptr"], tinf.st_wt, True)
    bad = np.flatnonzero((traceb != ref["traceb"]) | (score.view(np.uint64) != ref["score"].view(np.uint64)))
    print(env, "nodes", n, "differing", len(bad), "first", bad[:8], "max index", mi, o.find_max_index())
    if len(bad) and env == {"PGA_DP_KERNEL": "wave"}:
        i = int(bad[0])
        print(" first bad node", i, "batch", i // 64, "lane", i % 64, kind(i), "ndx", ref["ndx"][i], "stop_val", ref["stop_val"][i])
        print("  gpu   score %.17g traceb %d ov %d" % (score[i], traceb[i], ov[i]))
        print("  oracle score %.17g traceb %d ov %d" % (ref["score"][i], ref["traceb"][i], ref["ov_mark"][i]))
        for j in sorted(set([int(traceb[i]), int(ref["traceb"][i])])):
            if j >= 0: print("  source", j, "batch", j // 64, "lane", j % 64, kind(j), "ndx", ref["ndx"][j], "stop_val", ref["stop_val"][j], "score %.17g" % ref["score"][j], "traceb", ref["traceb"][j], "star_ptr", ref["star_ptr"][j])
        print("  star_ptr of the node", ref["star_ptr"][i], [(int(p), kind(int(p)), int(ref["ndx"][p]), int(ref["stop_val"][p])) for p in ref["star_ptr"][i] if p >= 0])
        lo = max(0, (i // 64) * 64 - 8)
        for j in range(lo, min(n, i + 6)):
            print("   %5d b%d l%2d %-6s ndx %6d sv %6d cs %9.4f ss %9.4f sc %12.6f tb %5d | gpu sc %12.6f tb %5d %s" % (j, j // 64, j % 64, kind(j), ref["ndx"][j], ref["stop_val"][j], ref["cscore"][j], ref["sscore"][j],
                  ref["score"][j], ref["traceb"][j], score[j], traceb[j], "<<<" if j in bad else ""))
