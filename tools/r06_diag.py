"""Diagnosis of a stress_variants mismatch: python tools/r06_diag.py SEED -- which variant differs from the default on which contig, and who agrees with the oracle."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(sys.path[0])
import importlib.util
from tests.util import synthetic_contig
from pyrodigal_amd import _cabi, benchdata
from oracle import oracle as orc
spec = importlib.util.spec_from_file_location("mm", "tests/golden/make_models.py"); mm = importlib.util.module_from_spec(spec); spec.loader.exec_module(mm)
models = [b for _, b in benchdata.load_model_set()]
ctx = _cabi.Context(0)
seed = int(sys.argv[1])
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
VARS = ({}, {"PGA_DP_KERNEL": "scan", "PGA_TAIL": "host"}, {"PGA_DP_KERNEL": "tree1", "PGA_TAIL": "device"}, {"PGA_DP_KERNEL": "tree3", "PGA_TAIL": "device"},
        {"PGA_DP_KERNEL": "wave"}, {"PGA_DP_KERNEL": "wave", "PGA_DPW_SCHED": "0", "PGA_CS_LDS": "0"},
        {"PGA_DP_KERNEL": "wave", "PGA_TP_STEPS": "1", "PGA_STAGE_SHIFT": "5", "PGA_DPW_TOPO_WALK": "1"},
        {"PGA_DP_KERNEL": "scan"}, {"PGA_TAIL": "host"}, {"PGA_DP_KERNEL": "wave", "PGA_TAIL": "host"})
for meta, mask, closed in ((True, False, False), (False, True, True), (True, True, False)):
    use = models if meta else models[int(rng.integers(0, 16)):][:1]
    ctx.set_models(use)
    res = []
    for env in VARS:
        for k in ("PGA_DP_KERNEL", "PGA_TAIL", "PGA_CS_LDS", "PGA_TP_STEPS", "PGA_STAGE_SHIFT", "PGA_DPW_TOPO_WALK", "PGA_DPW_SCHED"): os.environ.pop(k, None)
        os.environ.update(env)
        res.append(ctx.find_genes_batch(seqs, meta=meta, mask=mask, closed=closed))
    for vi, r in enumerate(res[1:], 1):
        same = r.genes.tobytes() == res[0].genes.tobytes() and np.array_equal(r.contigs["model"], res[0].contigs["model"])
        print("mode", meta, mask, closed, "variant", VARS[vi], "same as default:", same, "genes", len(r.genes), len(res[0].genes))
        if same: continue
        for ci in range(len(seqs)):
            a = res[0].genes_of(ci); b = r.genes_of(ci)
            if a.tobytes() != b.tobytes() or res[0].contigs["model"][ci] != r.contigs["model"][ci]:
                o = orc.Oracle(seqs[ci], mask=mask, mask_size=50)
                trs = [orc.Training(x) for x in use]
                if meta: o.find_genes_meta(trs, orc.Params(closed=closed))
                else: o.find_genes_single(trs[0], orc.Params(closed=closed))
                og = o.genes()
                def eq(g): return len(g) == len(og) and all(np.array_equal(g[k], og[k]) for k in ("begin", "end", "start_ndx", "stop_ndx"))
                print("  contig", ci, "len", len(seqs[ci]), "genes default", len(a), "variant", len(b), "oracle", len(og), "| default == oracle:", eq(a), " variant == oracle:", eq(b))
                n = min(len(a), len(b))
                for k in range(n):
                    if a[k].tobytes() != b[k].tobytes(): print("   first differing gene", k, "\n   default", a[k], "\n   variant", b[k], "\n   oracle ", og[k] if k < len(og) else None); break
print(res[0].contigs.dtype.names)
