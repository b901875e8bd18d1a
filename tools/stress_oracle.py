"""Randomised sweep on the GPU box: `python tools/stress_oracle.py SEED0 SEED1` -- like stress_variants.py, but every contig is also
run through the CPU oracle (64 host threads) and the gene calls and winning models must be identical."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(sys.path[0])
import importlib.util
from concurrent.futures import ThreadPoolExecutor
from oracle import oracle as orc
from tests.util import synthetic_contig
from pyrodigal_amd import _cabi, benchdata
spec = importlib.util.spec_from_file_location("mm", "tests/golden/make_models.py"); mm = importlib.util.module_from_spec(spec); spec.loader.exec_module(mm)
named = benchdata.load_model_set()
models = [b for _, b in named]
bins = [orc.Training(b) for b in models]
ctx = _cabi.Context(0)
t0 = time.time(); ngenes = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
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
    for meta in (True, False):
        mask, closed = bool(rng.random() < 0.4), bool(rng.random() < 0.5)
        min_gene = int(rng.choice([60, 90, 150])); min_edge = int(rng.choice([30, 60, 90])); max_ov = int(rng.choice([0, 30, 60]))
        mi = int(rng.integers(0, 16))
        single = orc.Training(models[mi]); single.set_trans_table(int(rng.choice([11, 4, 1, 25])))
        ctx.set_models(models if meta else [single.tobytes()])
        res = ctx.find_genes_batch(seqs, meta=meta, mask=mask, closed=closed, min_gene=min_gene, min_edge_gene=min_edge, max_overlap=max_ov)
        def one(i):
            o = orc.Oracle(seqs[i], mask=mask, mask_size=50)
            p = orc.Params(closed=closed, min_gene=min_gene, min_edge_gene=min_edge, max_overlap=max_ov)
            ph = o.find_genes_meta(bins, p) if meta else (o.find_genes_single(single, p), 0)[1]
            og, gg = o.genes(), res.genes_of(i)
            ok = (not meta or ph == res.contigs[i]["model"]) and len(og) == len(gg) and all(np.array_equal(og[k], gg[k]) for k in ("begin", "end", "start_ndx", "stop_ndx"))
            return ok, len(og)
        with ThreadPoolExecutor(64) as ex:
            out = list(ex.map(one, range(len(seqs))))
        bad = [i for i, (ok, _) in enumerate(out) if not ok]
        if bad:
            print("MISMATCH vs oracle: seed", seed, "meta", meta, mask, closed, min_gene, min_edge, max_ov, single.trans_table, "contigs", bad[:5]); sys.exit(1)
        ngenes += sum(n for _, n in out)
print("seeds", sys.argv[1], "-", sys.argv[2], "identical to the oracle;", ngenes, "genes; %.0f s" % (time.time() - t0))
