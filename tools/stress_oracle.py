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
    for meta, mask, closed in ((True, False, False), (False, True, True)):
        mi = int(rng.integers(0, 16))
        ctx.set_models(models if meta else [models[mi]])
        res = ctx.find_genes_batch(seqs, meta=meta, mask=mask, closed=closed)
        def one(i):
            o = orc.Oracle(seqs[i], mask=mask, mask_size=50)
            p = orc.Params(closed=closed)
            ph = o.find_genes_meta(bins, p) if meta else (o.find_genes_single(bins[mi], p), 0)[1]
            og, gg = o.genes(), res.genes_of(i)
            ok = (not meta or ph == res.contigs[i]["model"]) and len(og) == len(gg) and all(np.array_equal(og[k], gg[k]) for k in ("begin", "end", "start_ndx", "stop_ndx"))
            return ok, len(og)
        with ThreadPoolExecutor(64) as ex:
            out = list(ex.map(one, range(len(seqs))))
        bad = [i for i, (ok, _) in enumerate(out) if not ok]
        if bad:
            print("MISMATCH vs oracle: seed", seed, "meta", meta, "contigs", bad[:5]); sys.exit(1)
        ngenes += sum(n for _, n in out)
print("seeds", sys.argv[1], "-", sys.argv[2], "identical to the oracle;", ngenes, "genes; %.0f s" % (time.time() - t0))
