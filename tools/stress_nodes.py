"""Randomised sweep on the GPU box: `python tools/stress_nodes.py SEED0 SEED1 [SECONDS]` (the connection scorer and the coding-score form
are drawn per seed: default / wave / wave without the step schedule, LDS tables or per-lane gathers for the coding score) -- every node field of every contig (scores, RBS bins,
motifs, traceback, elimination flags) and every gene against the CPU oracle, in meta and single mode, open and closed ends, with
and without masking."""
import importlib.util
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(sys.path[0])
from oracle import oracle as orc  # noqa: E402
from pyrodigal_amd import _cabi, benchdata  # noqa: E402
from tests.test_finder_gpu import compare_contig  # noqa: E402
from tests.util import synthetic_contig  # noqa: E402

spec = importlib.util.spec_from_file_location("mm", "tests/golden/make_models.py"); mm = importlib.util.module_from_spec(spec); spec.loader.exec_module(mm)
blobs = [b for _, b in benchdata.load_model_set()]
models = [orc.Training(b) for b in blobs]
ctx = _cabi.Context(0)
t0 = time.time(); ngenes = 0; ncontigs = 0
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
kinds = {}
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    if time.time() - t0 > budget: break
    rng = np.random.default_rng(seed)
    kern = [None, 'wave', 'wavedyn'][seed % 3]
    for k_ in ('PGA_DP_KERNEL', 'PGA_CS_LDS', 'PGA_DPW_SCHED'): os.environ.pop(k_, None)
    if kern: os.environ['PGA_DP_KERNEL'] = 'wave'
    if kern == 'wavedyn': os.environ['PGA_DPW_SCHED'] = '0'
    if (seed // 4) % 2: os.environ['PGA_CS_LDS'] = '0'           # the coding score by per-lane table gathers instead of the LDS tables
    kinds[(kern or 'default', (seed // 4) % 2)] = kinds.get((kern or 'default', (seed // 4) % 2), 0) + 1
    seqs = []
    for k in range(120):
        L = int(rng.choice([60, 300, 900, 2500, 7000, 20000], p=[0.05, 0.1, 0.2, 0.25, 0.25, 0.15]))
        gc = float(rng.uniform(0.22, 0.78))
        s = bytearray(mm.planted_genome(L, gc, seed * 1000 + k) if rng.random() < 0.7 else synthetic_contig(L, gc, seed * 1000 + k))
        if rng.random() < 0.2 and L > 1000:
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(0, L - 200)); n = int(rng.choice([1, 3, 49, 50, 200]))
                s[at:at + n] = b"N" * n
        seqs.append(bytes(s))
    for meta in (True, False):
        closed = bool(rng.random() < 0.5); mask = bool(rng.random() < 0.4)
        use = models if meta else [models[int(rng.integers(0, 16))]]
        ctx.set_models([m.tobytes() for m in use])
        res = ctx.find_genes_batch(seqs, meta=meta, closed=closed, mask=mask, want_nodes=True)

        def one(i):
            return compare_contig(res, i, seqs[i], orc.Oracle(seqs[i], mask=mask, mask_size=50), use, meta=meta, closed=closed)
        with ThreadPoolExecutor(32) as ex:
            ngenes += sum(ex.map(one, range(len(seqs))))
        ncontigs += len(seqs)
print("kernels drawn (dp, coding score by gathers): ", kinds)
print("seeds from", sys.argv[1], ":", ncontigs, "contig runs, every node field and", ngenes, "genes identical to the oracle; %.0f s" % (time.time() - t0))
