"""The reference's thread-pool calling pattern over one GeneFinder, for a sweep of finder settings (contexts) and pool sizes.
usage: python tools/pool_probe.py [contexts,...] [threads,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiprocessing.pool import ThreadPool
from pyrodigal_amd import benchdata, lib

ctx_list = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4,8").split(",")]
thr_list = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "32,128").split(",")]
models = benchdata.load_model_set()
bins = lib.MetagenomicBins([lib.MetagenomicBin(lib.TrainingInfo(raw=b), n) for n, b in models])
import numpy as np
rng = np.random.default_rng(5)
seqs = [benchdata.synthetic_contig(20_000, 0.3 + 0.4 * rng.random(), 1000 + i) for i in range(4000)]
bases = sum(len(s) for s in seqs)
for nctx in ctx_list:
    for nthreads in thr_list:
        finder = lib.GeneFinder(meta=True, metagenomic_bins=bins, keep_nodes=False, contexts=nctx)
        with ThreadPool(nthreads) as pool:
            pool.map(finder.find_genes, seqs[:512])
            best = 0.0
            for rep in range(3):
                finder.stats.update(device_calls=0, sequences=0, max_calls_per_device_call=0)
                t0 = time.perf_counter()
                genes = sum(len(g) for g in pool.map(finder.find_genes, seqs))
                dt = time.perf_counter() - t0
                best = max(best, bases / dt / 1e6)
            st = finder.stats
        print("contexts %d threads %3d: %7.1f Mbp/s (best of 3)  device calls %d, %.1f contigs per call, %.3f ms per device call and context"
              % (nctx, nthreads, best, st["device_calls"], st["sequences"] / max(st["device_calls"], 1), 1e3 * dt * nctx / max(st["device_calls"], 1)), flush=True)
        del finder
