import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyrodigal_amd import _cabi, benchdata
models = [b for _, b in benchdata.load_model_set()]
seqs = [benchdata.synthetic_contig(20000, 0.3 + 0.4 * (c % 41) / 40, 1000000 + c) for c in range(6250)]
for rep in range(3):
    t0 = time.perf_counter(); ctx = _cabi.Context(0); t1 = time.perf_counter()
    ctx.set_models(models); t2 = time.perf_counter()
    ctx.find_genes_batch(seqs, meta=True); t3 = time.perf_counter()
    ctx.find_genes_batch(seqs, meta=True); t4 = time.perf_counter()
    ctx.close(); t5 = time.perf_counter()
    print("rep %d: create %.1f ms  models %.1f  first call %.1f  second call %.1f  close %.1f" % (rep, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), 1e3*(t5-t4)))
