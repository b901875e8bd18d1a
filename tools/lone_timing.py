"""Latency of a lone 20 kbp find_genes call (nobody to share a device call with), and where it goes (PGA_TIMING stage marks)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
from pyrodigal_amd import _cabi, benchdata
ctx = _cabi.Context(0)
ctx.set_models([b for _, b in benchdata.load_model_set()])
seqs = [benchdata.synthetic_contig(20000, 0.3 + 0.4 * (c % 41) / 40, 1000000 + c) for c in range(60)]
for s in seqs[:10]: ctx.find_genes_batch([s], meta=True)
lat = []
for rep in range(5):
    for s in seqs[10:]:
        t = time.perf_counter(); ctx.find_genes_batch([s], meta=True); lat.append(time.perf_counter() - t)
lat.sort(); print("lone call: median %.3f ms, min %.3f ms, p90 %.3f ms over %d calls" % (1e3 * lat[len(lat) // 2], 1e3 * lat[0], 1e3 * lat[len(lat) * 9 // 10], len(lat)))
os.environ["PGA_TIMING"] = "1"
for s in seqs[10:13]: ctx.find_genes_batch([s], meta=True)
