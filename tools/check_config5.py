"""BASELINE.json configs[4]: one 200 Mbp contig at 65 % GC, single mode with the full-genome TrainingInfo fixture,
GPU against the CPU oracle (about 25 s of CPU).  Run on the GPU box: gpurun -- 'python tools/check_config5.py'."""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import oracle as orc
from tests.util import golden_path
from pyrodigal_amd import _cabi, benchdata
L = 200_000_000
t0 = time.time(); seq = benchdata.synthetic_contig(L, 0.65, 5); print("gen", time.time() - t0, flush=True)
tinf = orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz"))
ctx = _cabi.Context(0)
ctx.set_models([tinf.tobytes()])
b = ctx.upload([seq])
for it in range(2):
    t0 = time.time(); res = ctx.find_genes(b, meta=False, closed=True); dt = time.time() - t0
    print("GPU: genes", len(res.genes), "nodes", res.contigs[0]["n_nodes"], "ms %.1f dp_ms %.1f Mbp/s %.1f" % (dt * 1e3, res.t_dp_ms, L / dt / 1e6), flush=True)
t0 = time.time(); o = orc.Oracle(seq); o.find_genes_single(tinf, orc.Params(closed=True)); dt = time.time() - t0
og = o.genes()
print("CPU oracle: genes", len(og), "s %.1f" % dt, "identical:", len(og) == len(res.genes) and all(np.array_equal(res.genes[k], og[k]) for k in ("begin", "end", "start_ndx", "stop_ndx")))
