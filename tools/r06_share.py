"""A rank's share of the 8-GPU job (12 500 contigs) on this GPU, steps back to back: four calls of 3 125 on four contexts (bench.py's plan for a
rank) against two calls of 6 250 dealt to four contexts across steps, and one call of 12 500 on four contexts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyrodigal_amd import benchdata, _cabi
import bench
models = benchdata.load_model_set()
lengths, gcs, seeds = benchdata.config4_spec(100_000)
seqs = benchdata.generate(lengths[:12500], gcs[:12500], seeds[:12500], procs=8)
ctxs = [_cabi.Context(0) for _ in range(4)]
for c in ctxs: c.set_models([m[1] for m in models])
steps = 12
for sub, nctx in ((3125, 4), (6250, 4), (6250, 2), (4167, 4), (12500, 4), (3125, 4), (6250, 4)):
    groups = [seqs[i:i + sub] for i in range(0, len(seqs), sub)]
    lanes = bench.Lanes(ctxs[:nctx])
    call = lambda c, k: c.find_genes_batch(groups[k], meta=True)
    lanes.run_back_to_back(3, len(groups), call, lambda r: None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lanes.run_back_to_back(steps, len(groups), call, lambda r: [x.genes for x in r])
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    print("calls of %5d contigs (%d per step) on %d contexts, steps back to back: %.2f ms per step = %.0f Mbp/s" % (sub, len(groups), nctx, ms, 250e6 / ms / 1e3), flush=True)
    lanes.close()
