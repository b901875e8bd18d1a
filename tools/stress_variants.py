"""Randomised sweep on the GPU box: `python tools/stress_variants.py SEED0 SEED1 [SECONDS]` -- every seed makes 300 gene-dense / random contigs
(planted ORFs, runs of N) and checks that the tree DP kernels, the scan DP kernel, the wave-batch and lane-per-chain DP kernels (with the LDS form
of the coding score forced), the device tail and the host tail give byte-identical gene records in meta, single+mask+closed and meta+mask modes.
Stops after SECONDS (if given) and reports what it covered."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(sys.path[0])
import importlib.util
from tests.util import synthetic_contig
from pyrodigal_amd import _cabi, benchdata
spec = importlib.util.spec_from_file_location("mm", "tests/golden/make_models.py"); mm = importlib.util.module_from_spec(spec); spec.loader.exec_module(mm)
models = [b for _, b in benchdata.load_model_set()]
ctx = _cabi.Context(0)
tot = 0
t0 = time.time()
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
seeds_done = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    if time.time() - t0 > budget: break
    seeds_done += 1
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
    for meta, mask, closed in ((True, False, False), (False, True, True), (True, True, False)):
        ctx.set_models(models if meta else models[int(rng.integers(0, 16)):][:1])
        res = []
        for env in ({}, {"PGA_DP_KERNEL": "scan", "PGA_TAIL": "host"}, {"PGA_DP_KERNEL": "tree1", "PGA_TAIL": "device"}, {"PGA_DP_KERNEL": "tree3", "PGA_TAIL": "device"},
                    {"PGA_DP_KERNEL": "wave"}, {"PGA_DP_KERNEL": "wave", "PGA_DPW_SCHED": "0", "PGA_CS_LDS": "0"},
                    {"PGA_DP_KERNEL": "wave", "PGA_TP_STEPS": "1", "PGA_STAGE_SHIFT": "5", "PGA_DPW_TOPO_WALK": "1"}):
            for k in ("PGA_DP_KERNEL", "PGA_TAIL", "PGA_CS_LDS", "PGA_TP_STEPS", "PGA_STAGE_SHIFT", "PGA_DPW_TOPO_WALK", "PGA_DPW_SCHED"): os.environ.pop(k, None)
            os.environ.update(env)
            res.append(ctx.find_genes_batch(seqs, meta=meta, mask=mask, closed=closed))
        for r in res[1:]:
            if r.genes.tobytes() != res[0].genes.tobytes() or not np.array_equal(r.contigs["model"], res[0].contigs["model"]):
                print("MISMATCH seed", seed, meta, mask, closed); sys.exit(1)
        tot += len(res[0].genes)
print("seeds", sys.argv[1], "+", seeds_done, ": seven kernel / tail / staging variants x three modes agree on", 900 * seeds_done, "contig runs;", tot, "genes; %.0f s" % (time.time() - t0))
