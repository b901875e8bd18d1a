"""Randomised sweep on the GPU box: `python tools/stress_train.py SEED0 SEED1` -- device training (pga_train) against the oracle's
training on planted-ORF genomes of random size, GC, translation table and options; the TrainingInfo must be byte-identical."""
import importlib.util
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(sys.path[0])
from oracle import oracle as orc  # noqa: E402
from pyrodigal_amd import _cabi  # noqa: E402
from tests.util import synthetic_contig  # noqa: E402

spec = importlib.util.spec_from_file_location("mm", "tests/golden/make_models.py"); mm = importlib.util.module_from_spec(spec); spec.loader.exec_module(mm)
ctx = _cabi.Context(0)
t0 = time.time(); n = 0; nonsd = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    L = int(rng.choice([20000, 60000, 150000, 400000]))
    gc = float(rng.uniform(0.25, 0.75))
    seq = bytearray(mm.planted_genome(L, gc, seed) if rng.random() < 0.8 else synthetic_contig(L, gc, seed))
    if rng.random() < 0.3:
        at = int(rng.integers(0, L - 500)); seq[at:at + 120] = b"N" * 120
    seq = bytes(seq)
    kw = dict(closed=bool(rng.random() < 0.5), force_nonsd=bool(rng.random() < 0.3), tt=int(rng.choice([11, 11, 4, 1, 25])),
              sw=float(rng.choice([4.35, 3.0])), mask=bool(rng.random() < 0.3))
    want = orc.Oracle(seq, mask=kw["mask"], mask_size=50).train(orc.Params(closed=kw["closed"]), force_nonsd=kw["force_nonsd"], start_weight=kw["sw"], tt=kw["tt"]).tobytes()
    got = ctx.train(seq, closed=kw["closed"], force_nonsd=kw["force_nonsd"], translation_table=kw["tt"], start_weight=kw["sw"], mask=kw["mask"])
    if got != want:
        print("MISMATCH seed", seed, L, gc, kw); sys.exit(1)
    n += 1; nonsd += int(np.frombuffer(got[72:76], np.int32)[0] == 0)
print("seeds", sys.argv[1], "-", sys.argv[2], ":", n, "trainings byte-identical to the oracle (%d motif models); %.0f s" % (nonsd, time.time() - t0))
