"""Randomised sweep on the GPU box: `python tools/stress_extract.py` -- node extraction (positions, types, strands, stop_val, edge flags) against
the oracle on short and odd sequences, every translation table, open and closed ends, unusual minimum gene lengths, with masks."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from pyrodigal_amd import _cabi  # noqa: E402

ctx = _cabi.Context(0)
rng = np.random.default_rng(5)
TABLES = [1, 2, 3, 4, 5, 6, 9, 10, 11, 12, 13, 14, 15, 16, 21, 22, 23, 24, 25, 26, 29, 30, 32, 33]
letters = np.frombuffer(b"ACGTN", np.uint8)
bad = total = 0
for rnd in range(120):
    seqs = []
    for k in range(150):
        L = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 11, 30, 63, 64, 65, 200, 1000, 3071, 3072, 3073, 3075, 6200, 20000]))
        gc = float(rng.uniform(0.2, 0.8))
        p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2, 0.0]) * 0.97 + np.array([0, 0, 0, 0, 0.03])
        seqs.append(letters[rng.choice(5, size=L, p=p / p.sum())].tobytes())
    tt = int(rng.choice(TABLES)); closed = bool(rng.random() < 0.5); mask = bool(rng.random() < 0.5)
    mg = int(rng.choice([3, 30, 90, 150])); me = int(rng.choice([4, 20, 60, 120])); mm = int(rng.choice([0, 5, 50]))
    out = ctx.nodes_stage(seqs, _cabi.STAGE_EXTRACT, translation_table=tt, closed=closed, min_gene=mg, min_edge_gene=me, mask=mask, min_mask=mm,
                          max_overlap=min(60, mg))
    for s, nd in zip(seqs, out):
        o = orc.Oracle(s, mask=mask, mask_size=mm)
        o.extract(tt, orc.Params(closed=closed, min_gene=mg, min_edge_gene=me, max_overlap=min(60, mg))); o.sort()
        on = o.nodes()
        ok = nd["n"] == len(on) and all(np.array_equal(nd[f].astype(np.int64), on[f].astype(np.int64)) for f in ("ndx", "stop_val", "type", "strand", "edge"))
        total += 1; bad += not ok
        if not ok and bad < 5:
            print("MISMATCH len", len(s), "tt", tt, "closed", closed, "mask", mask, mm, "min_gene", mg, me, "nodes", nd["n"], len(on))
print("sequences", total, "mismatches", bad)
