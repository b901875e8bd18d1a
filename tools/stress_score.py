"""Randomised sweep on the GPU box: `python tools/stress_score.py [SECONDS]` -- the scoring stage (Nodes.score: coding scores, RBS bins / upstream motifs,
start scores, edge conversion) and the overlapping-start stage against the oracle, field by field, on short and odd sequences, SD and motif models,
is_meta on and off, open and closed ends."""
import importlib.util
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(sys.path[0])
from oracle import oracle as orc  # noqa: E402
from pyrodigal_amd import _cabi, benchdata  # noqa: E402
from tests.test_stages_gpu import check  # noqa: E402

spec = importlib.util.spec_from_file_location("mm", "tests/golden/make_models.py"); mm = importlib.util.module_from_spec(spec); spec.loader.exec_module(mm)
blobs = [b for _, b in benchdata.load_model_set()]
ctx = _cabi.Context(0)
rng = np.random.default_rng(9)
letters = np.frombuffer(b"ACGTN", np.uint8)
total = 0
import time
t_begin = time.time(); budget = float(sys.argv[1]) if len(sys.argv) > 1 else 1e18      # optional: stop after this many seconds
for rnd in range(160):
    if time.time() - t_begin > budget: break
    seqs = []
    for k in range(100):
        L = int(rng.choice([0, 3, 30, 95, 130, 260, 700, 1499, 1501, 2999, 3001, 9000]))
        gc = float(rng.uniform(0.2, 0.8))
        if rng.random() < 0.6 and L > 200:
            s = bytearray(mm.planted_genome(L, gc, rnd * 1000 + k))
        else:
            p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2, 0.0]) * 0.99 + np.array([0, 0, 0, 0, 0.01])
            s = bytearray(letters[rng.choice(5, size=L, p=p / p.sum())].tobytes())
        seqs.append(bytes(s))
    t = orc.Training(blobs[int(rng.integers(0, 16))])
    closed = bool(rng.random() < 0.5); is_meta = bool(rng.random() < 0.5); stage = int(rng.choice([2, 3]))
    ctx.set_models([t.tobytes()])
    out = ctx.nodes_stage(seqs, stage, closed=closed, is_meta=is_meta)
    for s, nd in zip(seqs, out):
        o = orc.Oracle(s)
        o.extract(t.trans_table, orc.Params(closed=closed)); o.sort(); o.reset_scores(); o.score_nodes(t, closed, is_meta)
        if stage == 3:
            o.overlapping_starts(t, 1, 60)
        try:
            check(nd, o.nodes(), stage)
        except AssertionError as e:
            print("MISMATCH len", len(s), "uses_sd", t.uses_sd, "tt", t.trans_table, "closed", closed, "is_meta", is_meta, "stage", stage, str(e)[:200]); sys.exit(1)
        total += 1
print("sequences", total, "all node fields identical to the oracle")
