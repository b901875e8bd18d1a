"""`k_score_starts` looks for "an edge node of the same ORF" only among the first and last 12 nodes of a contig instead of
scanning the 500 nodes the reference scans (lib.pyx:2413-2434).  That rests on a property of node extraction
(lib.pyx:1905-2117, Prodigal node.c add_nodes): nodes flagged `edge`, and start nodes the scorer converts to edge nodes,
sit on the first or last three positions of the sequence, hence among the first / last few nodes of the sorted list.
Pinned here on the oracle's extraction over random contigs (open ends, unknown runs, masks, several translation tables)."""
import importlib.util
import os

import numpy as np

from oracle import oracle as orc
from tests.util import synthetic_contig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EDGE_SPAN = 12          # PGA_EDGE_SPAN in pyrodigal_amd/csrc/pipeline.hip


def test_edge_nodes_sit_at_the_ends_of_the_node_list():
    spec = importlib.util.spec_from_file_location("make_models", os.path.join(ROOT, "tests", "golden", "make_models.py"))
    mm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mm)
    rng = np.random.default_rng(7)
    worst, seen = 0, 0
    for k in range(600):
        L = int(rng.choice([90, 150, 300, 900, 2500, 7000, 20000]))
        gc = float(rng.uniform(0.2, 0.8))
        s = bytearray(mm.planted_genome(L, gc, 10000 + k) if rng.random() < 0.5 else synthetic_contig(L, gc, 10000 + k))
        if rng.random() < 0.3 and L > 400:
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(0, L - 100)); n = int(rng.choice([1, 3, 49, 50, 200]))
                s[at:at + n] = b"N" * n
        o = orc.Oracle(bytes(s), mask=bool(rng.random() < 0.4), mask_size=50)
        p = orc.Params(closed=False, min_gene=int(rng.choice([60, 90])), min_edge_gene=int(rng.choice([30, 60, 90])))
        o.extract(int(rng.choice([11, 4, 1, 25])), p)
        o.sort()
        nd = o.nodes()
        n = len(nd)
        if n == 0:
            continue
        start = nd["type"] != 3
        conv = start & (nd["edge"] == 0) & (((nd["ndx"] <= 2) & (nd["strand"] == 1)) | ((nd["ndx"] >= L - 3) & (nd["strand"] == -1)))
        e = np.nonzero((nd["edge"] != 0) | conv)[0]
        if len(e):
            seen += 1
            worst = max(worst, int(np.minimum(e, n - 1 - e).max()))
    assert seen > 300
    assert worst < EDGE_SPAN, worst
