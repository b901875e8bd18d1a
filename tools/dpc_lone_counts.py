"""One contig-per-wavefront launch on one chain (PGA_DP_KERNEL=contig through the scorer-level call): for rocprofv3 --pmc runs that
attribute wave instructions to node kinds.  python tools/dpc_lone_counts.py <gc> <length>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PGA_DP_KERNEL"] = "contig"
import numpy as np
from oracle import oracle as orc
from pyrodigal_amd import _cabi, benchdata
from tests.util import golden_path, synthetic_contig

gc = float(sys.argv[1]); L = int(sys.argv[2])
tinf = orc.Training.load(golden_path("SRR492066.training.bin.gz"))
seq = synthetic_contig(L, gc, 1234)
o = orc.Oracle(seq)
o.extract(tinf.trans_table, orc.Params(closed=False)); o.sort(); o.reset_scores()
o.score_nodes(tinf, False, True)
o.overlapping_starts(tinf, 1, 60)
ref = o.nodes()
ctx = _cabi.Context(0)
kinds = (ref["strand"] != 1).astype(int) * 2 + (ref["type"] == 3).astype(int)
print("n=%d F5=%d F3=%d R5=%d R3=%d" % (len(ref), (kinds == 0).sum(), (kinds == 1).sum(), (kinds == 2).sum(), (kinds == 3).sum()))
for _ in range(3):
    ctx.score_connections(ref["ndx"], ref["stop_val"], ref["type"], ref["strand"], ref["cscore"], ref["sscore"], ref["rscore"], ref["uscore"], ref["star_ptr"], tinf.st_wt, True)
