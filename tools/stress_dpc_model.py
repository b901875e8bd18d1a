"""Randomised sweep of the contig-per-wavefront scorer's step function (tests/dpc_model.cpp) against the oracle:
python tools/stress_dpc_model.py [n] [tiny]   (tiny: four-node history, one-entry candidate lists -- the memory paths)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_dpc_model as T
from tests.util import synthetic_contig
from oracle import oracle as orc
from pyrodigal_amd import benchdata

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tiny = len(sys.argv) > 2 and sys.argv[2] == "tiny"
L = T.build("_tiny", ["-DDPC_HIST=4", "-DDPC_REACH=4", "-DDPC_CAND=1"]) if tiny else T.build("", [])
models = [orc.Training(m[1]) for m in benchdata.load_model_set()]
rng = np.random.default_rng(int(time.time()))
tot = np.zeros(8, np.int64); nodes = 0
for k in range(n):
    length = int(rng.choice([200, 900, 3000, 20000, 20000, 50000, 120000]))
    gc = float(rng.uniform(0.25, 0.75))
    seq = synthetic_contig(length, gc, int(rng.integers(1 << 30)))
    if rng.random() < 0.3:                      # plant ORFs: gene-dense input
        s = bytearray(seq)
        for _ in range(length // 600):
            at = int(rng.integers(0, max(1, length - 700))); ln = 3 * int(rng.integers(30, 200))
            orf = bytearray(b"ATG") + bytearray(rng.choice(np.frombuffer(b"ACGT", np.uint8), ln).tobytes()) + bytearray(b"TAA")
            for stop in (b"TAA", b"TAG", b"TGA"):
                for p in range(3, len(orf) - 3, 3):
                    if bytes(orf[p:p + 3]) == stop: orf[p + 1:p + 2] = b"C"
            if rng.random() < 0.5: orf = bytearray(bytes(orf).translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1])
            s[at:at + len(orf)] = orf[:max(0, length - at)]
        seq = bytes(s)
    m = models[int(rng.integers(len(models)))]
    nn, st = T.check(L, seq, m, closed=bool(rng.integers(2)), is_meta=bool(rng.integers(2)))
    tot += st; nodes += nn
assert tot[7] == 0, "a fast routine read beyond the history's reach"
print("ok: %d contigs, %d nodes identical to the oracle; nodes through the slow routine %d, sources read back from memory %d / from the history %d, "
      "list candidates %d" % (n, nodes, tot[0], tot[2], tot[3], tot[4]))
