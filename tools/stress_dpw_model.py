"""Randomised sweep of the wave-batch decomposition (tests/dpw_model.cpp) against the oracle: python tools/stress_dpw_model.py [n]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_dpw_model as T
from tests.util import synthetic_contig
from oracle import oracle as orc
from pyrodigal_amd import benchdata

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
L = T.model.__wrapped__() if hasattr(T.model, "__wrapped__") else None
if L is None:
    import ctypes, subprocess
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", T.LIB, T.SRC], check=True)
    L = ctypes.CDLL(T.LIB); vp = ctypes.c_void_p
    L.dpw_model_run.restype = ctypes.c_int
    L.dpw_model_run.argtypes = [ctypes.c_int] + [vp] * 9 + [ctypes.c_double] + [vp] * 5
models = [orc.Training(m[1]) for m in benchdata.load_model_set()]
rng = np.random.default_rng(int(time.time()))
tot = np.zeros(8, np.int64); nodes = 0
for k in range(n):
    length = int(rng.choice([200, 900, 3000, 20000, 50000, 120000]))
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
    mx6 = max(int(tot[6]), int(st[6])); tot += st; tot[6] = mx6; nodes += nn
print("ok: %d contigs, %d nodes identical to the oracle; generic far-field fallbacks %d, near steps %d, chain candidates %d + %d" % (n, nodes, tot[0], tot[1], tot[2], tot[3]))
if tot[5]:
    print("fixed-point mode: %.2f rounds per batch on average over %d batches, at most %d" % (tot[4] / tot[5], tot[5], tot[6]))
