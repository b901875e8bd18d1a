"""Randomised sweep on the GPU box: `python tools/stress_sequence.py` -- GC fraction, unknown-base count and masked regions of the sequence stage
against the oracle on random sequences with IUPAC letters, lower case and runs of N of every length, at several `min_mask` values."""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import oracle as orc
from pyrodigal_amd import _cabi
ctx = _cabi.Context(0)
rng = np.random.default_rng(11)
letters = np.frombuffer(b"ACGTacgtNnRYKMSWBDHVX-", np.uint8)
bad = 0; total = 0
for rnd in range(40):
    seqs = []
    for k in range(200):
        L = int(rng.choice([0, 1, 2, 5, 40, 300, 5000, 70000]))
        p = np.r_[np.full(8, 0.11), np.full(len(letters) - 8, 0.12 / (len(letters) - 8))]
        s = letters[rng.choice(len(letters), size=L, p=p / p.sum())].copy()
        for _ in range(int(rng.integers(0, 6))):
            if L > 10:
                at = int(rng.integers(0, L)); n = int(rng.choice([1, 2, 49, 50, 51, 500, 4000]))
                s[at:at + n] = ord("N") if rng.random() < 0.7 else ord("n")
        seqs.append(s.tobytes())
    mm = int(rng.choice([0, 1, 10, 50, 51]))
    r = ctx.nodes_stage(seqs, _cabi.STAGE_SEQUENCE, mask=True, min_mask=mm)
    for i, s in enumerate(seqs):
        o = orc.Oracle(s, mask=True, mask_size=mm)
        unk = sum(1 for c in s if c not in b"ACGTacgt")
        ok = np.array_equal(r.masks[i], o.masks()) and r.contigs["n_unknown"][i] == unk and (len(s) == 0 or r.contigs["gc"][i] == o.gc)
        bad += not ok; total += 1
        if not ok and bad < 4: print("MISMATCH", len(s), mm, r.masks[i][:4], o.masks()[:4], r.contigs["gc"][i], o.gc)
print("sequences", total, "mismatches", bad)
