"""Synthetic workloads and the 16-model metagenomic bin set used by bench.py and the batch tests.

Generators follow SURVEY.md section 8(d): i.i.d. bases with P(G)=P(C)=gc/2 from
``numpy.random.default_rng(seed)``.  The reference's 50 built-in Prodigal models are not part of
its checkout, so meta mode runs on 16 custom bins: the 3 TrainingInfo fixtures of the reference's
test-suite plus 13 models trained on synthetic planted-ORF genomes (tests/golden/make_models.py).
"""
import glob
import gzip
import os

import numpy as np

_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def synthetic_contig(length, gc, seed):
    rng = np.random.default_rng(seed)
    p = [(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]
    return _ACGT[rng.choice(4, size=length, p=p)].tobytes()


_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def planted_contig(length, gc, seed):
    """"Metagenome-like" sequence (the planted-ORF series): spacers of i.i.d. bases (geometric, mean 120) between open reading
    frames on either strand -- ATG, a geometric number of sense codons (mean 300) drawn with the codon's base frequencies, TAA.  Real
    node density (about 0.06 nodes per base) and long real ORFs: where windows, operon steps and start tweaks fire."""
    rng = np.random.default_rng(seed)
    pb = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])          # A C G T
    codons = [(a, b, c) for a in range(4) for b in range(4) for c in range(4)]
    stops = {(3, 0, 0), (3, 0, 2), (3, 2, 0)}                             # TAA TAG TGA (A0 C1 G2 T3)
    sense = [cd for cd in codons if cd not in stops]
    w = np.array([pb[a] * pb[b] * pb[c] for a, b, c in sense]); w /= w.sum()
    sense_arr = np.array(sense, np.uint8)
    parts, n = [], 0
    while n < length:
        sp = rng.geometric(1 / 120.0)
        parts.append(_ACGT[rng.choice(4, size=sp, p=pb)].tobytes()); n += sp
        L = rng.geometric(1 / 300.0)
        body = _ACGT[sense_arr[rng.choice(len(sense), size=L, p=w)].reshape(-1)].tobytes()
        orf = b"ATG" + body + b"TAA"
        if rng.random() < 0.5:
            orf = orf.translate(_COMP)[::-1]
        parts.append(orf); n += len(orf)
    return b"".join(parts)[:length]


def config2(rank=0):
    """One 5 Mbp contig, 50 % GC (BASELINE.json configs[1]); other ranks get their own seed."""
    return [synthetic_contig(5_000_000, 0.50, 1234 + rank)]


def config3(n=1000, length=50_000, seed0=10_000, first=0):
    """n x 50 kbp contigs, GC 30..70 % (BASELINE.json configs[2])."""
    return [synthetic_contig(length, 0.30 + 0.40 * ((first + c) % 41) / 40, seed0 + first + c) for c in range(n)]


def config4_shard(rank, world, n_total=100_000, length=20_000):
    """This rank's contigs of the 100 000 x 20 kbp metagenome-like set (BASELINE.json configs[3])."""
    return [synthetic_contig(length, 0.30 + 0.40 * (c % 41) / 40, 1_000_000 + c) for c in range(rank, n_total, world)]


def config4_spec(n_total=100_000, length=20_000):
    """(length, gc, seed) of every contig of the 100 000 x 20 kbp job, known without generating a base."""
    c = np.arange(n_total)
    return np.full(n_total, length), 0.30 + 0.40 * (c % 41) / 40, 1_000_000 + c


def config5():
    """One 200 Mbp contig at 65 % GC (BASELINE.json configs[4]; run in single mode)."""
    return [synthetic_contig(200_000_000, 0.65, 5)]


def _gen_chunk(args):
    return [(planted_contig if len(a) > 3 and a[3] else synthetic_contig)(int(a[0]), float(a[1]), int(a[2])) for a in args]


def generate(lengths, gcs, seeds, procs=None, planted=False):
    """`synthetic_contig` (or `planted_contig`) for many contigs on several host cores (the generator itself is the spec's, one seed
    per contig, so the result does not depend on how the work is split)."""
    items = [(n, gc, seed, bool(planted)) for n, gc, seed in zip(lengths, gcs, seeds)]
    procs = procs or min(32, os.cpu_count() or 1)
    import sys
    main_mod = sys.modules.get("__main__")
    # spawned workers re-import the main module by path: a script fed through stdin (or an interactive session) has none, its
    # workers would die on start and the pool would wait for them for ever -- generate in this process instead
    if procs <= 1 or len(items) < 64 or not getattr(main_mod, "__file__", None) or not os.path.exists(getattr(main_mod, "__file__", "")):
        return _gen_chunk(items)
    # spawned workers (never forked: the parent may hold a HIP runtime, and a forked copy of one is not usable); they import
    # this module only, which loads nothing but numpy
    import multiprocessing as mp
    step = max(16, len(items) // (procs * 8))
    chunks = [items[i:i + step] for i in range(0, len(items), step)]
    with mp.get_context("spawn").Pool(procs) as pool:
        out = pool.map(_gen_chunk, chunks)
    return [s for part in out for s in part]


def _read(path):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        return f.read()


def load_model_set():
    """16 ``struct _training`` blobs sorted by GC, as (name, bytes) pairs."""
    files = [
        os.path.join(_GOLDEN, "SRR492066.training.bin.gz"),
        os.path.join(_GOLDEN, "GCF_001457455.1_NCTC11397_genomic_100kb.tinf_closed.bin.gz"),
        os.path.join(_GOLDEN, "GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz"),
    ] + sorted(glob.glob(os.path.join(_GOLDEN, "models", "*.tinf.bin.gz")))
    models = [(os.path.basename(f), _read(f)) for f in files]
    models.sort(key=lambda m: np.frombuffer(m[1][:8], np.float64)[0])
    return models
