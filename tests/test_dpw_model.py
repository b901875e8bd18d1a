"""The decomposition behind the wave-batch connection-scoring kernel (pyrodigal_amd/csrc/dp_wave.hip), checked on the CPU.

tests/dpw_model.cpp runs the kernel's algorithm -- 64 targets at a time, block / prefix / suffix maxima for the far gene
ends, uniform carries, near steps, candidate chains, in-batch walk -- with plain loops over the lanes, sharing every scalar
routine with the kernel (dpw_core.h).  Here its results are compared bit for bit with the oracle's straightforward
dynamic programme (ref: lib.pyx:1205-1237, _connection.h:94-408) on the reference's fixtures and on synthetic contigs, so
the case analysis is pinned before the HIP code runs (the `-m gpu` tests then pin the kernel itself)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import golden_path, read_fasta, synthetic_contig

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "dpw_model.cpp")
CORE = os.path.join(os.path.dirname(HERE), "pyrodigal_amd", "csrc", "dpw_core.h")
LIB = os.path.join(HERE, "libdpw_model.so")


@pytest.fixture(scope="module")
def model():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(CORE)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB, SRC], check=True)
    L = ctypes.CDLL(LIB)
    vp = ctypes.c_void_p
    L.dpw_model_run.restype = ctypes.c_int
    L.dpw_model_run.argtypes = [ctypes.c_int] + [vp] * 9 + [ctypes.c_double] + [vp] * 5
    return L


def run_model(L, nodes, st_wt):
    n = len(nodes)
    c = lambda a, t: np.ascontiguousarray(a, dtype=t)
    arrs = [c(nodes["ndx"], np.int32), c(nodes["stop_val"], np.int32), c(nodes["type"], np.uint8), c(nodes["strand"], np.int8),
            c(nodes["cscore"], np.float64), c(nodes["sscore"], np.float64), c(nodes["rscore"], np.float64), c(nodes["uscore"], np.float64),
            c(nodes["star_ptr"], np.int32).reshape(-1)]
    score = np.zeros(n, np.float64); traceb = np.zeros(n, np.int32); ov = np.zeros(n, np.int8)
    mi = np.full(1, -1, np.int32); stats = np.zeros(8, np.int64)
    rc = L.dpw_model_run(n, *[a.ctypes.data for a in arrs], float(st_wt), score.ctypes.data, traceb.ctypes.data, ov.ctypes.data,
                         mi.ctypes.data, stats.ctypes.data)
    assert rc == 0
    return score, traceb, ov, int(mi[0]), stats


def check(L, seq, tinf, closed=False, is_meta=False, mask=False):
    o = orc.Oracle(seq, mask=mask) if mask else orc.Oracle(seq)
    o.extract(tinf.trans_table, orc.Params(closed=closed)); o.sort(); o.reset_scores()
    o.score_nodes(tinf, closed, is_meta)
    o.overlapping_starts(tinf, 1, 60)
    before = o.nodes()
    o.dprog_raw(tinf, True)
    ref = o.nodes()
    if len(ref) == 0:
        return 0, np.zeros(8, np.int64)
    score, traceb, ov, mi, stats = run_model(L, before, tinf.st_wt)
    bad = np.flatnonzero(traceb != ref["traceb"])
    assert len(bad) == 0, (bad[:10], traceb[bad[:10]], ref["traceb"][bad[:10]], before["type"][bad[:10]], before["strand"][bad[:10]])
    assert np.array_equal(score.view(np.uint64), ref["score"].view(np.uint64))
    reached = ref["traceb"] != -1
    assert np.array_equal(ov[reached], ref["ov_mark"][reached])
    assert mi == o.find_max_index()
    return len(ref), stats


@pytest.mark.parametrize("name,model_file,closed", [
    ("SRR492066", "SRR492066.training.bin.gz", False),
    ("GCF_001457455.1_NCTC11397_genomic_100kb", "GCF_001457455.1_NCTC11397_genomic_100kb.tinf_closed.bin.gz", True),
    ("MIIJ01000039", "SRR492066.training.bin.gz", False),
    ("KK037166", "GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz", False),
])
def test_model_on_reference_fixtures(model, name, model_file, closed):
    seq = read_fasta(name + ".fna.gz")[0][1]
    tinf = orc.Training.load(golden_path(model_file))
    for is_meta in (False, True):
        n, stats = check(model, seq, tinf, closed=closed, is_meta=is_meta)
        assert n > 1000 and stats[0] == 0            # the generic fallback of the far gene ends never runs on real sequence


def test_model_on_the_full_genome(model):
    # 153 296 nodes: windows of 1000 nodes slide over ~2400 blocks; giant-ORF windows (ref: lib.pyx:1221-1233)
    seq = read_fasta("GCF_001457455.1_NCTC11397_genomic.fna.gz")[0][1]
    tinf = orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz"))
    n, stats = check(model, seq, tinf, closed=True)
    assert n == 153296 and stats[0] == 0 and stats[2] > 0 and stats[3] > 0


@pytest.mark.parametrize("gc", [0.30, 0.42, 0.50, 0.58, 0.66, 0.70])
def test_model_on_synthetic_contigs(model, gc):
    models = [orc.Training.load(golden_path(f)) for f in ("SRR492066.training.bin.gz", "GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz")]
    for k, L in enumerate((300, 2000, 20_000, 50_000, 150_000)):
        seq = synthetic_contig(L, gc, 4000 + 17 * k + int(gc * 100))
        for tinf in models:
            check(model, seq, tinf, is_meta=True, closed=bool(k & 1))


def test_model_reverse_start_two_bases_before_a_reverse_stop(model, monkeypatch):
    # contig 178 of tools/stress_variants.py seed 830022: a reverse start two bases before a reverse stop it may not connect to, and no other
    # reverse stop nearby -- the walk's shortcut for a reverse start without a second schedule word must not apply (see words_of); with the
    # fix switched off the model, which takes the words apart as the assembly does, leaves the oracle exactly where the kernel did
    from pyrodigal_amd import benchdata
    seq = read_fasta("sweep_830022_178.fna.gz")[0][1]
    tinf = orc.Training(benchdata.load_model_set()[7][1])
    n, _ = check(model, seq, tinf, closed=True)
    assert n == 3989
    monkeypatch.setenv("DPW_MODEL_NO_PLAIN_FIX", "1")
    with pytest.raises(AssertionError):
        check(model, seq, tinf, closed=True)


def test_model_on_tiny_and_degenerate_inputs(model):
    tinf = orc.Training.load(golden_path("SRR492066.training.bin.gz"))
    for L in (0, 3, 61, 100, 130, 200, 400, 700, 1000):
        for closed in (False, True):
            check(model, synthetic_contig(L, 0.45, 100 + L), tinf, is_meta=True, closed=closed)
    # gene-dense input: planted ORFs on both strands, many overlapping ends
    rng = np.random.default_rng(9)
    parts = []
    for k in range(400):
        orf = b"ATG" + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 3 * int(rng.integers(30, 200)))).replace(b"TAA", b"TCA").replace(b"TAG", b"TCG").replace(b"TGA", b"TCA") + b"TAA"
        if rng.random() < 0.5:
            orf = orf.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]
        parts.append(orf + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(0, 40)))))
    dense = b"".join(parts)
    for closed in (False, True):
        n, stats = check(model, dense, tinf, closed=closed)
        assert n > 2000
    # masked input
    seq = bytearray(synthetic_contig(30000, 0.5, 3)); seq[5000:5300] = b"N" * 300; seq[20000:20060] = b"N" * 60
    check(model, bytes(seq), tinf, mask=True)
