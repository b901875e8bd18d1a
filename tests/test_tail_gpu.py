"""GPU parity of the parallel traceback tail (tail.inl: path by pointer doubling, per-edge untangling, scans for the
gene list, in-order start-tweak fix-up) -- ref: lib.pyx:1253-1311, 3231-3401, Prodigal dprog.c eliminate_bad_genes.
Long gene-dense contigs and degenerate batches, every node field (traceb / tracef / ov_mark / elim / start scores after
elimination) and every gene against the oracle, and the three tails (parallel, one thread per contig, host threads)
against each other."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests.test_finder_gpu import compare_contig
from tests.util import golden_path, synthetic_contig

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def planted(length, gc, seed):
    spec = importlib.util.spec_from_file_location("make_models", os.path.join(ROOT, "tests", "golden", "make_models.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return bytes(mod.planted_genome(length, gc, seed))


@pytest.fixture(scope="module")
def ctx():
    from pyrodigal_amd import _cabi
    c = _cabi.Context(0)
    yield c
    c.close()


def run_modes(ctx, monkeypatch, seqs, **kw):
    out = {}
    for mode in ("par", "device", "host"):
        monkeypatch.setenv("PGA_TAIL", mode)
        out[mode] = ctx.find_genes_batch(seqs, want_nodes=True, **kw)
    monkeypatch.delenv("PGA_TAIL")
    return out


@pytest.mark.parametrize("gc,seed,closed", [(0.38, 301, False), (0.55, 302, True), (0.68, 303, False)])
def test_long_gene_dense_contig_single_mode(ctx, monkeypatch, gc, seed, closed):
    seq = planted(700_000, gc, seed)
    tinf = orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz"))
    ctx.set_models([tinf.buf])
    runs = run_modes(ctx, monkeypatch, [seq], meta=False, closed=closed)
    n = compare_contig(runs["par"], 0, seq, orc.Oracle(seq), [tinf], meta=False, closed=closed)
    assert n > 300
    assert ctx.dp_stats()["chains"] == 1                       # and the connection scoring ran in segments
    for mode in ("device", "host"):
        assert runs[mode].genes.tobytes() == runs["par"].genes.tobytes(), mode
        for k in ("traceb", "tracef", "ov_mark", "elim"):
            assert np.array_equal(runs[mode].nodes[0][k], runs["par"].nodes[0][k]), (mode, k)
        assert np.array_equal(runs[mode].nodes[0]["sscore"].view(np.uint64), runs["par"].nodes[0]["sscore"].view(np.uint64)), mode


def test_degenerate_batch(ctx, monkeypatch):
    """Empty contigs, contigs without nodes, without genes, one-gene contigs and a long one in the same batch."""
    from pyrodigal_amd import benchdata
    models = [b for _, b in benchdata.load_model_set()]
    ctx.set_models(models)
    bins = [orc.Training(b) for b in models]
    seqs = [b"", b"ACGT", b"N" * 500, synthetic_contig(89, 0.5, 1), synthetic_contig(400, 0.5, 2), planted(3_000, 0.5, 3),
            planted(250_000, 0.45, 4), b"ATG" + b"GCA" * 60 + b"TAA", synthetic_contig(30_000, 0.3, 5), planted(1_200, 0.6, 6)]
    runs = run_modes(ctx, monkeypatch, seqs, meta=True)
    total = 0
    for i, s in enumerate(seqs):
        total += compare_contig(runs["par"], i, s, orc.Oracle(s), bins, meta=True)
    assert total > 150
    for mode in ("device", "host"):
        assert runs[mode].genes.tobytes() == runs["par"].genes.tobytes(), mode
        assert np.array_equal(runs[mode].contigs["model"], runs["par"].contigs["model"]), mode


def test_many_contigs_with_frequent_start_tweaks(ctx, monkeypatch):
    """Random sequence has many alternative starts per ORF: the in-order fix-up of the start tweaks has work to do."""
    from pyrodigal_amd import benchdata
    models = [b for _, b in benchdata.load_model_set()]
    ctx.set_models(models)
    seqs = [synthetic_contig(int(L), gc, 900 + k) for k, (L, gc) in enumerate(
        [(120_000, 0.5), (60_000, 0.42), (200_000, 0.58), (15_000, 0.5), (90_000, 0.35)] * 4)]
    runs = run_modes(ctx, monkeypatch, seqs, meta=True)
    assert len(runs["par"].genes) > 1000
    for mode in ("device", "host"):
        assert runs[mode].genes.tobytes() == runs["par"].genes.tobytes(), mode
    bins = [orc.Training(b) for b in models]
    for i in (0, 2, 7, 13):
        compare_contig(runs["par"], i, seqs[i], orc.Oracle(seqs[i]), bins, meta=True)


def test_short_contigs_one_launch_tail(ctx, monkeypatch):
    """Batches whose chains all fit a workgroup's LDS take the one-launch tail (k_tp_small): every node field and every gene against
    the oracle, and against the many-launch form (PGA_TP_STEPS), the one-thread-per-contig tail and the host tail."""
    from pyrodigal_amd import benchdata
    models = [b for _, b in benchdata.load_model_set()]
    ctx.set_models(models)
    bins = [orc.Training(b) for b in models]
    seqs = [synthetic_contig(3_000 + 977 * (k % 23), 0.30 + 0.40 * (k % 41) / 40, 4000 + k) for k in range(96)]
    seqs += [planted(40_000, 0.45, 7), planted(9_000, 0.62, 8), b"", b"ATG" + b"GCA" * 60 + b"TAA", synthetic_contig(70_000, 0.5, 9)]
    runs = run_modes(ctx, monkeypatch, seqs, meta=True)
    monkeypatch.setenv("PGA_TP_STEPS", "1")
    runs["steps"] = ctx.find_genes_batch(seqs, want_nodes=True, meta=True)          # pointer jumping eight hops per launch (round 6)
    monkeypatch.setenv("PGA_TP_JUMP8", "0")
    runs["steps2"] = ctx.find_genes_batch(seqs, want_nodes=True, meta=True)         # ... and by doubling only
    monkeypatch.delenv("PGA_TP_JUMP8")
    monkeypatch.delenv("PGA_TP_STEPS")
    assert max(c["n_nodes"] for c in runs["par"].contigs) <= 4096           # or the batch would not take the one-launch form
    total = sum(compare_contig(runs["par"], i, s, orc.Oracle(s), bins, meta=True) for i, s in enumerate(seqs))
    assert total > 500
    for mode in ("steps", "steps2", "device", "host"):
        assert runs[mode].genes.tobytes() == runs["par"].genes.tobytes(), mode
        for i in range(len(seqs)):
            for k in ("traceb", "tracef", "ov_mark", "elim"):
                assert np.array_equal(runs[mode].nodes[i][k], runs["par"].nodes[i][k]), (mode, i, k)
            assert np.array_equal(runs[mode].nodes[i]["sscore"].view(np.uint64), runs["par"].nodes[i]["sscore"].view(np.uint64)), (mode, i)
