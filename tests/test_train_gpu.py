"""Training on the device (pga_train / GeneFinder.train): the resulting TrainingInfo is byte-identical to the reference's
own fixtures (ref: tests/test_training_info.py:57-66 `test_train_closed`, tests/test_gene_finder.py:329-345) and to
the oracle at every intermediate stage, for Shine-Dalgarno and motif-based start models."""
import gzip

import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import golden_path, read_fasta

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pyrodigal_amd import _cabi
    c = _cabi.Context(0)
    yield c
    c.close()


def test_train_closed_full_genome_matches_reference_fixture(ctx):
    seq = read_fasta("GCF_001457455.1_NCTC11397_genomic.fna.gz")[0][1]
    want = gzip.open(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz")).read()
    assert ctx.train(seq, closed=True) == want


def test_train_100kb_matches_reference_fixture(ctx):
    seq = read_fasta("GCF_001457455.1_NCTC11397_genomic_100kb.fna.gz")[0][1]
    want = gzip.open(golden_path("GCF_001457455.1_NCTC11397_genomic_100kb.tinf_closed.bin.gz")).read()
    assert ctx.train(seq, closed=True) == want


def test_train_srr492066_matches_reference_fixture(ctx):
    seq = read_fasta("SRR492066.fna.gz")[0][1]
    want = gzip.open(golden_path("SRR492066.training.bin.gz")).read()
    got = ctx.train(seq)
    assert got == want
    f = np.frombuffer(got, np.float64)
    assert f[0] == pytest.approx(0.3010045159434068) and f[3] == pytest.approx(2.6770525781861187)   # gc, bias[0] (test_train_info)


@pytest.mark.parametrize("upto", [1, 2, 3, 0])
@pytest.mark.parametrize("name,kw", [("KK037166", {}), ("SRR492066", {"force_nonsd": True}), ("MIIJ01000039", {"tt": 4}),
                                      ("SRR492066", {"start_weight": 3.0, "min_gene": 120})])
def test_training_stages_match_the_oracle(ctx, name, kw, upto):
    seq = read_fasta(name + ".fna.gz")[0][1]
    tt, fn, sw, mg = kw.get("tt", 11), kw.get("force_nonsd", False), kw.get("start_weight", 4.35), kw.get("min_gene", 90)
    want = orc.Oracle(seq).train(orc.Params(min_gene=mg), force_nonsd=fn, start_weight=sw, tt=tt, upto=upto).tobytes()
    got = ctx.train(seq, translation_table=tt, force_nonsd=fn, start_weight=sw, min_gene=mg, upto=upto)
    assert got == want
    if upto == 0 and name == "KK037166":
        assert np.frombuffer(got[72:76], np.int32)[0] == 0          # this genome trains a motif model (uses_sd == 0)


def test_train_then_find_through_the_host_layer():
    """ref: tests/test_gene_finder.py:101-130 -- train() + find_genes() reproduce the reference's single-mode goldens."""
    from pyrodigal_amd import lib
    from tests.util import parse_prodigal_header
    for name in ("SRR492066", "KK037166"):
        seq = read_fasta(name + ".fna.gz")[0][1]
        finder = lib.GeneFinder()
        tinf = finder.train(seq)
        assert finder.training_info is tinf and tinf.raw.tobytes() == orc.Oracle(seq).train().tobytes()
        genes = finder.find_genes(seq)
        want = read_fasta(name + ".single.faa.gz")
        assert [(g.begin, g.end, g.strand) for g in genes] == [parse_prodigal_header(h)[:3] for h, _ in want]
    # several contigs of one genome are joined with the reference's linker (ref: lib.pyx:5510-5532)
    a, b = seq[:30000], seq[30000:60000]
    t2 = lib.GeneFinder().train(a, b)
    assert t2.raw.tobytes() == orc.Oracle(a + "TTAATTAATTAA" + b + "TTAATTAATTAA").train().tobytes()
    with pytest.raises(ValueError):
        lib.GeneFinder().train(seq[:1000])                       # shorter than MIN_SINGLE_GENOME
    with pytest.warns(UserWarning):
        lib.GeneFinder().train(seq[:50000])                      # shorter than IDEAL_SINGLE_GENOME
    with pytest.raises(RuntimeError):
        lib.GeneFinder(meta=True).train(seq)


def test_train_leaves_the_loaded_model_set_in_place():
    """pga_set_models(bins) ... pga_train ... pga_find_genes(meta) keeps scoring with the bins: the training runs its
    half-trained model through the context's model slot and puts the caller's set back."""
    from pyrodigal_amd import _cabi, benchdata
    models = [m[1] for m in benchdata.load_model_set()]
    seqs = [benchdata.synthetic_contig(30000, gc, 40 + i) for i, gc in enumerate((0.4, 0.55))]
    c = _cabi.Context(0)
    try:
        c.set_models(models)
        want = c.find_genes_batch(seqs, meta=True)
        c.train(read_fasta("SRR492066.fna.gz")[0][1])
        got = c.find_genes_batch(seqs, meta=True)
        assert np.array_equal(want.contigs["model"], got.contigs["model"])
        assert want.genes.tobytes() == got.genes.tobytes()
    finally:
        c.close()
