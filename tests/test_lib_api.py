"""The Cython host layer (pyrodigal_amd.lib): reference-style API tests.
CPU part: construction / validation / TrainingInfo round trip (ref: tests/test_gene_finder.py:366-424,
tests/test_training_info.py).  GPU part: the reference's single-mode golden test, through GeneFinder."""
import io
import os

import numpy as np
import pytest

from tests.util import golden_path, read_fasta, parse_prodigal_header

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    try:
        from pyrodigal_amd import lib as L
    except ImportError:
        import __graft_entry__
        __graft_entry__.build_cython_host()
        from pyrodigal_amd import lib as L
    return L


def test_constructor_validation_matches_reference(lib):
    t = lib.TrainingInfo.load(golden_path("SRR492066.training.bin.gz"))
    with pytest.raises(ValueError):
        lib.GeneFinder(t, meta=True)                    # ref: lib.pyx:5166-5167
    for kw in ({"min_gene": 0}, {"min_edge_gene": -1}, {"min_mask": -1}, {"max_overlap": -1},
               {"max_overlap": 100, "min_gene": 90}):
        with pytest.raises(ValueError):
            lib.GeneFinder(t, **kw)
    with pytest.raises(ValueError):
        lib.GeneFinder(t, backend="avx")                # only the HIP backend exists here
    with pytest.raises(ValueError):
        lib.GeneFinder(t, min_edge_gene=3)              # one-codon edge genes: the one restriction against the reference
    lib.GeneFinder(t, min_edge_gene=3, closed=True)
    with pytest.raises(RuntimeError):
        lib.GeneFinder().find_genes("ATGC")             # single mode without training info (ref: lib.pyx:5429-5430)
    with pytest.raises(RuntimeError):
        lib.GeneFinder(meta=True).train("A" * 30000)    # ref: lib.pyx:5526-5527
    with pytest.raises(ValueError):
        lib.GeneFinder().train("A" * 30000, translation_table=7)


def test_training_info_roundtrip_and_fields(lib):
    t = lib.TrainingInfo.load(golden_path("SRR492066.training.bin.gz"))
    # ref: tests/test_gene_finder.py:329-345
    assert t.translation_table == 11 and t.uses_sd
    assert t.gc == pytest.approx(0.3010045159434068)
    assert t.start_weight == pytest.approx(4.35)
    assert t.bias[0] == pytest.approx(2.6770525781861187)
    assert t.type_weights[0] == pytest.approx(0.71796361273324)
    buf = io.BytesIO(); t.dump(buf)
    t2 = lib.TrainingInfo.load(io.BytesIO(buf.getvalue()))
    assert np.array_equal(t.raw, t2.raw)
    # the bytes are handed out read-only: only the setters change a model, and they are what marks its device copies stale
    with pytest.raises(ValueError):
        t.raw[16] = 0
    t2.start_weight = 4.0
    assert t2.start_weight == 4.0 and t2 != t
    with pytest.raises(EOFError):
        lib.TrainingInfo.load(io.BytesIO(b"abc"))
    with pytest.raises(ValueError):
        lib.TrainingInfo(0.5, translation_table=7)
    bins = lib.MetagenomicBins([lib.MetagenomicBin(t, "x"), lib.MetagenomicBin(t2, "y")])
    assert len(bins) == 2 and bins[1].description == "y" and len(bins[:1]) == 1
    assert len(lib.METAGENOMIC_BINS) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["SRR492066", "KK037166", "MIIJ01000039"])
def test_find_genes_single_goldens(lib, name):
    """ref: tests/test_gene_finder.py:101-130 (TestSingle*): genes, coordinates, gene_data strings."""
    from oracle import oracle as orc
    seq = read_fasta(name + ".fna.gz")[0][1]
    tinf = lib.TrainingInfo(raw=orc.Oracle(seq).train().tobytes())   # training is host-side, outside this path
    finder = lib.GeneFinder(tinf)
    for s in (seq, seq.encode("ascii")):                             # text and binary inputs (TestSingleTxt / TestSingleBin)
        genes = finder.find_genes(s)
        want_p = read_fasta(name + ".single.faa.gz")
        want_g = read_fasta(name + ".single.fna.gz")
        assert len(genes) == len(want_p)
        for i, (g, (hdr, _), (_, nuc)) in enumerate(zip(genes, want_p, want_g)):
            w = parse_prodigal_header(hdr)
            assert (g.begin, g.end, g.strand) == w[:3]
            assert g.sequence() == nuc
            assert "%d%d" % (g.partial_begin, g.partial_end) == w[3]
            assert g.start_type == w[4] and str(g.rbs_motif) == w[5] and str(g.rbs_spacer) == w[6]
            assert "%.3f" % g.gc_cont == w[7]
            assert g._gene_data(1, i).split(";", 1)[1] == hdr.split(" # ")[4].split(";", 1)[1]
            assert 50.0 <= g.confidence() <= 100.0
    assert genes.training_info is tinf and not genes.meta and genes.metagenomic_bin is None
    assert len(genes.nodes) > 0 and genes.nodes[0].index >= 0


@pytest.mark.gpu
def test_find_genes_meta_custom_bins_and_results_outlive_finder(lib):
    """ref: tests/test_gene_finder.py:302-324 (custom / empty bins) and 353-363 (results outlive the finder)."""
    from oracle import oracle as orc
    from pyrodigal_amd import benchdata
    models = benchdata.load_model_set()
    bins = lib.MetagenomicBins([lib.MetagenomicBin(lib.TrainingInfo(raw=b), n) for n, b in models])
    finder = lib.GeneFinder(meta=True, metagenomic_bins=bins, keep_nodes=False)
    seqs = [benchdata.synthetic_contig(25000, gc, 40 + i) for i, gc in enumerate((0.32, 0.5, 0.68))]
    results = finder.find_genes_batch(seqs)
    del finder
    obins = [orc.Training(b) for _, b in models]
    for s, genes in zip(seqs, results):
        o = orc.Oracle(s)
        phase = o.find_genes_meta(obins)
        assert genes.meta and genes.metagenomic_bin is bins[phase]
        og = o.genes()
        assert [(g.begin, g.end) for g in genes] == [(int(a), int(b)) for a, b in zip(og["begin"], og["end"])]
        assert genes.score == 0.0                       # reference quirk: DP fields are reset after the final re-score
    empty = lib.GeneFinder(meta=True, metagenomic_bins=lib.MetagenomicBins([])).find_genes("ATG" * 500)
    assert len(empty) == 0 and empty.metagenomic_bin is None


def test_stage_level_classes_validate_like_the_reference(lib):
    with pytest.raises(ValueError):
        lib.ConnectionScorer(backend="sse")             # ref: lib.pyx:1418-1435 (unsupported backend)
    assert lib.ConnectionScorer().backend == "hip"
    nodes = lib.Nodes()
    assert len(nodes) == 0 and nodes.sort() is None
    with pytest.raises(ValueError):
        nodes.extract(lib.Sequence("ATG" * 40), translation_table=7)
    with pytest.raises(RuntimeError):
        nodes.score(lib.Sequence("ATG" * 40), lib.TrainingInfo.load(golden_path("SRR492066.training.bin.gz")))


@pytest.mark.gpu
def test_nodes_extract_score_and_connection_scorer(lib):
    """ref: tests/test_nodes.py:28-40 (node counts), tests/test_connection_scorer.py (scorer driven by hand)."""
    from oracle import oracle as orc
    text = read_fasta("SRR492066.fna.gz")[0][1]
    seq = lib.Sequence(text)
    t = lib.TrainingInfo.load(golden_path("SRR492066.training.bin.gz"))
    nodes = lib.Nodes()
    assert nodes.extract(seq, translation_table=11) == 2293
    nodes.sort(); nodes.reset_scores()
    nodes.score(seq, t, is_meta=True)
    o = orc.Oracle(text)
    ot = orc.Training.load(golden_path("SRR492066.training.bin.gz"))
    o.extract(11, orc.Params()); o.sort(); o.reset_scores(); o.score_nodes(ot, False, True)
    on = o.nodes()
    for k in ("cscore", "sscore", "rscore", "uscore", "tscore"):
        assert np.array_equal(nodes.array(k).view(np.uint64), on[k].view(np.uint64)), k
    assert np.array_equal(nodes.array("rbs"), on["rbs"]) and np.array_equal(nodes.array("edge"), on["edge"])
    # whole-array connection scoring of those nodes (star_ptr still zero, as after reset_scores)
    scorer = lib.ConnectionScorer(backend="hip")
    scorer.index(nodes)
    scorer.compute_skippable(0, 500)
    best = scorer.score_connections(nodes, t, final=True)
    o.dprog_raw(ot, True)
    on = o.nodes()
    assert np.array_equal(nodes.array("traceb"), on["traceb"])
    assert np.array_equal(nodes.array("score").view(np.uint64), on["score"].view(np.uint64))
    assert best == o.find_max_index()
    # final=False, the reference's default (ref: lib.pyx:1336-1357): the training pass, driven by the nodes' frame-bias scores
    o2 = orc.Oracle(text)
    t2 = orc.Training(); t2.set_trans_table(11); t2._f64(16)[0] = 4.35
    o2.extract(11, orc.Params()); o2.sort(); o2.record_gc_bias(t2); o2.overlapping_starts(t2, 0, 60)
    before = o2.nodes()
    o2.dprog_raw(t2, False)
    after = o2.nodes()
    tr = lib.Nodes()
    tr.extract(seq, translation_table=11); tr.sort(); tr.reset_scores()
    tr._f["gc_score"] = before["gc_score"].copy(); tr._f["star_ptr"] = before["star_ptr"].copy()
    tt = lib.TrainingInfo(0.5, start_weight=4.35)
    tt.bias = t2.bias
    best = scorer.score_connections(tr, tt)               # final defaults to False
    assert np.array_equal(tr.array("traceb"), after["traceb"]) and (after["traceb"] != -1).sum() > 100
    assert np.array_equal(tr.array("score").view(np.uint64), after["score"].view(np.uint64))
    assert best == o2.find_max_index()
    other = nodes.copy()
    other.clear()
    assert len(other) == 0 and len(nodes) == 2293
    with pytest.raises(ValueError):
        nodes.score(lib.Sequence(text[:5000]), t)       # not the sequence the nodes came from


@pytest.mark.gpu
def test_sequence_properties_and_region_masking(lib):
    """ref: tests/test_sequence.py:8-52, tests/test_mask.py, tests/test_gene_finder.py:237-270 (mask=True)."""
    from oracle import oracle as orc
    from pyrodigal_amd import benchdata
    s = lib.Sequence("ATGCNNNNNNNNNNATGCNNNNNNNNTGC", mask=True, mask_size=0)
    assert [(m.begin, m.end) for m in s.masks] == [(4, 14), (18, 26)]
    assert len(lib.Sequence("ATGCNNNNNNNNNNATGCNNNNNNNNTGC", mask=True, mask_size=10).masks) == 1
    assert lib.Sequence("ATGCNNNNNNNNNNATGCNNNNNNNNTGC", mask=False).masks == []
    assert s.unknown == 18 and s.gc == 6 / 29 and s.gc_known == 6 / 11
    assert repr(lib.Mask(1, 2)).endswith("Mask begin=1 end=2>")
    text = bytearray(benchdata.synthetic_contig(40000, 0.5, 77))
    for at, n in ((3000, 60), (12000, 400), (25000, 49), (31000, 1000)):
        text[at:at + n] = b"N" * n
    text = bytes(text)
    t = lib.TrainingInfo.load(golden_path("GCF_001457455.1_NCTC11397_genomic_100kb.tinf_closed.bin.gz"))
    genes = lib.GeneFinder(t, mask=True).find_genes(text)
    assert [(m.begin, m.end) for m in genes.sequence.masks] == [(3000, 3060), (12000, 12400), (31000, 32000)]
    o = orc.Oracle(text, mask=True, mask_size=50)
    o.find_genes_single(orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic_100kb.tinf_closed.bin.gz")))
    og = o.genes()
    assert [(g.begin, g.end) for g in genes] == [(int(a), int(b)) for a, b in zip(og["begin"], og["end"])]
    for g in genes:                                      # no gene runs across a masked region
        assert not any(m.begin < g.end and g.begin - 1 < m.end for m in genes.sequence.masks)
    # a Sequence built under another masking rule is re-wrapped with the finder's (ref: lib.pyx:5433-5438)
    again = lib.GeneFinder(t, mask=True).find_genes(lib.Sequence(text, mask=False))
    assert [(g.begin, g.end) for g in again] == [(g.begin, g.end) for g in genes] and again.sequence.mask


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["SRR492066", "KK037166", "MIIJ01000039"])
def test_writers_reproduce_reference_output_files(lib, name):
    """Gene.translate / Genes.write_translations / write_genes against the reference's own golden files, byte for
    byte (ref: tests/test_genes.py; goldens written by Prodigal: *.single.faa / *.single.fna)."""
    import gzip
    from oracle import oracle as orc
    hdr, seq = read_fasta(name + ".fna.gz")[0]
    seq_id = hdr.split()[0]
    tinf = lib.TrainingInfo(raw=orc.Oracle(seq).train().tobytes())
    genes = lib.GeneFinder(tinf).find_genes(seq)
    out = io.StringIO()
    n = genes.write_translations(out, seq_id)
    assert out.getvalue() == gzip.open(golden_path(name + ".single.faa.gz"), "rt").read() and n == len(out.getvalue())
    out = io.StringIO()
    genes.write_genes(out, seq_id)
    assert out.getvalue() == gzip.open(golden_path(name + ".single.fna.gz"), "rt").read()
    # translate(): options (ref: lib.pyx:2932-3047)
    g = next(x for x in genes if not x.partial_begin and not x.partial_end)
    full = g.translate()
    assert full.endswith("*") and g.translate(include_stop=False) == full[:-1] and full[0] == "M"
    assert g.translate(translation_table=11) == full
    with pytest.raises(ValueError):
        g.translate(translation_table=7)
    with pytest.warns(UserWarning):
        g.translate(translation_table=4)                 # TGA is not a stop in table 4
    # GFF / scores / GenBank: structure checks (the reference has no single-mode goldens for these)
    out = io.StringIO(); genes.write_gff(out, seq_id)
    lines = out.getvalue().splitlines()
    assert lines[0] == "##gff-version  3" and lines[1].startswith("# Sequence Data: seqnum=1;seqlen=%d;" % len(seq))
    assert 'run_type=Single;model="Ab initio"' in lines[2] and len(lines) == 3 + len(genes)
    f = lines[3].split("\t")
    assert f[0] == seq_id and f[2] == "CDS" and (int(f[3]), int(f[4])) == (genes[0].begin, genes[0].end)
    assert f[5] == "%.1f" % genes[0].score and f[8].startswith("ID=%s_1;partial=" % seq_id) and ";conf=" in f[8]
    # ... and every numeric field of every line against the oracle's nodes and genes
    o = orc.Oracle(seq)
    otinf = orc.Oracle(seq).train()
    o.find_genes_single(otinf)
    on, og = o.nodes(), o.genes()
    recs = orc.gene_records(og, on, otinf)
    assert len(recs) == len(genes)
    for line, rec, g in zip(lines[3:], recs, og):
        f = line.split("\t")
        sn = on[g["start_ndx"]]
        assert (int(f[3]), int(f[4]), f[6]) == (rec[0], rec[1], "+" if rec[2] == 1 else "-") and f[7] == "0"
        assert f[5] == "%.1f" % (sn["cscore"] + sn["sscore"])
        attrs = dict(kv.split("=", 1) for kv in f[8].rstrip(";").split(";"))
        assert (attrs["partial"], attrs["start_type"], attrs["rbs_motif"], attrs["rbs_spacer"], attrs["gc_cont"]) == rec[3:8]
        for key in ("cscore", "sscore", "rscore", "uscore", "tscore"):
            assert attrs[key] == "%.2f" % sn[key], key
        assert attrs["score"] == "%.2f" % (sn["cscore"] + sn["sscore"]) and 50.0 <= float(attrs["conf"]) <= 100.0
    out = io.StringIO(); genes.write_scores(out, seq_id)
    rows = [r for r in out.getvalue().splitlines()[3:] if r]
    starts = genes.nodes.array("type") != 3
    assert len(rows) == int(starts.sum()) and all(len(r.split("\t")) == 13 for r in rows)
    # the score table lists every start node; its numeric columns against the oracle's nodes (same multiset of rows)
    want = sorted(("%.2f" % (n["cscore"] + n["sscore"]), "%.2f" % n["cscore"], "%.2f" % n["sscore"], "%.2f" % n["uscore"],
                   "%.2f" % n["tscore"], "%.3f" % n["gc_cont"]) for n in on if n["type"] != 3)
    got = sorted((c[3], c[4], c[5], c[10], c[11], c[12]) for c in (r.split("\t") for r in rows))
    assert got == want
    import datetime
    out = io.StringIO(); genes.write_genbank(out, seq_id, date=datetime.date(2024, 1, 2))
    gb = out.getvalue()
    assert gb.startswith("LOCUS       %-23s %d bp    DNA     linear   BCT 02-JAN-24\n" % (seq_id, len(seq)))
    assert gb.count("     CDS             ") == len(genes) and gb.rstrip().endswith("//")
    assert '/translation="%s' % genes[0].translate(include_stop=False)[:30] in gb


@pytest.mark.gpu
def test_translate_unknown_bases_and_strictness(lib):
    from pyrodigal_amd import benchdata
    t = lib.TrainingInfo.load(golden_path("GCF_001457455.1_NCTC11397_genomic_100kb.tinf_closed.bin.gz"))
    text = bytearray(benchdata.synthetic_contig(30000, 0.5, 5))
    genes = lib.GeneFinder(t).find_genes(bytes(text))
    g = max(genes, key=lambda x: x.end - x.begin)
    # put one N in a third codon position and a whole unknown codon inside the longest gene, call again
    at = g.begin - 1 + 3 * 20
    codon_at = at if g.strand == 1 else at
    text[codon_at + 2] = ord("N")
    text[at + 30:at + 33] = b"NNN"
    genes2 = lib.GeneFinder(t).find_genes(bytes(text))
    g2 = next(x for x in genes2 if (x.begin, x.end) == (g.begin, g.end))
    strict, loose = g2.translate(), g2.translate(strict=False)
    assert strict.count("X") >= 1 and loose.count("X") <= strict.count("X") and len(strict) == len(loose)
    assert g2.translate(unknown_residue="?").count("?") == strict.count("X")
    assert "N" in g2.sequence()


@pytest.mark.gpu
def test_thread_pool_of_find_genes_calls_rides_the_batch_path(lib):
    """The reference's calling pattern (cli.py:289-302: a ThreadPool mapping `find_genes` over the records; lib.pyx:5424-5446
    is re-entrant): 32 threads x `find_genes` over 2 000 contigs of the config-4 job on ONE finder.  Every result is the
    oracle's, in input order, and the calls did not run one tiny device call each: waiting callers were packed together."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    from pyrodigal_amd import benchdata
    models = benchdata.load_model_set()
    bins = lib.MetagenomicBins([lib.MetagenomicBin(lib.TrainingInfo(raw=b), n) for n, b in models])
    lengths, gcs, seeds = benchdata.config4_spec(2000)
    seqs = benchdata.generate(lengths, gcs, seeds, procs=8)
    finder = lib.GeneFinder(meta=True, metagenomic_bins=bins)

    def call(s):
        genes = finder.find_genes(s)
        return (bins._bins.index(genes.metagenomic_bin) if genes.metagenomic_bin is not None else -1,
                [(g.begin, g.end, g.strand, g.partial_begin, g.partial_end, g.start_type) for g in genes], genes._num_seq)

    with ThreadPoolExecutor(32) as ex:
        got = list(ex.map(call, seqs))
    obins = [orc.Training(b) for _, b in models]

    def expect(s):
        o = orc.Oracle(s)                                   # ctypes releases the GIL inside the C calls
        phase = o.find_genes_meta(obins)
        if phase < 0:
            return phase, []
        recs = orc.gene_records(o.genes(), o.nodes(copy=False), obins[phase])
        return phase, [(b, e, strand, partial[0] == "1", partial[1] == "1", start_type) for b, e, strand, partial, start_type, *_ in recs]

    with ThreadPoolExecutor(min(64, os.cpu_count() or 8)) as ex:
        want = list(ex.map(expect, seqs))
    assert [(m, g) for m, g, _ in got] == want
    assert sorted(n for _, _, n in got) == list(range(1, 2001))          # every call got its own sequence number
    st = finder.stats
    assert st["sequences"] == 2000 and st["device_calls"] < 1000 and st["max_calls_per_device_call"] > 1, st
    # a lone caller still gets a device call of its own, at once
    alone = lib.GeneFinder(meta=True, metagenomic_bins=bins, keep_nodes=False)
    assert [(g.begin, g.end) for g in alone.find_genes(seqs[0])] == [(b, e) for b, e, *_ in want[0][1]]
    assert alone.stats["device_calls"] == 1


@pytest.mark.gpu
def test_a_failing_device_call_only_fails_the_request_that_caused_it(lib, monkeypatch):
    """Waiting `find_genes` calls ride one device call.  When that call fails, every request is retried in a device call of its own:
    the caller whose input made it fail gets the error, the others get their genes -- the reference's calls have private state and
    never fail each other (ref: lib.pyx:5400-5469).  The failure is injected by contig length (PGA_FAULT_CONTIG_LEN)."""
    from concurrent.futures import ThreadPoolExecutor
    from pyrodigal_amd import benchdata
    models = benchdata.load_model_set()
    bins = lib.MetagenomicBins([lib.MetagenomicBin(lib.TrainingInfo(raw=b), n) for n, b in models])
    seqs = [benchdata.synthetic_contig(12_000 + 10 * k, 0.35 + 0.3 * (k % 11) / 10, 7000 + k) for k in range(400)]
    bad = {37, 151, 152, 390}
    for k in bad:
        seqs[k] = seqs[k][:11_111]
    ref = lib.GeneFinder(meta=True, metagenomic_bins=bins, keep_nodes=False)
    want = [[(g.begin, g.end, g.strand) for g in genes] for genes in ref.find_genes_batch(seqs)]
    monkeypatch.setenv("PGA_FAULT_CONTIG_LEN", "11111")
    finder = lib.GeneFinder(meta=True, metagenomic_bins=bins, keep_nodes=False)

    def call(k):
        try:
            return [(g.begin, g.end, g.strand) for g in finder.find_genes(seqs[k])]
        except RuntimeError as e:
            return str(e)

    with ThreadPoolExecutor(32) as ex:
        got = list(ex.map(call, range(len(seqs))))
    for k, g in enumerate(got):
        if k in bad:
            assert isinstance(g, str) and "PGA_FAULT_CONTIG_LEN" in g, (k, g)
        else:
            assert g == want[k], k
    st = finder.stats
    assert st["max_calls_per_device_call"] > 1 and st.get("device_calls_retried_per_request", 0) >= 1, st
    # the finder is as good as new afterwards
    monkeypatch.delenv("PGA_FAULT_CONTIG_LEN")
    assert [(g.begin, g.end, g.strand) for g in finder.find_genes(seqs[37])] == want[37]


@pytest.mark.gpu
def test_reference_edge_cases_through_the_api(lib):
    """ref: tests/test_gene_finder.py:198-234 (TestMeta.test_overflow / test_short_sequences / test_empty_sequence) and 366-387 (the
    same in single mode).  The reference runs them with Prodigal's built-in bins, which are not available offline: here the same
    inputs run with the 16-model bin set and in single mode, the gene calls are checked against the oracle, and the properties the
    reference asserts (a gene running over both edges is an `Edge` start with both partial flags; nothing on sequences too short
    for a gene; empty input) are asserted as the reference does."""
    import textwrap
    from oracle import oracle as orc
    from pyrodigal_amd import benchdata
    models = benchdata.load_model_set()
    bins = lib.MetagenomicBins([lib.MetagenomicBin(lib.TrainingInfo(raw=b), n) for n, b in models])
    obins = [orc.Training(b) for _, b in models]
    # > 180195.SAMN03785337.LFLS01000089 (the reference's test_overflow input: one ORF running over both ends of the sequence)
    seq = textwrap.dedent("""
        AACCAGGGCAATATCAGTACCGCGGGCAATGCAACCCTGACTGCCGGCGGTAACCTGAAC
        AGCACTGGCAATCTGACTGTGGGCGGTGTTACCAACGGCACTGCTACTACTGGCAACATC
        GCACTGACCGGTAACAATGCGCTGAGCGGTCCGGTCAATCTGAATGCGTCGAATGGCACG
        GTGACCTTGAACACGACCGGCAATACCACGCTCGGTAACGTGACGGCACAAGGCAATGTG
        ACGACCAATGTGTCCAACGGCAGTCTGACGGTTACCGGCAATACGACAGGTGCCAACACC
        AACCTCAGTGCCAGCGGCAACCTGACCGTGGGTAACCAGGGCAATATCAGTACCGCAGGC
        AATGCAACCCTGACGGCCGGCGACAACCTGACGAGCACTGGCAATCTGACTGTGGGCGGC
        GTCACCAACGGCACGGCCACCACCGGCAACATCGCGCTGACCGGTAACAATGCACTGGCT
        GGTCCTGTCAATCTGAACGCGCCGAACGGCACCGTGACCCTGAACACAACCGGCAATACC
        ACGCTGGGTAATGTCACCGCACAAGGCAATGTGACGACTAATGTGTCCAACGGCAGCCTG
        ACAGTCGCTGGCAATACCACAGGTGCCAACACCAACCTGAGTGCCAGCGGCAATCTGACC
        GTGGGCAACCAGGGCAATATCAGTACCGCGGGCAATGCAACCCTGACTGCCGGCGGTAAC
        CTGAGC
        """).replace("\n", "")
    meta = lib.GeneFinder(meta=True, metagenomic_bins=bins, closed=False)
    genes = meta.find_genes(seq)
    o = orc.Oracle(seq.encode())
    phase = o.find_genes_meta(obins)
    og = o.genes()
    assert [(g.begin, g.end) for g in genes] == [(int(a), int(b)) for a, b in zip(og["begin"], og["end"])]
    assert (bins._bins.index(genes.metagenomic_bin) if genes.metagenomic_bin is not None else -1) == phase
    for g in genes:
        if g.partial_begin and g.partial_end:
            assert g.start_type == "Edge" and g.begin == 1 and g.end >= len(seq) - 2
    assert any(g.partial_begin and g.partial_end for g in genes)          # the one gene of the reference's test: it runs over both edges
    # sequences too short for a gene, and the empty one: no genes, an exhausted iterator (meta and single mode)
    short = "AATGTAGGAAAAACAGCATTTTCATTTCGCCATTTT"
    single = lib.GeneFinder(lib.TrainingInfo.load(golden_path("SRR492066.training.bin.gz")))
    for finder in (meta, single):
        for i in list(range(1, len(short))) + [0]:
            genes = finder.find_genes(short[:i])
            assert len(genes) == 0
            with pytest.raises(StopIteration):
                next(iter(genes))
    batch = meta.find_genes_batch([short[:i] for i in range(len(short))] + [seq])      # the same through one device call
    assert [len(g) for g in batch[:-1]] == [0] * len(short) and len(batch[-1]) == len(og)


@pytest.mark.gpu
def test_find_genes_is_thread_safe(lib):
    """ref: README.md:105-122 / tests/test_gene_finder.py (ThreadPool use): one finder shared by threads, and one finder
    per thread, give the single-threaded result."""
    from concurrent.futures import ThreadPoolExecutor
    from pyrodigal_amd import benchdata
    t = lib.TrainingInfo.load(golden_path("GCF_001457455.1_NCTC11397_genomic_100kb.tinf_closed.bin.gz"))
    seqs = [benchdata.synthetic_contig(15000 + 500 * i, 0.45 + 0.01 * i, 600 + i) for i in range(12)]
    shared = lib.GeneFinder(t)
    want = [[(g.begin, g.end, g.strand) for g in shared.find_genes(s)] for s in seqs]
    with ThreadPoolExecutor(4) as ex:
        got = list(ex.map(lambda s: [(g.begin, g.end, g.strand) for g in shared.find_genes(s)], seqs))
    assert got == want
    with ThreadPoolExecutor(4) as ex:
        got = list(ex.map(lambda s: [(g.begin, g.end, g.strand) for g in lib.GeneFinder(t).find_genes(s)], seqs))
    assert got == want
    ids = sorted(shared.find_genes(s)._num_seq for s in seqs[:3])
    assert ids == sorted(set(ids))                                   # every call gets its own sequence number


def test_training_info_fields_setters_and_pickling(lib):
    """ref: tests/test_training_info.py (properties, setters, pickle round trip), tests/test_gene_finder.py (pickle)."""
    import pickle
    t = lib.TrainingInfo.load(golden_path("SRR492066.training.bin.gz"))
    assert t.upstream_compositions.shape == (32, 4) and t.motif_weights.shape == (4, 4, 4096) and t.coding_statistics.shape == (4096,)
    t2 = pickle.loads(pickle.dumps(t))
    assert t2 == t and t2 is not t
    t2.uses_sd = False; t2.start_weight = 3.0; t2.bias = (1, 2, 3); t2.missing_motif_weight = -1.5
    t2.rbs_weights = np.arange(28); t2.type_weights = (0.5, 0.25, 0.125)
    assert not t2.uses_sd and t2.start_weight == 3.0 and tuple(t2.bias) == (1.0, 2.0, 3.0) and t2.missing_motif_weight == -1.5
    assert t2.rbs_weights[27] == 27.0 and tuple(t2.type_weights) == (0.5, 0.25, 0.125) and t2 != t
    with pytest.raises(ValueError):
        t2.gc = 1.5
    bins = pickle.loads(pickle.dumps(lib.MetagenomicBins([lib.MetagenomicBin(t, "first"), lib.MetagenomicBin(t2, "second")])))
    assert [b.description for b in bins] == ["first", "second"] and bins[0].training_info == t
    f = pickle.loads(pickle.dumps(lib.GeneFinder(t, closed=True, min_gene=120, max_overlap=30, mask=True, min_mask=40)))
    assert (f.closed, f.min_gene, f.max_overlap, f.mask, f.min_mask) == (True, 120, 30, True, 40) and f.training_info == t
    s = pickle.loads(pickle.dumps(lib.Sequence("ACGTNN", mask=True, mask_size=1)))
    assert str(s) == "ACGTNN" and s.mask and s.mask_size == 1


def test_training_info_setters_bump_the_version(lib):
    t = lib.TrainingInfo.load(golden_path("SRR492066.training.bin.gz"))
    v0 = t._version
    t.start_weight = 4.0
    t.bias = (1.0, 2.0, 3.0)
    assert t._version == v0 + 2


@pytest.mark.gpu
def test_training_info_changed_in_place_is_reloaded(lib):
    """The reference shares `struct _training` by pointer: a setter takes effect at the next find_genes call."""
    from pyrodigal_amd import benchdata
    t = lib.TrainingInfo.load(golden_path("GCF_001457455.1_NCTC11397_genomic_100kb.tinf_closed.bin.gz"))
    seq = benchdata.synthetic_contig(60000, 0.5, 321)
    finder = lib.GeneFinder(t)
    before = [(g.begin, g.end, g.strand, g.sscore) for g in finder.find_genes(seq)]
    t.start_weight = 9.5
    t.type_weights = (0.1, -2.0, -3.0)
    after = [(g.begin, g.end, g.strand, g.sscore) for g in finder.find_genes(seq)]
    fresh = [(g.begin, g.end, g.strand, g.sscore) for g in lib.GeneFinder(t).find_genes(seq)]
    assert after == fresh and after != before
    # same through the stage-level context (Nodes.score)
    s = lib.Sequence(seq)
    n1 = lib.Nodes(); n1.extract(s); n1.sort(); n1.score(s, t)
    t.start_weight = 4.35
    n2 = lib.Nodes(); n2.extract(s); n2.sort(); n2.score(s, t)
    assert not np.array_equal(n1.array("sscore"), n2.array("sscore"))


@pytest.mark.gpu
def test_find_genes_rewraps_a_sequence_with_the_finders_masking(lib):
    """ref: lib.pyx:5433-5438: `Sequence(sequence, mask=self.mask, mask_size=self.min_mask)` whatever was passed."""
    from pyrodigal_amd import benchdata
    t = lib.TrainingInfo.load(golden_path("GCF_001457455.1_NCTC11397_genomic_100kb.tinf_closed.bin.gz"))
    text = bytearray(benchdata.synthetic_contig(40000, 0.5, 77))
    text[10000:10080] = b"N" * 80
    plain = lib.Sequence(bytes(text))                       # built without masking
    masked_finder = lib.GeneFinder(t, mask=True, min_mask=50)
    a = [(g.begin, g.end, g.strand) for g in masked_finder.find_genes(plain)]
    b = [(g.begin, g.end, g.strand) for g in masked_finder.find_genes(bytes(text))]
    assert a == b
    genes = masked_finder.find_genes(plain)
    assert genes.sequence.mask and [(m.begin, m.end) for m in genes.sequence.masks] == [(10000, 10080)]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["SRR492066", "KK037166", "MIIJ01000039"])
def test_device_translation_reproduces_the_reference_proteins(lib, name):
    """the proteins translated on the device while the batch is resident (one thread per codon) are the
    reference's `*.single.faa` files byte for byte, through `Genes.write_translations` and `Gene.translate()`."""
    import gzip
    from oracle import oracle as orc
    hdr, seq = read_fasta(name + ".fna.gz")[0]
    seq_id = hdr.split()[0]
    tinf = lib.TrainingInfo(raw=orc.Oracle(seq).train().tobytes())
    genes = lib.GeneFinder(tinf).find_genes_batch([seq], translate=True)[0]
    out = io.StringIO()
    genes.write_translations(out, seq_id)
    assert out.getvalue() == gzip.open(golden_path(name + ".single.faa.gz"), "rt").read()
    plain = lib.GeneFinder(tinf).find_genes(seq)
    assert [g.translate() for g in genes] == [g.translate() for g in plain]                 # device == host, gene by gene
    assert genes[0].translate(include_stop=False) == plain[0].translate(include_stop=False)   # other arguments: the host path


@pytest.mark.gpu
def test_device_translation_options_and_tables_against_the_host_translation(lib):
    """pga_translate_genes with every option: unknown bases (strict and not), no stop letter, a table with other codons,
    reverse-strand and edge genes, against the host restatement of `Gene.translate` (ref: lib.pyx:2932-3047)."""
    from pyrodigal_amd import _cabi, benchdata
    models = benchdata.load_model_set()
    rng = np.random.default_rng(12)
    seqs = []
    for c in range(30):
        s = bytearray(benchdata.synthetic_contig(4000 + 900 * c, 0.35 + 0.01 * c, 4400 + c))
        for _ in range(12):
            at = int(rng.integers(0, len(s) - 3)); s[at] = ord("N")
        seqs.append(bytes(s))
    ctx = _cabi.Context(0)
    try:
        ctx.set_models([b for _, b in models])
        batch = ctx.upload(seqs)
        res = ctx.find_genes(batch, meta=True)
        bins = lib.MetagenomicBins([lib.MetagenomicBin(lib.TrainingInfo(raw=b), n) for n, b in models])
        host = lib.GeneFinder(meta=True, metagenomic_bins=bins).find_genes_batch(seqs)
        assert sum(len(g) for g in host) == len(res.genes) > 200
        for kw in (dict(), dict(include_stop=False), dict(strict=False), dict(unknown_residue="?", strict=False, include_stop=False)):
            letters, off = ctx.translate_genes(batch, res, **kw)
            k = 0
            for genes in host:
                for g in genes:
                    assert letters[off[k]:off[k + 1]].tobytes().decode() == g.translate(**kw), (k, kw)
                    k += 1
        # another genetic code for every contig (table 4 reads TGA as W; the host warns about the changed stop codons)
        import warnings
        letters, off = ctx.translate_genes(batch, res, tables=[4] * len(seqs))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            k = 0
            for genes in host:
                for g in genes:
                    assert letters[off[k]:off[k + 1]].tobytes().decode() == g.translate(translation_table=4), k
                    k += 1
        with pytest.raises(ValueError):
            ctx.translate_genes(batch, res, tables=[7] * len(seqs))
        batch.close()
    finally:
        ctx.close()
