"""BASELINE.json's full-size workloads through properties that need no oracle run: every gene is a well-formed ORF of
its contig under the winning model's genetic code, results do not depend on batch composition or order, repeated calls
are identical, and a sample of contigs is compared with the oracle in full."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

STOPS_11 = {"TAA", "TAG", "TGA"}
STARTS = {"ATG", "GTG", "TTG"}
_COMP = bytes.maketrans(b"ACGT", b"TGCA")


@pytest.fixture(scope="module")
def setup():
    from pyrodigal_amd import _cabi, benchdata
    models = benchdata.load_model_set()
    ctx = _cabi.Context(0)
    ctx.set_models([m[1] for m in models])
    tts = [int(np.frombuffer(m[1][8:12], np.int32)[0]) for m in models]
    yield ctx, models, tts
    ctx.close()


def gene_dna(seq, g):
    s = seq[g["begin"] - 1:g["end"]]
    return s if g["strand"] == 1 else s.translate(_COMP)[::-1]


def check_contig(seq, genes, tt, min_gene=90):
    L = len(seq)
    for g in genes:
        assert 1 <= g["begin"] < g["end"] <= L and g["strand"] in (1, -1)
        dna = gene_dna(seq, g).decode()
        n = g["end"] - g["begin"] + 1
        first_edge = g["partial_begin"] if g["strand"] == 1 else g["partial_end"]
        last_edge = g["partial_end"] if g["strand"] == 1 else g["partial_begin"]
        if not first_edge and not last_edge:
            assert n % 3 == 0 and n >= min_gene
        if not first_edge:
            assert dna[:3] in STARTS and ["ATG", "GTG", "TTG"][g["start_type"]] == dna[:3]
        else:
            assert g["start_type"] == 3
        if not last_edge:
            stop = dna[-3:]
            assert stop in STOPS_11 and not (tt == 4 and stop == "TGA")
        body = dna[3 if not first_edge else n % 3:-3 if not last_edge else None]
        codons = {body[i:i + 3] for i in range(0, len(body) - 2, 3)}
        assert not (codons & ({"TAA", "TAG"} if tt == 4 else STOPS_11))        # no stop inside the reading frame
        assert np.isfinite(g["cscore"]) and np.isfinite(g["sscore"])
    begins = genes["begin"]
    assert np.all(np.diff(begins) >= 0)                                         # genes come in sequence order


def test_config3_full_size(setup):
    from pyrodigal_amd import benchdata
    ctx, models, tts = setup
    seqs = benchdata.config3(1000, 50_000)
    res = ctx.find_genes_batch(seqs, meta=True)
    assert len(res.genes) > 50_000 and res.n_chains > 1000
    for i in range(0, 1000, 7):
        c = res.contigs[i]
        assert c["model"] >= 0
        check_contig(seqs[i], res.genes_of(i), tts[c["model"]])
    # idempotence
    res2 = ctx.find_genes_batch(seqs, meta=True)
    assert res2.genes.tobytes() == res.genes.tobytes() and np.array_equal(res2.contigs["model"], res.contigs["model"])
    # batch composition and order do not matter: a shuffled subset gives the same per-contig genes
    rng = np.random.default_rng(0)
    pick = rng.permutation(1000)[:150]
    sub = ctx.find_genes_batch([seqs[i] for i in pick], meta=True)
    key = ["begin", "end", "strand", "start_ndx", "stop_ndx", "start_type", "cscore", "sscore", "rscore", "uscore", "tscore"]
    for k, i in enumerate(pick):
        a, b = sub.genes_of(k), res.genes_of(i)
        assert sub.contigs[k]["model"] == res.contigs[i]["model"] and len(a) == len(b)
        for name in key:
            assert np.array_equal(a[name], b[name]), name
    # a sample against the oracle, in full
    bins = [orc.Training(m[1]) for m in models]
    for i in pick[:12]:
        o = orc.Oracle(seqs[i])
        assert o.find_genes_meta(bins) == res.contigs[i]["model"]
        og, gg = o.genes(), res.genes_of(i)
        assert len(og) == len(gg) and all(np.array_equal(og[k], gg[k]) for k in ("begin", "end", "start_ndx", "stop_ndx"))
        on = o.nodes()
        for name in ("cscore", "sscore", "rscore", "uscore", "tscore", "mot_score"):      # the full-size run's own node scores, as bit patterns
            assert np.array_equal(gg[name].view(np.uint64), on[name][gg["start_ndx"]].view(np.uint64)), (i, name)
    from tests.test_finder_gpu import compare_contig
    nodes_run = ctx.find_genes_batch([seqs[i] for i in pick[:12]], meta=True, want_nodes=True)
    assert sum(compare_contig(nodes_run, k, seqs[i], orc.Oracle(seqs[i]), bins, meta=True) for k, i in enumerate(pick[:12])) > 100


def test_config2_full_size(setup):
    from pyrodigal_amd import benchdata
    ctx, models, tts = setup
    seq = benchdata.config2(0)[0]
    res = ctx.find_genes_batch([seq], meta=True)
    genes = res.genes_of(0)
    assert len(genes) > 5000
    check_contig(seq, genes, tts[res.contigs[0]["model"]])
    assert ctx.find_genes_batch([seq], meta=True).genes.tobytes() == res.genes.tobytes()
    # the same contig inside a batch of others
    others = benchdata.config3(5, 20_000)
    mixed = ctx.find_genes_batch(others[:2] + [seq] + others[2:], meta=True)
    assert mixed.contigs[2]["model"] == res.contigs[0]["model"]
    a = mixed.genes_of(2)
    for name in ("begin", "end", "strand", "start_ndx", "stop_ndx", "cscore", "sscore"):
        assert np.array_equal(a[name], genes[name]), name
    # the whole contig against the oracle (the segmented scorer on the workload it is benchmarked on): the winning model, every gene,
    # and the scores of every gene's start node as bit patterns
    bins = [orc.Training(m[1]) for m in models]
    o = orc.Oracle(seq)
    assert o.find_genes_meta(bins) == res.contigs[0]["model"]
    og = o.genes()
    assert len(og) == len(genes)
    for name in ("begin", "end", "start_ndx", "stop_ndx"):
        assert np.array_equal(og[name], genes[name]), name
    on = o.nodes()
    assert np.array_equal(on["strand"][genes["start_ndx"]], genes["strand"])
    for name in ("cscore", "sscore", "rscore", "uscore", "tscore", "mot_score"):
        assert np.array_equal(genes[name].view(np.uint64), on[name][genes["start_ndx"]].view(np.uint64)), name


def test_config5_full_size_single_mode(capsys):
    """BASELINE.json configs[4]: one 200 Mbp contig at 65 % GC, single mode with the reference's full-genome TrainingInfo
    fixture (closed ends, as the fixture was trained): every gene tuple against the oracle (about 30 s of CPU)."""
    import time
    from pyrodigal_amd import _cabi, benchdata
    from tests.util import golden_path
    L = 200_000_000
    seq = benchdata.synthetic_contig(L, 0.65, 5)
    tinf = orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz"))
    ctx = _cabi.Context(0)
    try:
        ctx.set_models([tinf.tobytes()])
        b = ctx.upload([seq])
        res = ctx.find_genes(b, meta=False, closed=True)
        t0 = time.perf_counter()
        res2 = ctx.find_genes(b, meta=False, closed=True)
        gpu_s = time.perf_counter() - t0
        seg = ctx.dp_stats()
        b.close()
    finally:
        ctx.close()
    assert res2.genes.tobytes() == res.genes.tobytes()
    genes = res.genes_of(0)
    assert res.contigs[0]["n_nodes"] > 10_000_000 and len(genes) > 100_000
    assert seg["chains"] == 1 and seg["segments"] > 100 and seg["serial"] == 0      # the dense chain was cut, verified, not walked serially
    check_contig(seq, genes[:: max(1, len(genes) // 3000)], 11)
    t0 = time.perf_counter()
    o = orc.Oracle(seq)
    o.find_genes_single(tinf, orc.Params(closed=True))
    cpu_s = time.perf_counter() - t0
    og = o.genes()
    assert len(og) == len(genes)
    for k in ("begin", "end", "start_ndx", "stop_ndx"):
        assert np.array_equal(og[k], genes[k]), k
    with capsys.disabled():
        print("\n[config5] 1 x 200 Mbp @65%% GC single mode: %d nodes, %d genes identical to the oracle; GPU %.1f ms (connection scoring %.1f ms, "
              "%d segments, rejected %s), oracle %.1f s on one core" % (res.contigs[0]["n_nodes"], len(genes), gpu_s * 1e3, res2.t_dp_ms,
                                                                         seg["segments"], seg["rejected"], cpu_s))


def test_config4_one_gpu_share(setup, capsys):
    """BASELINE.json configs[3]: rank 0's share of the 100 000 x 20 kbp job on 8 GPUs (12 500 contigs, seeds 1 000 000 + c),
    in one call: ORF properties on every 25th contig, idempotence, and 56 contigs against the oracle in full."""
    import time
    from pyrodigal_amd import benchdata
    ctx, models, tts = setup
    seqs = benchdata.config4_shard(0, 8)
    assert len(seqs) == 12_500 and all(len(s) == 20_000 for s in seqs[:50])
    res = ctx.find_genes_batch(seqs, meta=True)
    t0 = time.perf_counter()
    res2 = ctx.find_genes_batch(seqs, meta=True)
    gpu_s = time.perf_counter() - t0
    assert res2.genes.tobytes() == res.genes.tobytes() and np.array_equal(res2.contigs["model"], res.contigs["model"])
    assert res.n_chains > 30_000 and len(res.genes) > 100_000
    for i in range(0, len(seqs), 25):
        c = res.contigs[i]
        if c["model"] >= 0:
            check_contig(seqs[i], res.genes_of(i), tts[c["model"]])
    bins = [orc.Training(m[1]) for m in models]
    rng = np.random.default_rng(4)
    sample = sorted(rng.permutation(len(seqs))[:56].tolist())
    n_genes = 0
    for i in sample:
        o = orc.Oracle(seqs[i])
        assert o.find_genes_meta(bins) == res.contigs[i]["model"]
        og, gg = o.genes(), res.genes_of(i)
        n_genes += len(og)
        assert len(og) == len(gg) and all(np.array_equal(og[k], gg[k]) for k in ("begin", "end", "start_ndx", "stop_ndx"))
    assert n_genes > 500
    # ... every f64 field of the genes' start nodes as the FULL-SIZE run left them (bit patterns), and then every field of every
    # node of the sampled contigs, run as a sub-batch with want_nodes
    from tests.test_finder_gpu import compare_contig
    for i in sample:
        o = orc.Oracle(seqs[i])
        o.find_genes_meta(bins)
        on, gg = o.nodes(), res.genes_of(i)
        for name in ("cscore", "sscore", "rscore", "uscore", "tscore", "mot_score"):
            assert np.array_equal(gg[name].view(np.uint64), on[name][gg["start_ndx"]].view(np.uint64)), (i, name)
    sub = ctx.find_genes_batch([seqs[i] for i in sample], meta=True, want_nodes=True)
    assert sum(compare_contig(sub, k, seqs[i], orc.Oracle(seqs[i]), bins, meta=True) for k, i in enumerate(sample)) > 500
    with capsys.disabled():
        print("\n[config4 share] 12 500 x 20 kbp: %d chains, %d node-passes, %d genes; %.1f ms per call incl. upload (connection scoring %.2f ms); "
              "%d sampled contigs / %d genes identical to the oracle" % (res.n_chains, res.node_passes, len(res.genes), gpu_s * 1e3, res2.t_dp_ms,
                                                                        len(sample), n_genes))
