"""GPU parity of the finder-level C-ABI call (`pga_find_genes_batch`) against the CPU oracle:
gene calls bit-identical, winning model identical, every returned node field bit-identical
(north_star tolerance is 1e-6 on scores; we assert exact equality)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import golden_path, read_fasta, synthetic_contig

pytestmark = pytest.mark.gpu

NODE_F64 = ["cscore", "sscore", "rscore", "uscore", "tscore", "mot_score", "score"]
NODE_INT = ["ndx", "stop_val", "type", "strand", "edge", "traceb", "tracef", "ov_mark", "elim", "mot_ndx", "mot_len",
            "mot_spacer", "mot_spacendx"]


@pytest.fixture(scope="module")
def ctx():
    from pyrodigal_amd import _cabi
    c = _cabi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def models():
    m = [orc.Training.load(golden_path("SRR492066.training.bin.gz")),
         orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic_100kb.tinf_closed.bin.gz")),
         orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz"))]
    kk = orc.Oracle(read_fasta("KK037166.fna.gz")[0][1]).train()      # a non-SD model (uses_sd == 0)
    assert kk.uses_sd == 0
    m.append(kk)
    # same statistics under other GC labels / translation table so that several bins land in every GC window
    for src, gc, tt in [(0, 0.36, 11), (1, 0.47, 11), (3, 0.60, 11), (2, 0.42, 4), (0, 0.33, 4), (1, 0.64, 11)]:
        t = m[src].copy(); t.set_gc(gc); t.set_trans_table(tt); m.append(t)
    return m


def compare_contig(res, i, seq, o, models, meta, closed=False):
    cr = res.contigs[i]
    if meta:
        phase = o.find_genes_meta(models, orc.Params(closed=closed))
        assert cr["model"] == phase
    else:
        o.find_genes_single(models[0], orc.Params(closed=closed))
    og, on = o.genes(), o.nodes()
    gg = res.genes_of(i)
    assert len(gg) == len(og)
    for k in ("begin", "end", "start_ndx", "stop_ndx"):
        assert np.array_equal(gg[k], og[k]), k
    if res.nodes is not None and (not meta or cr["model"] >= 0):
        nd = res.nodes[i]
        assert nd["n"] == len(on)
        for k in NODE_INT:
            assert np.array_equal(nd[k].astype(np.int64), on[k].astype(np.int64)), k
        for k in NODE_F64:
            assert np.array_equal(nd[k].view(np.uint64), on[k].view(np.uint64)), k
        assert np.array_equal(nd["gc_cont"].view(np.uint32), on["gc_cont"].view(np.uint32))
        assert np.array_equal(nd["rbs"], on["rbs"])
        if not meta:
            assert np.array_equal(nd["star_ptr"], on["star_ptr"])
    # gene attributes as Gene properties would read them
    if len(gg):
        s = on[og["start_ndx"]]
        assert np.array_equal(gg["strand"], s["strand"])
        assert np.array_equal(gg["cscore"].view(np.uint64), s["cscore"].view(np.uint64))
        assert np.array_equal(gg["sscore"].view(np.uint64), s["sscore"].view(np.uint64))
        assert np.array_equal(gg["start_type"], np.where(s["edge"] != 0, 3, s["type"]))
    return len(gg)


def test_single_mode_goldens_through_gpu(ctx):
    # same goldens the oracle is pinned on (ref: tests/test_gene_finder.py:101-130), now via the HIP path
    from tests.util import parse_prodigal_header
    for name in ["SRR492066", "KK037166", "MIIJ01000039"]:
        seq = read_fasta(name + ".fna.gz")[0][1]
        tinf = orc.Oracle(seq).train()
        ctx.set_models([tinf.buf])
        res = ctx.find_genes_batch([seq], meta=False, want_nodes=True)
        want = [parse_prodigal_header(h) for h, _ in read_fasta(name + ".single.faa.gz")]
        gg = res.genes_of(0)
        assert len(gg) == len(want)
        for g, w in zip(gg, want):
            assert (g["begin"], g["end"], g["strand"]) == w[:3]
            assert "%d%d" % (g["partial_begin"], g["partial_end"]) == w[3]
            assert orc.NODE_TYPE[g["start_type"]] == w[4]
            assert "%.3f" % g["gc_cont"] == w[7]
        compare_contig(res, 0, seq, orc.Oracle(seq), [tinf], meta=False)


def test_single_mode_full_genome_closed(ctx):
    seq = read_fasta("GCF_001457455.1_NCTC11397_genomic.fna.gz")[0][1]
    tinf = orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz"))
    ctx.set_models([tinf.buf])
    res = ctx.find_genes_batch([seq], meta=False, closed=True, want_nodes=True)
    n = compare_contig(res, 0, seq, orc.Oracle(seq), [tinf], meta=False, closed=True)
    assert n > 2000


def test_meta_mode_fixture_contigs(ctx, models):
    ctx.set_models([m.buf for m in models])
    seqs = [read_fasta(n + ".fna.gz")[0][1] for n in ("SRR492066", "KK037166", "GCF_001457455.1_NCTC11397_genomic_100kb")]
    res = ctx.find_genes_batch(seqs, meta=True, want_nodes=True)
    for i, s in enumerate(seqs):
        assert compare_contig(res, i, s, orc.Oracle(s), models, meta=True) > 0


def test_meta_mode_synthetic_batch_mixed_gc(ctx, models):
    ctx.set_models([m.buf for m in models])
    seqs = [synthetic_contig(20000 + 997 * c, 0.30 + 0.40 * (c % 41) / 40, 10000 + c) for c in range(48)]
    res = ctx.find_genes_batch(seqs, meta=True, want_nodes=True)
    total = sum(compare_contig(res, i, s, orc.Oracle(s), models, meta=True) for i, s in enumerate(seqs))
    assert total > 0
    assert res.node_passes > 0 and res.t_dp_ms > 0


def test_meta_mode_short_fragments_and_edge_cases(ctx, models):
    # short fragments exercise the < 3000 bp meta penalties, edge genes, empty and sub-codon inputs
    # (ref: tests/test_gene_finder.py:198-234)
    ctx.set_models([m.buf for m in models])
    seqs = [b"", b"A", b"AT", b"ATG", b"ATGTAA", synthetic_contig(61, 0.5, 1), synthetic_contig(130, 0.4, 2)]
    seqs += [synthetic_contig(L, gc, 500 + L) for L in (300, 700, 1400, 1600, 2900, 3100) for gc in (0.35, 0.55)]
    seqs += [b"N" * 500, synthetic_contig(900, 0.5, 7) + b"NNNNNNNNNN" * 12 + synthetic_contig(900, 0.5, 8)]
    for closed in (False, True):
        res = ctx.find_genes_batch(seqs, meta=True, closed=closed, want_nodes=True)
        for i, s in enumerate(seqs):
            compare_contig(res, i, s, orc.Oracle(s), models, meta=True, closed=closed)


def _with_unknown_runs(length, gc, seed, runs):
    """A synthetic contig with runs of N written over it at fixed relative places."""
    rng = np.random.default_rng(seed)
    s = bytearray(synthetic_contig(length, gc, seed))
    for k, n in enumerate(runs):
        at = int(rng.integers(0, max(1, length - n)))
        s[at:at + n] = b"N" * n
    return bytes(s)


@pytest.mark.parametrize("min_mask", [50, 10])
def test_region_masking_single_and_meta(ctx, models, min_mask):
    # ref: lib.pyx:699-713 (Sequence._mask), 1959-1966 / 2053-2061 (extraction); tests/test_gene_finder.py mask cases
    seqs = [_with_unknown_runs(30000, 0.45, 70, [9, 10, 11, 49, 50, 51, 120, 400, 1500]),
            _with_unknown_runs(12000, 0.6, 71, [50] * 12),
            b"N" * 70 + synthetic_contig(4000, 0.5, 72) + b"N" * 55,          # masks touching both ends
            synthetic_contig(3000, 0.5, 73),                                  # no unknown base at all
            synthetic_contig(5000, 0.5, 74) + b"NNN",                       # a short trailing run is a mask too
            b"N" * 400, b""]
    for meta in (True, False):
        ctx.set_models([m.buf for m in models] if meta else [models[1].buf])
        for closed in (False, True):
            res = ctx.find_genes_batch(seqs, meta=meta, closed=closed, want_nodes=True, mask=True, min_mask=min_mask)
            n = 0
            for i, s in enumerate(seqs):
                n += compare_contig(res, i, s, orc.Oracle(s, mask=True, mask_size=min_mask), models if meta else [models[1]], meta=meta, closed=closed)
            assert n > 0
    # masking changes the calls on these inputs (otherwise the test would prove nothing)
    ctx.set_models([models[1].buf])
    a = ctx.find_genes_batch(seqs[:1], meta=False, mask=True, min_mask=min_mask).genes
    b = ctx.find_genes_batch(seqs[:1], meta=False).genes
    assert len(a) != len(b) or not np.array_equal(a["begin"], b["begin"])


def test_no_model_in_gc_window(ctx, models):
    ctx.set_models([models[0].buf])            # gc 0.30 only
    seq = synthetic_contig(5000, 0.70, 3)
    res = ctx.find_genes_batch([seq], meta=True)
    assert res.contigs[0]["model"] == -1 and res.contigs[0]["n_genes"] == 0


def test_invalid_arguments_raise(ctx, models):
    ctx.set_models([models[0].buf])
    with pytest.raises(ValueError):
        ctx.find_genes_batch([b"ACGT"], min_gene=0)
    with pytest.raises(ValueError):
        ctx.find_genes_batch([b"ACGT"], max_overlap=100, min_gene=90)


@pytest.mark.parametrize("tail", ["device", "host"])
def test_both_tails_give_the_same_result(ctx, models, tail, monkeypatch):
    # the traceback tail runs on the device for many small contigs and on host threads for few / long ones; force each
    monkeypatch.setenv("PGA_TAIL", tail)
    ctx.set_models([m.buf for m in models])
    seqs = [synthetic_contig(9000 + 611 * c, 0.32 + 0.36 * (c % 13) / 12, 7000 + c) for c in range(70)] + [b"", b"ATGAAATAA"]
    for meta in (True, False):
        if not meta:
            ctx.set_models([models[2].buf])
        for want_nodes in (True, False):
            res = ctx.find_genes_batch(seqs, meta=meta, want_nodes=want_nodes)
            n = sum(compare_contig(res, i, s, orc.Oracle(s), models if meta else [models[2]], meta=meta) for i, s in enumerate(seqs))
            assert n > 0


@pytest.mark.parametrize("kernel", ["wave", "wavedyn", "wavemiss", "tree3"])
def test_connection_scoring_kernels_inside_the_finder(ctx, models, kernel, monkeypatch):
    # batches this small take the chain kernel by default; forcing either kernel must not change one node field:
    # several models per contig, two translation-table groups, empty and sub-window contigs, every node against the oracle
    # (wave: step schedule + assembly steps; wavedyn: PGA_DPW_SCHED=0; wavemiss: the schedule reports it did not fit and the launch
    #  is repeated by k_dpw_dyn)
    monkeypatch.setenv("PGA_DP_KERNEL", "wave" if kernel.startswith("wave") else kernel)
    if kernel == "wavedyn":
        monkeypatch.setenv("PGA_DPW_SCHED", "0")
    if kernel == "wavemiss":
        monkeypatch.setenv("PGA_DPW_SCHED_MISS", "1")
    ctx.set_models([m.buf for m in models])
    seqs = [synthetic_contig(3000 + 2111 * c, 0.30 + 0.40 * (c % 41) / 40, 20_000 + c) for c in range(40)]
    seqs += [b"", b"ATGAAATAA", synthetic_contig(70_000, 0.52, 99), read_fasta("SRR492066.fna.gz")[0][1].encode()]
    for closed in (False, True):
        res = ctx.find_genes_batch(seqs, meta=True, closed=closed, want_nodes=True)
        n = sum(compare_contig(res, i, s, orc.Oracle(s), models, meta=True, closed=closed) for i, s in enumerate(seqs))
        assert n > 100
    ctx.set_models([models[2].buf])
    res = ctx.find_genes_batch(seqs, meta=False, want_nodes=True)          # single mode keeps the DP pass's node fields
    assert sum(compare_contig(res, i, s, orc.Oracle(s), [models[2]], meta=False) for i, s in enumerate(seqs)) > 100
    if kernel.startswith("wave"):
        assert (ctx.dp_stats()["sched_missed"] > 0) == (kernel == "wavemiss")


def test_schedule_miss_inside_the_finder_unforced(ctx, models, monkeypatch):
    # a node-dense contig (more than 64 nodes within 3 * OPER_DIST bases) among ordinary ones: its schedule reports a miss and the
    # launch is repeated by k_dpw_dyn; every node of every contig against the oracle
    monkeypatch.setenv("PGA_DP_KERNEL", "wave")
    ctx.set_models([m.buf for m in models])
    dense = (b"ATGCATCAC" * 150) + b"TAATTA" + (b"GTGCACCAT" * 100) + b"TAGCTA" + synthetic_contig(2500, 0.5, 78)
    seqs = [synthetic_contig(9000 + 1300 * c, 0.35 + 0.03 * c, 30_000 + c) for c in range(8)] + [dense]
    res = ctx.find_genes_batch(seqs, meta=True, want_nodes=True)
    assert sum(compare_contig(res, i, s, orc.Oracle(s), models, meta=True) for i, s in enumerate(seqs)) > 10
    assert ctx.dp_stats()["sched_missed"] > 0


def test_extraction_staging_overflow_takes_the_full_staging(models, monkeypatch):
    """The extraction stages the nodes of a tile in one slot per two positions and extracts again with two slots per position when
    a tile does not fit (GroupArrays::st_half).  Same results from the default, from a staging so small that ordinary sequence
    overflows it (PGA_STAGE_SHIFT=5: the second pass runs), and from full staging from the start; the node-densest periodic
    sequences (one node per two positions) go through the default."""
    from pyrodigal_amd import _cabi
    seqs = [synthetic_contig(20000 + 997 * c, 0.30 + 0.40 * (c % 41) / 40, 20000 + c) for c in range(24)]
    seqs += [(b"TGCA" * 4000)[:15000], (b"CATG" * 3000)[:9001], (b"CATGTG" * 2000)[:9000], b"TGCA" * 20]
    results = {}
    for mode, env in (("default", {}), ("tiny", {"PGA_STAGE_SHIFT": "5"}), ("full", {"PGA_STAGE_FULL": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = _cabi.Context(0)
        c.set_models([m.buf for m in models])
        results[mode] = [c.find_genes_batch(seqs, meta=True, want_nodes=True) for _ in range(2)]      # the second call: the context's sticky choice
        c.close()
        for k in env:
            monkeypatch.delenv(k)
    ref = results["full"][0]
    for i, s in enumerate(seqs):
        compare_contig(ref, i, s, orc.Oracle(s), models, meta=True)
    for mode in ("default", "tiny"):
        for r in results[mode]:
            assert r.genes.tobytes() == ref.genes.tobytes(), mode
            assert np.array_equal(r.contigs["model"], ref.contigs["model"]) and np.array_equal(r.contigs["n_nodes"], ref.contigs["n_nodes"]), mode
            for i in range(len(seqs)):
                for k in ("ndx", "stop_val", "type", "strand"):
                    assert np.array_equal(r.nodes[i][k], ref.nodes[i][k]), (mode, i, k)


def test_short_last_tiles_do_not_overflow_the_staging(models):
    """A contig's last extraction tile can be a few bases long (length = k * 3072 + 1 .. 8) and still hold the six edge nodes of an
    open end: every tile carries constant slack on top of its one-slot-per-two-positions share, so ordinary lengths stay on one pass."""
    from pyrodigal_amd import _cabi
    tile = 3072
    seqs = [synthetic_contig(k * tile + r, 0.35 + 0.3 * (r % 5) / 4, 4100 + 10 * k + r) for k in (1, 2, 5) for r in range(1, 9)]
    seqs += [synthetic_contig(L, 0.5, 4200 + L) for L in (7, 61, 130, 3071, 3073, 20000)]
    c = _cabi.Context(0)
    c.set_models([m.buf for m in models])
    res = c.find_genes_batch(seqs, meta=True, want_nodes=True)
    assert c.extract_stats()["passes"] == 1
    n = sum(compare_contig(res, i, s, orc.Oracle(s), models, meta=True) for i, s in enumerate(seqs))
    assert n > 20
    c.close()


def test_planted_orf_series_against_the_oracle(ctx, models):
    """SURVEY 8(d)'s metagenome-like series (bench.py --series planted / secondary.config4_planted): fifty 20 kbp contigs of planted
    ORFs -- real node density, long real ORFs, operon steps, overlapping 3' ends, start tweaks -- every node field and every gene
    against the oracle, through the many-chain kernels the bench runs them on."""
    from pyrodigal_amd import benchdata
    lengths, gcs, seeds = benchdata.config4_spec(50)
    seqs = benchdata.generate(lengths, gcs, seeds, planted=True)
    ctx.set_models([m.buf for m in models])
    os.environ["PGA_DP_KERNEL"] = "wave"
    try:
        res = ctx.find_genes_batch(seqs, meta=True, want_nodes=True)
    finally:
        del os.environ["PGA_DP_KERNEL"]
    n = sum(compare_contig(res, i, s, orc.Oracle(s), models, meta=True) for i, s in enumerate(seqs))
    dens = float(np.sum(res.contigs["n_nodes"])) / sum(len(s) for s in seqs)
    assert n > 300 and dens > 0.035, (n, dens)


def test_upload_beside_a_call_of_the_same_context(ctx, models):
    """`pga_batch_create` has a stream, a pinned staging area and a worker pool of its own (round 6): a second host thread may pack and
    upload the next batch while `pga_find_genes` works on the current one, on ONE context.  Every result equals the serial one."""
    import threading
    ctx.set_models([m.buf for m in models])
    rng = np.random.default_rng(61)
    groups = [[synthetic_contig(int(rng.integers(3000, 40000)), 0.3 + 0.4 * rng.random(), 6100 + 40 * g + i) for i in range(24)] for g in range(6)]
    # (one group large enough for the slices of the threaded packing: more than 8 MB)
    groups.append([synthetic_contig(1_200_000, 0.3 + 0.05 * i, 6400 + i) for i in range(8)])
    serial = [ctx.find_genes_batch(g, meta=True) for g in groups]
    got = [None] * len(groups)
    nxt = {"b": ctx.upload(groups[0]), "err": None}
    for k in range(len(groups)):
        b = nxt["b"]

        def ahead(k=k):
            try:
                nxt["b"] = ctx.upload(groups[k + 1]) if k + 1 < len(groups) else None
            except BaseException as e:          # noqa: BLE001 -- handed to the main thread
                nxt["err"] = e
        t = threading.Thread(target=ahead)
        t.start()
        got[k] = ctx.find_genes(b, meta=True)
        t.join()
        b.close()
        assert nxt["err"] is None, nxt["err"]
    for k, (a, r) in enumerate(zip(serial, got)):
        assert np.array_equal(a.contigs["model"], r.contigs["model"]), k
        assert a.genes.tobytes() == r.genes.tobytes(), k


def _start_dense():
    """Start codons every four bases in rotating frames around stops on either strand: stop nodes with more than four overlapping-start
    candidates inside max_overlap (k_ovl_stops prices the first four at once and goes on one by one)."""
    unit = b"ATGC" * 60 + b"TAA" + b"ATGC" * 61 + b"TAG" + b"ATGC" * 62 + b"TGA"
    fwd = unit * 3
    rev = fwd[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
    return fwd + synthetic_contig(1500, 0.5, 4242) + rev


def _max_overlap_candidates(on, maxov=60):
    best = 0
    ndx, sv, ty, st = on["ndx"], on["stop_val"], on["type"], on["strand"]
    for i in np.nonzero(ty == 3)[0]:
        my, fwd = ndx[i], st[i] == 1
        js = range(i + 3, -1, -1) if fwd else range(max(i - 3, 0), len(ndx))
        cnt = 0
        for j in js:
            if j < 0 or j >= len(ndx):
                continue
            if fwd:
                if ndx[j] > my + 2: continue
                if ndx[j] + maxov < my: break
                ok = st[j] == 1 and ty[j] != 3 and sv[j] > my
            else:
                if ndx[j] < my - 2: continue
                if ndx[j] - maxov > my: break
                ok = st[j] == -1 and ty[j] != 3 and sv[j] < my
            cnt += bool(ok)
        best = max(best, cnt)
    return best


@pytest.mark.parametrize("form", ["plan", "search"])
def test_more_than_four_overlapping_start_candidates(ctx, models, form, monkeypatch):
    # k_ovl_stops: the first four candidates of a stop node at once, the others one by one; the workgroup's chain from the call plan
    # or (PGA_OVL_SEARCH=1: what launches outside the finder's plan do) by a search.  With the wave-batch scorer the same pass builds the
    # extras records from the starts it kept: every node field, star_ptr and the connection scores against the oracle.
    if form == "search":
        monkeypatch.setenv("PGA_OVL_SEARCH", "1")
    dense = _start_dense()
    o = orc.Oracle(dense)
    o.find_genes_single(models[2], orc.Params())
    assert _max_overlap_candidates(o.nodes()) > 6
    seqs = [dense] + [synthetic_contig(6000 + 900 * c, 0.35 + 0.04 * c, 51_000 + c) for c in range(6)]
    for kernel in ("wave", "tree3"):
        monkeypatch.setenv("PGA_DP_KERNEL", kernel)
        ctx.set_models([m.buf for m in models])
        res = ctx.find_genes_batch(seqs, meta=True, want_nodes=True)
        assert sum(compare_contig(res, i, s, orc.Oracle(s), models, meta=True) for i, s in enumerate(seqs)) > 10
        res = ctx.find_genes_batch(seqs, meta=True)                           # the path proper (no node arrays: lean stores, no star_ptr fill)
        assert sum(compare_contig(res, i, s, orc.Oracle(s), models, meta=True) for i, s in enumerate(seqs)) > 10
        ctx.set_models([models[2].buf])
        res = ctx.find_genes_batch(seqs, meta=False, want_nodes=True)
        assert sum(compare_contig(res, i, s, orc.Oracle(s), [models[2]], meta=False) for i, s in enumerate(seqs)) > 10
