"""GPU parity of the segmented connection scoring (dp.hip "segmented chains"): a long chain is cut into segments that
are walked speculatively side by side, re-scored exactly and verified node by node.  Whatever the segment length and
the warm-up -- including settings so small that the speculation is often wrong and the verification has to repair
it or hand the chain to the serial walk -- score / traceb / ov_mark / max index must be those of the oracle's single
serial loop (ref: lib.pyx:1205-1237, _connection.h:94-408), bit for bit."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import golden_path, read_fasta, synthetic_contig

pytestmark = pytest.mark.gpu

SEG_ENV = ("PGA_DP_SEG", "PGA_DP_SEG_MIN", "PGA_DP_SEG_LEN", "PGA_DP_SEG_WARM", "PGA_DP_SEG_SLOTS", "PGA_DP_SEG_READBACK",
           "PGA_DP_SEG_WAVE", "PGA_DP_SEG_WSLOTS", "PGA_DP_SEG_WAVE_MIN", "PGA_DPW_SCHED_MISS")


@pytest.fixture(scope="module")
def ctx():
    from pyrodigal_amd import _cabi
    c = _cabi.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def clean_env():
    saved = {k: os.environ.pop(k, None) for k in SEG_ENV}
    yield
    for k, v in saved.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


@pytest.fixture(scope="module")
def genome():
    """153k scored nodes of a real genome and the oracle's raw DP state."""
    seq = read_fasta("GCF_001457455.1_NCTC11397_genomic.fna.gz")[0][1]
    tinf = orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz"))
    o = orc.Oracle(seq)
    o.extract(tinf.trans_table, orc.Params(closed=True)); o.sort(); o.reset_scores()
    o.score_nodes(tinf, True, False)
    o.overlapping_starts(tinf, 1, 60)
    o.dprog_raw(tinf, True)
    return o.nodes(), o.find_max_index(), tinf.st_wt


def run(ctx, ref, st_wt):
    return ctx.score_connections(ref["ndx"], ref["stop_val"], ref["type"], ref["strand"], ref["cscore"], ref["sscore"],
                                 ref["rscore"], ref["uscore"], ref["star_ptr"], st_wt, True)


def same(out, ref, ref_max):
    score, traceb, ov, mi, _ = out
    assert np.array_equal(traceb, ref["traceb"])
    assert np.array_equal(score.view(np.uint64), ref["score"].view(np.uint64)), "score not bit-identical"
    reached = ref["traceb"] != -1
    assert np.array_equal(ov[reached], ref["ov_mark"][reached])
    assert mi == ref_max


def test_default_plan_cuts_the_genome_and_verifies_clean(ctx, genome):
    ref, ref_max, st_wt = genome
    same(run(ctx, ref, st_wt), ref, ref_max)
    st = ctx.dp_stats()
    assert st["chains"] == 1 and 16 <= st["segments"] <= 256
    assert st["serial"] == 0                       # nothing fell back to the serial walk
    assert st["rejected"][-1] == 0                 # the last round that ran found no mismatch


def test_serial_walk_when_switched_off(ctx, genome):
    ref, ref_max, st_wt = genome
    os.environ["PGA_DP_SEG"] = "0"
    same(run(ctx, ref, st_wt), ref, ref_max)
    assert ctx.dp_stats()["chains"] == 0


@pytest.mark.parametrize("seg_len,warm", [(256, 64), (256, 512), (1024, 128), (4096, 64), (8192, 1024), (64, 64)])
def test_any_segment_length_and_warmup_is_exact(ctx, genome, seg_len, warm):
    """Short warm-ups make the speculation wrong somewhere; verification must catch every such node."""
    ref, ref_max, st_wt = genome
    os.environ.update({"PGA_DP_SEG_MIN": "300", "PGA_DP_SEG_LEN": str(seg_len), "PGA_DP_SEG_WARM": str(warm)})
    same(run(ctx, ref, st_wt), ref, ref_max)
    st = ctx.dp_stats()
    assert st["chains"] == 1 and st["segments"] >= 153296 // max(seg_len, 64) - 1
    if warm <= 128:
        assert st["rejected"][0] > 0, "a 64-128 node warm-up cannot settle every segment of a real genome"


def test_repair_rounds_and_serial_fallback_are_exercised(ctx, genome):
    """With a warm-up far too short the first claim is wrong in many places: some chains get repaired by the extra
    rounds, and a chain that is still inconsistent after the last round must be walked serially."""
    ref, ref_max, st_wt = genome
    seen_fallback = seen_repair = False
    for seg_len in (64, 128, 256):
        os.environ.update({"PGA_DP_SEG_MIN": "300", "PGA_DP_SEG_LEN": str(seg_len), "PGA_DP_SEG_WARM": "64"})
        same(run(ctx, ref, st_wt), ref, ref_max)
        st = ctx.dp_stats()
        seen_fallback |= st["serial"] == 1
        seen_repair |= st["rejected"][0] > 0 and st["serial"] == 0
    assert seen_fallback or seen_repair


@pytest.mark.parametrize("gc,seed,n_bp", [(0.35, 21, 400_000), (0.5, 22, 700_000), (0.65, 23, 400_000)])
def test_synthetic_long_contigs_meta_scoring(ctx, gc, seed, n_bp):
    seq = synthetic_contig(n_bp, gc, seed)
    tinf = orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic_100kb.tinf_closed.bin.gz"))
    o = orc.Oracle(seq)
    o.extract(tinf.trans_table, orc.Params(closed=False)); o.sort(); o.reset_scores()
    o.score_nodes(tinf, False, True)
    o.overlapping_starts(tinf, 1, 60)
    o.dprog_raw(tinf, True)
    ref, ref_max = o.nodes(), o.find_max_index()
    for env in ({}, {"PGA_DP_SEG_MIN": "300", "PGA_DP_SEG_LEN": "512", "PGA_DP_SEG_WARM": "256"}):
        os.environ.update(env)
        same(run(ctx, ref, tinf.st_wt), ref, ref_max)
    assert ctx.dp_stats()["chains"] == 1


def test_find_genes_reports_segments_and_matches_serial(ctx):
    """Whole pipeline, one 1.2 Mbp contig under the metagenomic models: identical genes with and without segments."""
    from pyrodigal_amd import _cabi, benchdata
    models = benchdata.load_model_set()
    c2 = _cabi.Context(0)
    try:
        c2.set_models([m[1] for m in models])
        seq = synthetic_contig(1_200_000, 0.5, 77)
        a = c2.find_genes_batch([seq], meta=True)
        st = c2.dp_stats()
        assert st["chains"] >= 1 and st["segments"] > st["chains"] and st["serial"] == 0
        os.environ["PGA_DP_SEG"] = "0"
        b = c2.find_genes_batch([seq], meta=True)
        assert c2.dp_stats()["chains"] == 0
        assert a.genes.tobytes() == b.genes.tobytes() and len(a.genes) > 100
        assert np.array_equal(a.contigs["model"], b.contigs["model"])
        # the finder reads the first round's verdict back and stops issuing rounds (round 5); the gated launches of all three rounds
        # are the other way to the same result -- also when the first round rejects nodes (64-node warm-up) and the repair rounds run
        for env in ({}, {"PGA_DP_SEG_MIN": "300", "PGA_DP_SEG_LEN": "2048", "PGA_DP_SEG_WARM": "64"}):
            got = []
            for readback in ("1", "0"):
                os.environ.pop("PGA_DP_SEG", None)
                os.environ.update(env)
                os.environ["PGA_DP_SEG_READBACK"] = readback
                r = c2.find_genes_batch([seq], meta=True)
                got.append((r.genes.tobytes(), dict(c2.dp_stats())))
            assert got[0][0] == got[1][0] == a.genes.tobytes()
            assert got[0][1]["rejected"] == got[1][1]["rejected"] and got[0][1]["serial"] == got[1][1]["serial"]
            if env:
                assert sum(got[0][1]["rejected"]) > 0
    finally:
        c2.close()


@pytest.mark.parametrize("walker", ["chain kernel", "wave kernel"])
def test_mixed_launch_segmented_and_whole_chains(ctx, walker):
    """One launch holds segments of the long chains next to the short chains walked whole; every node field of every contig
    against the oracle (the segment settings are shrunk so that a batch of modest contigs has both kinds).  wave kernel: the
    segments a wavefront each by k_dp_wave<6, true> (round 6), the chains that are not cut by the chain kernel in a launch of its own."""
    os.environ["PGA_DP_SEG_WAVE"] = "1" if walker == "wave kernel" else "0"
    from pyrodigal_amd import _cabi, benchdata
    from tests.test_finder_gpu import compare_contig
    models = benchdata.load_model_set()
    bins = [orc.Training(b) for _, b in models]
    c2 = _cabi.Context(0)
    try:
        c2.set_models([b for _, b in models])
        seqs = [synthetic_contig(L, gc, 4000 + k) for k, (L, gc) in enumerate(
            [(3_000, 0.5), (120_000, 0.45), (9_000, 0.6), (60_000, 0.5), (800, 0.4), (200_000, 0.55), (25_000, 0.35), (45_000, 0.65)])]
        os.environ.update({"PGA_DP_SEG_MIN": "2500", "PGA_DP_SEG_LEN": "512", "PGA_DP_SEG_WARM": "768"})
        res = c2.find_genes_batch(seqs, meta=True, want_nodes=True)
        st = c2.dp_stats()
        assert 0 < st["chains"] < res.n_chains and st["segments"] > st["chains"]       # some chains cut, others not
        total = sum(compare_contig(res, i, s, orc.Oracle(s), bins, meta=True) for i, s in enumerate(seqs))
        assert total > 300
    finally:
        c2.close()


def test_segments_walked_by_the_wave_kernel(ctx):
    """Round 6: the speculative walk of the segments by the wave-batch kernel (a wavefront per segment, up to 2048 at once) instead of the
    chain kernel (a workgroup per compute unit): the same genes as the serial walk -- with the default plan, with a warm-up so short that
    claims are wrong and the repair rounds run, and with a step schedule that reports every batch as missed (a segment then claims nothing:
    every chain ends in the serial walk)."""
    from pyrodigal_amd import _cabi, benchdata
    models = benchdata.load_model_set()
    c2 = _cabi.Context(0)
    try:
        c2.set_models([m[1] for m in models])
        seq = synthetic_contig(1_200_000, 0.5, 77)
        os.environ["PGA_DP_SEG"] = "0"
        b = c2.find_genes_batch([seq], meta=True)
        assert c2.dp_stats()["chains"] == 0 and len(b.genes) > 100
        os.environ.pop("PGA_DP_SEG")
        for env, want in (({}, "clean"), ({"PGA_DP_SEG_MIN": "300", "PGA_DP_SEG_LEN": "2048", "PGA_DP_SEG_WARM": "64"}, "repair"),
                          ({"PGA_DPW_SCHED_MISS": "1"}, "serial")):
            for k in SEG_ENV:
                os.environ.pop(k, None)
            os.environ["PGA_DP_SEG_WAVE"] = "1"
            os.environ.update(env)
            a = c2.find_genes_batch([seq], meta=True)
            st = c2.dp_stats()
            assert a.genes.tobytes() == b.genes.tobytes() and np.array_equal(a.contigs["model"], b.contigs["model"])
            assert st["chains"] >= 1 and st["segments"] > st["chains"]
            if want == "clean":
                assert st["segments"] > 256 and st["serial"] == 0 and st["rejected"][-1] == 0       # more segments than the chain kernel could walk at once
            elif want == "repair":
                assert sum(st["rejected"]) > 0
            else:
                assert st["serial"] >= 1
        # single mode, a real genome with its own model, every node field against the oracle
        for k in SEG_ENV:
            os.environ.pop(k, None)
        os.environ["PGA_DP_SEG_WAVE"] = "1"
        from tests.test_finder_gpu import compare_contig
        g = read_fasta("GCF_001457455.1_NCTC11397_genomic.fna.gz")[0][1]
        tinf = orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz"))
        c2.set_models([tinf.tobytes()])
        res = c2.find_genes_batch([g], meta=False, closed=True, want_nodes=True)
        st = c2.dp_stats()
        assert st["chains"] == 1 and st["segments"] > 256 and st["serial"] == 0
        assert compare_contig(res, 0, g, orc.Oracle(g), [tinf], meta=False, closed=True) > 2000
    finally:
        c2.close()
