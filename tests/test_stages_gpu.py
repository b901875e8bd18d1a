"""Stage-level C-ABI (pga_nodes_stage) against the oracle's restatement of Nodes.extract / sort / score /
_record_overlapping_starts (ref: lib.pyx:2501-2595, 2279-2329), bit for bit."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import golden_path, read_fasta, synthetic_contig

pytestmark = pytest.mark.gpu

TOPO = ["ndx", "stop_val", "type", "strand", "edge"]
SCORED_INT = ["mot_ndx", "mot_len", "mot_spacer", "mot_spacendx"]
SCORED_F64 = ["cscore", "sscore", "rscore", "uscore", "tscore", "mot_score"]


@pytest.fixture(scope="module")
def ctx():
    from pyrodigal_amd import _cabi
    c = _cabi.Context(0)
    yield c
    c.close()


def oracle_stage(seq, stage, tinf=None, tt=11, closed=False, is_meta=False, min_gene=90, min_edge_gene=60):
    o = orc.Oracle(seq)
    o.extract(tinf.trans_table if tinf is not None else tt, orc.Params(closed=closed, min_gene=min_gene, min_edge_gene=min_edge_gene))
    o.sort()
    if stage >= 2:
        o.reset_scores()
        o.score_nodes(tinf, closed, is_meta)
    if stage >= 3:
        o.overlapping_starts(tinf, 1, 60)
    return o.nodes()


def check(nd, on, stage):
    assert nd["n"] == len(on)
    for k in TOPO:
        assert np.array_equal(nd[k].astype(np.int64), on[k].astype(np.int64)), k
    assert np.all(nd["traceb"] == -1) and np.all(nd["tracef"] == -1) and np.all(nd["ov_mark"] == -1)
    if stage >= 2:
        for k in SCORED_INT:
            assert np.array_equal(nd[k].astype(np.int64), on[k].astype(np.int64)), k
        for k in SCORED_F64:
            assert np.array_equal(nd[k].view(np.uint64), on[k].view(np.uint64)), k
        assert np.array_equal(nd["gc_cont"].view(np.uint32), on["gc_cont"].view(np.uint32))
        assert np.array_equal(nd["rbs"], on["rbs"])
    assert np.array_equal(nd["star_ptr"], on["star_ptr"] if stage >= 3 else np.zeros_like(on["star_ptr"]))


@pytest.mark.parametrize("tt", [11, 4])
@pytest.mark.parametrize("closed", [False, True])
def test_extract_stage_fixtures(ctx, tt, closed):
    from pyrodigal_amd import _cabi
    seqs = [read_fasta("SRR492066.fna.gz")[0][1], read_fasta("MIIJ01000039.fna.gz")[0][1], synthetic_contig(30000, 0.62, 5), "ATGAAATAA", ""]
    out = ctx.nodes_stage(seqs, _cabi.STAGE_EXTRACT, translation_table=tt, closed=closed)
    for seq, nd in zip(seqs, out):
        check(nd, oracle_stage(seq, 1, tt=tt, closed=closed), 1)
    if tt == 11 and not closed:
        assert out[0]["n"] == 2293          # ref: tests/test_nodes.py:28-40


def test_extract_stage_with_masks(ctx):
    from pyrodigal_amd import _cabi
    rng = np.random.default_rng(3)
    s = bytearray(synthetic_contig(40000, 0.5, 33))
    for n in (20, 50, 80, 200, 1000):
        at = int(rng.integers(0, 39000)); s[at:at + n] = b"N" * n
    seq = bytes(s)
    for min_mask in (50, 0, 30):
        out = ctx.nodes_stage([seq], _cabi.STAGE_EXTRACT, mask=True, min_mask=min_mask)
        o = orc.Oracle(seq, mask=True, mask_size=min_mask)
        o.extract(11, orc.Params()); o.sort()
        check(out[0], o.nodes(), 1)
    assert ctx.nodes_stage([seq], _cabi.STAGE_EXTRACT)[0]["n"] != out[0]["n"]


def test_sequence_stage_gc_unknown_and_masks(ctx):
    from pyrodigal_amd import _cabi
    # ref: tests/test_sequence.py:36-52 (mask intervals), lib.pyx:664-697 (gc over all bases, unknown count)
    s = "ATGCNNNNNNNNNNATGCNNNNNNNNTGC"
    r = ctx.nodes_stage([s, s.lower(), "", "ACGTRYKM"], _cabi.STAGE_SEQUENCE, mask=True, min_mask=0)
    assert r.masks[0].tolist() == [[4, 14], [18, 26]] and r.masks[1].tolist() == [[4, 14], [18, 26]]
    assert len(r.masks[2]) == 0 and r.masks[3].tolist() == [[4, 8]]
    assert r.contigs["n_unknown"].tolist() == [18, 18, 0, 4]
    assert r.contigs["gc"][0] == 6 / 29 and r.contigs["gc"][3] == 2 / 8
    r = ctx.nodes_stage([s], _cabi.STAGE_SEQUENCE, mask=True, min_mask=10)
    assert r.masks[0].tolist() == [[4, 14]]
    assert ctx.nodes_stage([s], _cabi.STAGE_SEQUENCE).masks is None          # ref: test_no_region_masking
    # a run that reaches the end of the sequence is masked whatever its length (ref: lib.pyx:711-712)
    r = ctx.nodes_stage([b"ACGTNNNACGTNN", b"ACGTNNN" + b"A" * 60 + b"N"], _cabi.STAGE_SEQUENCE, mask=True, min_mask=50)
    assert r.masks[0].tolist() == [[11, 13]] and r.masks[1].tolist() == [[67, 68]]
    big = synthetic_contig(5000, 0.5, 1) + b"N" * 3000 + synthetic_contig(100, 0.5, 2) + b"n" * 50
    r = ctx.nodes_stage([big], _cabi.STAGE_SEQUENCE, mask=True)
    assert r.masks[0].tolist() == [[5000, 8000], [8100, 8150]]


@pytest.mark.parametrize("closed", [False, True])
def test_extract_stage_contig_lengths_around_the_tile_size(ctx, closed):
    # a workgroup owns 3072 forward positions of both strands; the last tile of a contig is ragged, one or two positions long
    # when the length is 1 or 2 past a multiple, and the reverse strand's first codons lie in it
    from pyrodigal_amd import _cabi
    lens = [3, 4, 95, 3070, 3071, 3072, 3073, 3074, 3075, 6143, 6144, 6145, 6146, 9216, 9217, 9218, 12290]
    seqs = [synthetic_contig(L, 0.35 + 0.03 * (k % 10), 400 + k) for k, L in enumerate(lens)]
    seqs += [s.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1] for s in seqs[3:9]]       # the same tiles seen from the other strand
    for tt in (11, 4):
        out = ctx.nodes_stage(seqs, _cabi.STAGE_EXTRACT, translation_table=tt, closed=closed)
        for seq, nd in zip(seqs, out):
            check(nd, oracle_stage(seq, 1, tt=tt, closed=closed), 1)


def test_masks_and_nodes_with_unknown_bases_at_tile_boundaries(ctx):
    from pyrodigal_amd import _cabi
    base = synthetic_contig(3 * 3072 + 2, 0.5, 77)
    seqs = []
    for L, runs in [(3073, [(3072, 1)]), (3074, [(3072, 2)]), (6145, [(6100, 45)]), (9218, [(3060, 30), (6140, 10), (9217, 1)]), (6144, [(3071, 2), (6143, 1)])]:
        s = bytearray(base[:L])
        for at, n in runs: s[at:at + n] = b"N" * n
        seqs.append(bytes(s))
    r = ctx.nodes_stage(seqs, _cabi.STAGE_SEQUENCE, mask=True, min_mask=50)
    # a run that reaches the end of the sequence is masked whatever its length (ref: lib.pyx:711-712), also when it is the only
    # position of the contig's last tile
    assert r.masks[0].tolist() == [[3072, 3073]] and r.masks[1].tolist() == [[3072, 3074]] and r.masks[2].tolist() == [[6100, 6145]]
    assert r.masks[3].tolist() == [[9217, 9218]] and r.masks[4].tolist() == [[6143, 6144]]
    for min_mask in (50, 0):
        out = ctx.nodes_stage(seqs, _cabi.STAGE_EXTRACT, mask=True, min_mask=min_mask)
        for seq, nd in zip(seqs, out):
            o = orc.Oracle(seq, mask=True, mask_size=min_mask)
            o.extract(11, orc.Params()); o.sort()
            check(nd, o.nodes(), 1)


def test_extract_stage_gene_length_options(ctx):
    from pyrodigal_amd import _cabi
    seq = synthetic_contig(50000, 0.45, 9)
    out = ctx.nodes_stage([seq], _cabi.STAGE_EXTRACT, min_gene=120, min_edge_gene=90)
    check(out[0], oracle_stage(seq, 1, min_gene=120, min_edge_gene=90), 1)


@pytest.mark.parametrize("is_meta", [False, True])
@pytest.mark.parametrize("stage", [2, 3])
def test_score_stage_sd_and_nonsd_models(ctx, stage, is_meta):
    seq_a = read_fasta("SRR492066.fna.gz")[0][1]
    seq_b = read_fasta("KK037166.fna.gz")[0][1]
    sd = orc.Training.load(golden_path("SRR492066.training.bin.gz"))
    nonsd = orc.Oracle(seq_b).train()
    assert sd.uses_sd == 1 and nonsd.uses_sd == 0
    for tinf in (sd, nonsd):
        ctx.set_models([tinf.tobytes()])
        seqs = [seq_a, seq_b[:60000], synthetic_contig(20000, 0.38, 21)]
        out = ctx.nodes_stage(seqs, stage, is_meta=is_meta)
        for seq, nd in zip(seqs, out):
            check(nd, oracle_stage(seq, stage, tinf=tinf, is_meta=is_meta), stage)


def test_rbs_search_next_to_the_sequence_ends(ctx):
    """Starts within 20 bases of an end: the forward strand skips upstream windows that begin before the sequence, the reverse
    strand searches them with the missing bases matching nothing (ref: lib.pyx:2256-2275)."""
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    tinf = orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz"))
    ctx.set_models([tinf.tobytes()])
    body = synthetic_contig(600, 0.5, 77).replace(b"TAA", b"TCA").replace(b"TAG", b"TCG").replace(b"TGA", b"TCA")
    orf = b"ATG" + body[:450] + b"TAA"
    seqs = []
    for k in range(0, 24, 3):
        up = (b"GGAGGAAAACAT" + b"AGGAGGTTTAAC")[:k]                        # what lies upstream of the start, partly cut off
        seqs.append(up[::-1][:k][::-1] + orf + synthetic_contig(300, 0.5, k))                            # forward start k bases from the left end
        seqs.append(synthetic_contig(300, 0.5, 100 + k) + (up[::-1][:k][::-1] + orf).translate(comp)[::-1])   # reverse start k bases from the right end
    for closed in (True, False):
        out = ctx.nodes_stage(seqs, 2, closed=closed)
        hits = 0
        for seq, nd in zip(seqs, out):
            on = oracle_stage(seq, 2, tinf=tinf, closed=closed)
            check(nd, on, 2)
            hits += int((on["rbs"] > 0).any())
        assert hits > 0


def test_score_stage_needs_a_model(ctx):
    from pyrodigal_amd import _cabi
    ctx.set_models([])
    with pytest.raises(ValueError):
        ctx.nodes_stage(["ATGC" * 100], _cabi.STAGE_SCORE)
    with pytest.raises(ValueError):
        ctx.nodes_stage(["ATGC" * 100], 7)


@pytest.mark.parametrize("closed", [False, True])
def test_extract_stage_tiles_without_a_stop(ctx, closed):
    # k_tile_stops looks for a tile's first / last stop of a frame in its first / last 384 positions and goes on only where there is none:
    # 12 kb without a stop codon in any frame of either strand (GCC repeats with start codons inside) span four tiles that hold no stop
    # at all, between ordinary sequence; and a GC-rich stretch whose ORFs run for hundreds of codons
    from pyrodigal_amd import _cabi
    quiet = (b"GCC" * 150 + b"ATG" + b"GCC" * 150 + b"GTG") * 13
    seqs = [synthetic_contig(2000, 0.5, 11) + b"ATG" + quiet + b"TAA" + synthetic_contig(2500, 0.45, 12),
            synthetic_contig(9000, 0.78, 13), quiet[:3080], b"CAT" + quiet[:6200][::-1]]
    for tt in (11, 4):
        out = ctx.nodes_stage(seqs, _cabi.STAGE_EXTRACT, translation_table=tt, closed=closed)
        for seq, nd in zip(seqs, out):
            check(nd, oracle_stage(seq, 1, tt=tt, closed=closed), 1)
    assert out[0]["n"] > 50
