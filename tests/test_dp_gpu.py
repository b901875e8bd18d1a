"""GPU parity: connection-scoring DP kernel vs the CPU oracle, through the C-ABI scorer-level call
(`pga_score_connections`, the whole-array drop-in for ConnectionScorer.index + score_connections,
ref: lib.pyx:1126-1237).  Scores must be bit-identical, traceb / ov_mark / max index equal."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import golden_path, read_fasta, synthetic_contig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pyrodigal_amd import _cabi
    c = _cabi.Context(0)
    yield c
    c.close()


def oracle_dp(seq, tinf, closed=False, is_meta=False):
    """Scored nodes + raw DP state (before the traceback fix-ups) from the oracle."""
    o = orc.Oracle(seq)
    o.extract(tinf.trans_table, orc.Params(closed=closed)); o.sort(); o.reset_scores()
    o.score_nodes(tinf, closed, is_meta)
    o.overlapping_starts(tinf, 1, 60)
    o.dprog_raw(tinf, True)
    return o.nodes(), o.find_max_index()


def check(ctx, seq, tinf, **kw):
    ref, ref_max = oracle_dp(seq, tinf, **kw)
    n = len(ref)
    score, traceb, ov, mi, ms = ctx.score_connections(
        ref["ndx"], ref["stop_val"], ref["type"], ref["strand"], ref["cscore"], ref["sscore"],
        ref["rscore"], ref["uscore"], ref["star_ptr"], tinf.st_wt, True)
    assert np.array_equal(traceb, ref["traceb"])
    assert np.array_equal(score.view(np.uint64), ref["score"].view(np.uint64)), "score not bit-identical"
    reached = ref["traceb"] != -1      # ov_mark is only defined once a connection was made
    assert np.array_equal(ov[reached], ref["ov_mark"][reached])
    assert mi == ref_max
    return n, ms


def test_dp_srr492066(ctx):
    seq = read_fasta("SRR492066.fna.gz")[0][1]
    tinf = orc.Training.load(golden_path("SRR492066.training.bin.gz"))
    n, _ = check(ctx, seq, tinf)
    assert n == 2293


def test_dp_nonsd_model_kk037166(ctx):
    seq = read_fasta("KK037166.fna.gz")[0][1]
    tinf = orc.Oracle(seq).train()
    assert tinf.uses_sd == 0
    check(ctx, seq, tinf)


def test_dp_miij(ctx):
    seq = read_fasta("MIIJ01000039.fna.gz")[0][1]
    tinf = orc.Oracle(seq).train()
    check(ctx, seq, tinf)


def test_dp_giant_orf_windows_full_genome(ctx):
    # 153k nodes; the window walk-back for giant ORFs fires ~100 times here
    seq = read_fasta("GCF_001457455.1_NCTC11397_genomic.fna.gz")[0][1]
    tinf = orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz"))
    n, ms = check(ctx, seq, tinf, closed=True)
    assert n == 153296
    print(f"full genome DP kernel: {ms:.2f} ms for {n} nodes")


@pytest.mark.parametrize("gc,seed", [(0.3, 11), (0.5, 12), (0.7, 13)])
def test_dp_synthetic_meta_scoring(ctx, gc, seed):
    seq = synthetic_contig(60000, gc, seed)
    tinf = orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic_100kb.tinf_closed.bin.gz"))
    check(ctx, seq, tinf, is_meta=True)


def test_dp_tiny_and_empty_inputs(ctx):
    tinf = orc.Training.load(golden_path("SRR492066.training.bin.gz"))
    s, t, v, mi, _ = ctx.score_connections([], [], [], [], [], [], [], [], np.zeros((0, 3)), tinf.st_wt)
    assert len(s) == 0 and mi == -1
    for L in (100, 130, 200, 400, 1000, 5000):
        seq = synthetic_contig(L, 0.45, 100 + L)
        ref, _ = oracle_dp(seq, tinf, is_meta=True)
        if len(ref):
            check(ctx, seq, tinf, is_meta=True)


@pytest.mark.parametrize("name,closed", [("SRR492066", False), ("KK037166", False), ("GCF_001457455.1_NCTC11397_genomic", True)])
def test_training_pass_of_the_scorer(ctx, name, closed):
    # final = 0 (ref: _connection.h `final == false` branches, call site lib.pyx:5265): frame-bias factors instead of scores
    seq = read_fasta(name + ".fna.gz")[0][1]
    st_wt = 4.35
    t = orc.Training()
    t.set_trans_table(11); t._f64(16)[0] = st_wt
    o = orc.Oracle(seq)
    o.extract(11, orc.Params(closed=closed)); o.sort()
    o.record_gc_bias(t)
    bias = t.bias.copy()
    o.overlapping_starts(t, 0, 60)
    before = o.nodes()
    o.dprog_raw(t, False)
    ref = o.nodes()
    score, traceb, ov, mi, ms = ctx.score_connections_training(before["ndx"], before["stop_val"], before["type"], before["strand"],
                                                                before["gc_score"], bias, before["star_ptr"], st_wt)
    assert np.array_equal(traceb, ref["traceb"])
    assert np.array_equal(score.view(np.uint64), ref["score"].view(np.uint64))
    reached = ref["traceb"] != -1
    assert np.array_equal(ov[reached], ref["ov_mark"][reached]) and mi == o.find_max_index() and reached.sum() > 100


def test_training_pass_is_rejected_loudly(ctx):
    with pytest.raises(ValueError):
        ctx.score_connections([0], [0], [0], [1], [0.0], [0.0], [0.0], [0.0], np.zeros((1, 3)), 4.35, final=False)


def _wave_env(monkeypatch, kernel):
    """wave: k_dp_wave (step schedule + assembly steps); wavedyn: k_dpw_dyn, the lane masks worked out per step (PGA_DPW_SCHED=0);
    wavemiss: a schedule that reports it did not fit, i.e. the fallback from the one to the other (PGA_DPW_SCHED_MISS=1)"""
    monkeypatch.setenv("PGA_DP_KERNEL", "wave")
    if kernel == "wavedyn":
        monkeypatch.setenv("PGA_DPW_SCHED", "0")
    if kernel == "wavemiss":
        monkeypatch.setenv("PGA_DPW_SCHED_MISS", "1")


@pytest.mark.parametrize("variant", ["wave", "wavedyn", "wavemiss", "tree1", "tree3", "scan1", "scan4", "scan16"])
def test_dp_kernel_variants_agree_with_oracle(ctx, variant, monkeypatch):
    # wave = the wave-batch kernel of launches with many chains (dp_wave.hip; see _wave_env); tree3 = the chain kernel of few long chains;
    # tree1 = its one-wave form; PGA_DP_KERNEL=scan selects the window-scanning kernels with 1, 4 or 16 wavefronts per chain,
    # kept as an independent cross-check
    if variant.startswith("scan"):
        monkeypatch.setenv("PGA_DP_KERNEL", "scan")
        monkeypatch.setenv("PGA_DP_WAVES", variant[4:])
    elif variant.startswith("wave"):
        _wave_env(monkeypatch, variant)
    else:
        monkeypatch.setenv("PGA_DP_KERNEL", variant)     # tree1: one wave per chain; tree3: three cooperating waves
    seq = read_fasta("MIIJ01000039.fna.gz")[0][1]
    tinf = orc.Training.load(golden_path("SRR492066.training.bin.gz"))
    check(ctx, seq, tinf, is_meta=True)
    seq = synthetic_contig(150000, 0.62, 5)
    check(ctx, seq, tinf, is_meta=True)
    for L in (100, 700, 4000):                        # chains shorter than / around one 64-node batch
        s2 = synthetic_contig(L, 0.5, 900 + L)
        if len(oracle_dp(s2, tinf, is_meta=True)[0]):
            check(ctx, s2, tinf, is_meta=True)


@pytest.mark.parametrize("name,model_file,closed", [
    ("SRR492066", "SRR492066.training.bin.gz", False),
    ("GCF_001457455.1_NCTC11397_genomic", "GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz", True),
    ("KK037166", "GCF_001457455.1_NCTC11397_genomic_100kb.tinf_closed.bin.gz", False),
])
@pytest.mark.parametrize("kernel", ["wave", "wavedyn"])
def test_wave_kernel_on_reference_fixtures(ctx, name, model_file, closed, kernel, monkeypatch):
    # the kernels of many-chain launches, forced onto single chains: short contigs, and the full genome, whose 153 296 nodes
    # slide the 1000-node window over 2400 blocks (suffix maxima, both block-range ends) and hit the giant-ORF windows
    _wave_env(monkeypatch, kernel)
    seq = read_fasta(name + ".fna.gz")[0][1]
    tinf = orc.Training.load(golden_path(model_file))
    for is_meta in (False, True):
        check(ctx, seq, tinf, closed=closed, is_meta=is_meta)


@pytest.mark.parametrize("kernel", ["wave", "wavedyn"])
def test_wave_kernel_on_synthetic_and_gene_dense_input(ctx, kernel, monkeypatch):
    _wave_env(monkeypatch, kernel)
    tinf = orc.Training.load(golden_path("SRR492066.training.bin.gz"))
    for k, (L, gc) in enumerate([(130, 0.45), (700, 0.5), (1500, 0.6), (20_000, 0.3), (20_000, 0.7), (64_000, 0.55), (200_000, 0.66)]):
        seq = synthetic_contig(L, gc, 7700 + k)
        if len(oracle_dp(seq, tinf, is_meta=True)[0]):
            check(ctx, seq, tinf, is_meta=True, closed=bool(k & 1))
    rng = np.random.default_rng(9)
    parts = []
    for k in range(600):                              # planted ORFs on both strands: many overlapping 3' ends and operons
        body = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 3 * int(rng.integers(30, 200))))
        orf = b"ATG" + body.replace(b"TAA", b"TCA").replace(b"TAG", b"TCG").replace(b"TGA", b"TCA") + b"TAA"
        if rng.random() < 0.5:
            orf = orf.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]
        parts.append(orf + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(0, 40)))))
    for closed in (False, True):
        n, _ = check(ctx, b"".join(parts), tinf, closed=closed)
        assert n > 3000


@pytest.mark.parametrize("kernel", ["wave", "wavedyn", "wavemiss", "tree1"])
def test_reverse_start_two_bases_before_a_reverse_stop(ctx, kernel, monkeypatch):
    # Contig 178 of tools/stress_variants.py seed 830022 (round 6): reverse start 1318 at 19849, reverse stop 1320 at 19851 -- too close to
    # connect (ref: _connection.h:337-342) -- and no other reverse stop within 3 * OPER_DIST bases behind it, so the node had no second
    # schedule word and the walk's shortcut for such a reverse start ("every gene begin behind it") connected the two: 2 644 of the
    # chain's 3 989 nodes differed from the oracle.  k_dpw_sched now puts a gene begin the node does NOT reach into its second word.
    from pyrodigal_amd import benchdata
    if kernel.startswith("wave"):
        _wave_env(monkeypatch, kernel)
    else:
        monkeypatch.setenv("PGA_DP_KERNEL", kernel)
    seq = read_fasta("sweep_830022_178.fna.gz")[0][1]
    tinf = orc.Training(benchdata.load_model_set()[7][1])
    n, _ = check(ctx, seq, tinf, closed=True)
    assert n == 3989


def test_topology_from_lds_near_its_node_limit(ctx, monkeypatch):
    # k_dpw_topo_lds stages a contig's node arrays in LDS (12 bytes per node + 2.3 KB): between 5 300 and 6 144 nodes that is more than
    # the 64 KB a kernel may use without asking (hipFuncAttributeMaxDynamicSharedMemorySize); same results as the global-memory kernel
    _wave_env(monkeypatch, "wave")
    tinf = orc.Training.load(golden_path("SRR492066.training.bin.gz"))
    seen = []
    for L, gc, seed in ((150_000, 0.50, 31), (143_000, 0.48, 32), (120_000, 0.58, 33)):
        seq = synthetic_contig(L, gc, seed)
        n, _ = check(ctx, seq, tinf, is_meta=True)
        seen.append(n)
        monkeypatch.setenv("PGA_DPW_TOPO_LDS", "0")
        check(ctx, seq, tinf, is_meta=True)
        monkeypatch.delenv("PGA_DPW_TOPO_LDS")
    assert any(5300 <= n <= 6144 for n in seen), seen


def test_schedule_miss_on_node_dense_sequence(ctx, monkeypatch):
    # more than 64 nodes within 3 * OPER_DIST bases: the near sources of a batch reach past the batch before it, the step schedule
    # reports the batch and the launch falls back to k_dpw_dyn -- unforced (the other tests force it with PGA_DPW_SCHED_MISS)
    _wave_env(monkeypatch, "wave")
    tinf = orc.Training.load(golden_path("SRR492066.training.bin.gz"))
    seq = (b"ATGCATCAC" * 150) + b"TAATTA" + (b"GTGCACCAT" * 100) + b"TAGCTA" + synthetic_contig(3000, 0.5, 77)
    for closed in (False, True):
        n, _ = check(ctx, seq, tinf, closed=closed)
        assert n > 500
