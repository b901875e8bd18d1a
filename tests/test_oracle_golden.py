"""Pins the CPU oracle against the reference's own golden fixtures.

* byte-exact TrainingInfo dumps (ref: tests/test_training_info.py:57-66, tests/test_gene_finder.py:329-345)
* single-mode gene goldens produced by the real Prodigal binary (ref: tests/test_gene_finder.py:101-130)
* node counts (ref: tests/test_nodes.py:28-40, 77-99)
"""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import golden_path, read_fasta, parse_prodigal_header, gene_sequence


def _first_diff(a, b):
    a = np.frombuffer(a, np.uint8); b = np.frombuffer(b, np.uint8)
    d = np.nonzero(a != b)[0]
    return None if len(d) == 0 else (int(d[0]), len(d))


def test_train_closed_100kb_byte_exact():
    seq = read_fasta("GCF_001457455.1_NCTC11397_genomic_100kb.fna.gz")[0][1]
    o = orc.Oracle(seq)
    t = o.train(orc.Params(closed=True))
    want = orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic_100kb.tinf_closed.bin.gz"))
    assert _first_diff(t.tobytes(), want.tobytes()) is None


def test_train_srr492066_byte_exact():
    seq = read_fasta("SRR492066.fna.gz")[0][1]
    t = orc.Oracle(seq).train()
    want = orc.Training.load(golden_path("SRR492066.training.bin.gz"))
    assert _first_diff(t.tobytes(), want.tobytes()) is None
    # ref: tests/test_gene_finder.py:329-345
    assert t.trans_table == 11 and t.uses_sd == 1
    assert t.gc == pytest.approx(0.3010045159434068, abs=0)
    assert t.st_wt == 4.35


def test_train_full_genome_closed_byte_exact():
    seq = read_fasta("GCF_001457455.1_NCTC11397_genomic.fna.gz")[0][1]
    t = orc.Oracle(seq).train(orc.Params(closed=True))
    want = orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz"))
    assert _first_diff(t.tobytes(), want.tobytes()) is None


@pytest.mark.parametrize("name", ["SRR492066", "KK037166", "MIIJ01000039"])
def test_single_mode_goldens(name):
    seq = read_fasta(name + ".fna.gz")[0][1]
    o = orc.Oracle(seq)
    tinf = o.train()
    o.find_genes_single(tinf)
    recs = orc.gene_records(o.genes(), o.nodes(), tinf)
    want_p = read_fasta(name + ".single.faa.gz")
    want_g = read_fasta(name + ".single.fna.gz")
    assert len(recs) == len(want_p) == len(want_g)
    for got, (hdr, _), (_, nuc) in zip(recs, want_p, want_g):
        assert got == parse_prodigal_header(hdr)
        assert gene_sequence(seq, got[0], got[1], got[2]) == nuc


def test_node_count_tt11():
    # ref: tests/test_nodes.py:28-40 -- 2293 nodes with translation table 11
    seq = read_fasta("SRR492066.fna.gz")[0][1]
    o = orc.Oracle(seq)
    assert o.extract(11) == 2293


def test_extract_edge_start_not_duplicated():
    # regression shape of ref tests/test_nodes.py:77-99: an ATG at position <= 2 in open mode
    # yields exactly one node at that position
    seq = "ATG" + "GCA" * 80 + "TAA" + "CCC" * 10
    o = orc.Oracle(seq)
    o.extract(11); o.sort()
    n = o.nodes()
    fwd0 = n[(n["ndx"] == 0) & (n["strand"] == 1)]
    assert len(fwd0) == 1
    keys = n["ndx"].astype(np.int64) * 2 + (n["strand"] < 0)
    assert len(np.unique(keys)) == len(keys)


def test_short_and_empty_sequences():
    # ref: tests/test_gene_finder.py:222-234 -- no genes, no crash
    tinf = orc.Training.load(golden_path("SRR492066.training.bin.gz"))
    for s in ["", "A", "AT", "ATG", "ATGTAA", "ATGAAATAA"]:
        o = orc.Oracle(s)
        assert o.find_genes_single(tinf) == 0
        assert o.find_genes_meta([tinf]) in (-1, 0)
        assert o.num_genes == 0


def test_region_masks_match_reference_vectors():
    """ref: tests/test_sequence.py:36-52 (mask=False / mask_size 0 and 10) -- pins Sequence._mask (lib.pyx:699-713)."""
    s = "ATGCNNNNNNNNNNATGCNNNNNNNNTGC"
    assert len(orc.Oracle(s, mask=False).masks()) == 0
    assert orc.Oracle(s, mask=True, mask_size=0).masks().tolist() == [[4, 14], [18, 26]]
    assert orc.Oracle(s, mask=True, mask_size=10).masks().tolist() == [[4, 14]]



def test_the_oracle_is_built_without_fused_multiply_adds(tmp_path):
    """The checker is compiled -O2 -mavx2; what stands between it and an FMA (one rounding where the reference makes two) is
    -ffp-contract=off.  The flag must stay in oracle/Makefile, and an -O0 build of the same source must produce the same bytes
    (every score of every node and gene of a fixture genome, single mode with training)."""
    import hashlib, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = [ln for ln in open(os.path.join(root, "oracle", "Makefile")).read().splitlines() if ln.startswith("CFLAGS")][0]
    assert "-ffp-contract=off" in flags
    so = str(tmp_path / "libprodigal_oracle_O0.so")
    subprocess.run(["gcc", "-O0", "-ffp-contract=off", "-fPIC", "-std=gnu11", "-shared", "-pthread", "-o", so,
                    os.path.join(root, "oracle", "prodigal_oracle.c"), "-lm"], check=True)
    script = (
        "import sys, hashlib; sys.path.insert(0, %r)\n"
        "from oracle import oracle as orc\n"
        "if len(sys.argv) > 1: orc._LIB_PATH = sys.argv[1]; orc.build = lambda force=False: orc._LIB_PATH\n"
        "from tests.util import golden_path, read_fasta\n"
        "seq = read_fasta('SRR492066.fna.gz')[0][1]\n"
        "tinf = orc.Training.load(golden_path('SRR492066.training.bin.gz'))\n"
        "o = orc.Oracle(seq); o.extract(tinf.trans_table, orc.Params()); o.sort(); o.reset_scores(); o.score_nodes(tinf, False, False)\n"
        "o.overlapping_starts(tinf, 1, 60); o.dprog_raw(tinf, True)\n"
        "print(len(o.nodes()), hashlib.sha256(o.nodes().tobytes()).hexdigest())\n" % root)
    a = subprocess.run([sys.executable, "-c", script], check=True, capture_output=True, text=True).stdout
    b = subprocess.run([sys.executable, "-c", script, so], check=True, capture_output=True, text=True).stdout
    assert a == b and int(a.split()[0]) > 1000
