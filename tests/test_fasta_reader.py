"""The library's FASTA reader against a restatement of the parser the reference feeds its finder with
(ref: src/pyrodigal/tests/fasta.py:59-86): same records on the reference's own fixtures and on edge cases.
Host-side code: runs without a GPU."""
import gzip
import os

import numpy as np
import pytest

from tests.util import golden_path


def reference_parse(text):
    """Records as the reference's `parse` yields them from an open text file (description without its line end)."""
    id_, seq, desc, out = None, [], "", []
    for line in text.splitlines(keepends=True):
        l = line.strip()
        if line.startswith(">"):
            if id_ is not None:
                out.append((id_, desc, "".join(seq)))
            fields = line[1:].split(maxsplit=1)
            id_ = fields[0] if fields else ""
            desc = fields[1].rstrip() if len(fields) > 1 else ""
            seq = []
        elif l:
            seq.append(l)
    if id_ is not None:
        out.append((id_, desc, "".join(seq)))
    elif seq:
        raise ValueError("not in FASTA format")
    return out


def read_all(path, **kw):
    from pyrodigal_amd import _cabi
    with _cabi.FastaReader(path) as r:
        return [(i, d, s.decode("ascii")) for b in r.batches(**kw) for i, d, s in b]


@pytest.mark.parametrize("name", ["SRR492066.fna.gz", "KK037166.fna.gz", "MIIJ01000039.fna.gz",
                                  "GCF_001457455.1_NCTC11397_genomic.fna.gz", "SRR492066.single.faa.gz"])
def test_fixtures_parse_like_the_reference(name):
    want = reference_parse(gzip.open(golden_path(name), "rt").read())
    assert read_all(golden_path(name)) == want and len(want) > 0
    # small batches give the same records in the same order
    assert read_all(golden_path(name), max_bases=1000) == want
    assert read_all(golden_path(name), max_bases=0, max_records=3) == want


def test_edge_cases(tmp_path):
    cases = {
        "crlf": ">a one two\r\nACGT\r\nacgt\r\n\r\n>b\r\nNNNN\r\n",
        "no_final_newline": ">x\nACG\nTTT",
        "blank_and_indented": "\n\n>id   spaced   description  \n  ACGT  \n\n\tGG\n>empty\n>last\nA\n",
        "junk_before_header": "junk line\nmore\n>r1\nAC\n",
        "only_header": ">solo",
        "empty": "",
        "gt_inside": ">a\nAC>GT\n",
    }
    for name, text in cases.items():
        p = tmp_path / (name + ".fa")
        p.write_bytes(text.encode("ascii"))
        assert read_all(str(p)) == reference_parse(text), name
        assert read_all(str(p), max_bases=1) == reference_parse(text), name
    # plain text without any header is rejected, like the reference
    p = tmp_path / "notfasta.txt"; p.write_text("ACGT\nACGT\n")
    with pytest.raises(ValueError):
        reference_parse(p.read_text())
    with pytest.raises(ValueError):
        read_all(str(p))
    with pytest.raises(OSError):
        read_all(str(tmp_path / "missing.fa"))


def test_long_lines_multi_member_gzip_and_batches(tmp_path):
    import numpy as np
    from pyrodigal_amd import _cabi, benchdata
    recs = [("c%d" % i, "len=%d" % n, benchdata.synthetic_contig(n, 0.5, i).decode()) for i, n in enumerate([5, 70, 71, 100000, 9_000_000, 1, 33])]
    p = tmp_path / "multi.fa.gz"
    with open(p, "wb") as f:                      # two gzip members, one record on a single 9 Mbp line
        for part in (recs[:4], recs[4:]):
            text = "".join(">%s %s\n%s\n" % (i, d, s if len(s) > 1_000_000 else "\n".join(s[k:k + 70] for k in range(0, len(s), 70))) for i, d, s in part)
            f.write(gzip.compress(text.encode("ascii")))
    assert read_all(str(p)) == recs
    with _cabi.FastaReader(str(p)) as r:
        sizes = [[len(s) for _, _, s in b] for b in r.batches(max_bases=100)]
    assert sizes == [[5, 70, 71], [100000], [9_000_000], [1, 33]]
    assert sum(len(b) for b in sizes) == len(recs)


@pytest.mark.gpu
def test_fasta_file_to_genes_through_pinned_staging(tmp_path):
    """gzipped multi-record FASTA -> reader (pinned staging arenas, filled in turn) -> one DMA per batch ->
    genes; every record must get exactly the genes of a call on that record alone, whatever the batch it travelled in."""
    import gzip
    from pyrodigal_amd import _cabi, benchdata, pipeline
    from tests.util import read_fasta
    recs = []
    for name in ("SRR492066", "KK037166", "MIIJ01000039", "GCF_001457455.1_NCTC11397_genomic_100kb"):
        recs += [(h.split()[0], s) for h, s in read_fasta(name + ".fna.gz")]
    recs += [("synthetic_%d" % c, benchdata.synthetic_contig(3000 + 977 * c, 0.30 + 0.40 * (c % 41) / 40, 5000 + c).decode()) for c in range(60)]
    recs.insert(3, ("empty_record", ""))
    path = tmp_path / "mixed.fna.gz"
    with gzip.open(path, "wt") as f:
        for rid, seq in recs:
            f.write(">%s some description\n" % rid)
            for k in range(0, len(seq), 70):
                f.write(seq[k:k + 70] + "\n")
    models = [b for _, b in benchdata.load_model_set()]
    got = {}
    n_batches = 0
    for ids, descs, lens, res in pipeline.find_genes_fasta(str(path), models, n_contexts=2, max_bases=200_000, meta=True):
        n_batches += 1
        assert all(d == "some description" for d in descs)
        for i, rid in enumerate(ids):
            g = res.genes_of(i)
            got[rid] = (int(lens[i]), int(res.contigs[i]["model"]), g[["begin", "end", "strand", "start_ndx", "stop_ndx"]].tolist())
    assert n_batches >= 4 and list(got) == [r[0] for r in recs]
    ctx = _cabi.Context(0)
    try:
        ctx.set_models(models)
        for rid, seq in recs[::5] + recs[:6]:
            res = ctx.find_genes_batch([seq], meta=True)
            want = (len(seq), int(res.contigs[0]["model"]), res.genes[["begin", "end", "strand", "start_ndx", "stop_ndx"]].tolist())
            assert got[rid] == want, rid
        # the packed upload on its own: same batch, same result as the pointer-per-contig upload
        rd = _cabi.FastaReader(str(path))
        pb = next(rd.packed_batches(max_bases=0))
        seqs = [pb.sequence(i) for i in range(pb.n)]
        b = ctx.upload_packed(pb)
        a = ctx.find_genes(b, meta=True)
        b.close(); rd.close()
        w = ctx.find_genes_batch(seqs, meta=True)
        assert a.genes.tobytes() == w.genes.tobytes() and np.array_equal(a.contigs["model"], w.contigs["model"])
    finally:
        ctx.close()
