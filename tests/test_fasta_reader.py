"""The library's FASTA reader against a restatement of the parser the reference feeds its finder with
(ref: src/pyrodigal/tests/fasta.py:59-86): same records on the reference's own fixtures and on edge cases.
Host-side code: runs without a GPU."""
import gzip
import os

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
