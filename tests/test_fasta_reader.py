"""The library's FASTA reader against the records the REFERENCE's own parser yields (ref: src/pyrodigal/tests/fasta.py:59-86
`parse`, 16-57 `zopen`): tests/golden/fasta_records.json was written by tests/golden/make_fasta_records.py, which imports that
parser in the build container and runs it over the reference's sequence fixtures and over the edge-case inputs in
tests/golden/fasta/ (plain, CRLF, gzip, two-member gzip, bz2, xz).  Both sources of the C reader are checked -- the mapped,
multi-threaded one of plain files and the line-by-line one of streams -- unpacked and packed, at several batch budgets.
Host-side code: runs without a GPU (the packed form needs pinned memory, i.e. a HIP runtime, and is marked gpu).

One deliberate difference: the reference leaves the line terminator on a record's description; the reader strips it."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests.util import GOLDEN, golden_path

with open(os.path.join(GOLDEN, "fasta_records.json")) as _f:
    WANT = json.load(_f)


def read_all(path, **kw):
    from pyrodigal_amd import _cabi
    with _cabi.FastaReader(path) as r:
        return [(i, d, s.decode("ascii")) for b in r.batches(**kw) for i, d, s in b]


def check(name, got):
    want = WANT[name]
    if "records" in want:
        assert got == [(i, d.rstrip(), s) for i, d, s in want["records"]], name
    else:
        h = hashlib.sha256()
        for _, _, s in got:
            h.update(s.encode())
        assert [g[0] for g in got] == want["ids"] and [g[1] for g in got] == [d.rstrip() for d in want["descriptions"]], name
        assert [len(g[2]) for g in got] == want["lens"] and h.hexdigest() == want["sha256"], name


@pytest.mark.parametrize("name", sorted(WANT))
@pytest.mark.parametrize("source", ["default", "stream"])
def test_reader_yields_the_reference_parsers_records(name, source, monkeypatch):
    if source == "stream":
        monkeypatch.setenv("PGA_FASTA_NO_MMAP", "1")          # plain files through the line-by-line reader as well
    path = golden_path(name)
    if "error" in WANT[name]:
        with pytest.raises(ValueError):
            read_all(path)
        return
    check(name, read_all(path))
    # small batches give the same records in the same order
    check(name, read_all(path, max_bases=1000))
    check(name, read_all(path, max_bases=0, max_records=3))
    check(name, read_all(path, max_bases=1))


@pytest.mark.parametrize("threads", ["1", "3", "16"])
def test_mapped_reader_with_many_threads_and_batch_sizes(tmp_path, threads, monkeypatch):
    """The multi-threaded parse of a plain file: records of very different sizes (one larger than a thread's piece, empty ones,
    one-line and wrapped) cut into pieces at record boundaries, at budgets that end batches inside and between pieces."""
    from pyrodigal_amd import _cabi, benchdata
    monkeypatch.setenv("PGA_FASTA_THREADS", threads)
    sizes = [5, 70, 71, 0, 100_000, 4_000_000, 1, 33, 0, 1_500_000] + [20_000 + 977 * k for k in range(120)]
    recs = [("c%d" % i, "len=%d" % n, benchdata.synthetic_contig(n, 0.5, i).decode()) for i, n in enumerate(sizes)]
    p = tmp_path / "big.fa"
    with open(p, "w") as f:
        for i, d, s in recs:
            f.write(">%s %s\n" % (i, d))
            if len(s) > 3_000_000:
                f.write(s + "\n")                                   # one 4 Mbp line
            else:
                f.write("".join(s[k:k + 60] + "\r\n" for k in range(0, len(s), 60)))
    assert read_all(str(p)) == recs
    for budget in (1, 50_000, 1_000_000, 3_999_999, 10_000_000):
        assert read_all(str(p), max_bases=budget) == recs, budget
    with _cabi.FastaReader(str(p)) as r:
        got = [[len(s) for _, _, s in b] for b in r.batches(max_bases=100)]
    assert got[:3] == [[5, 70, 71], [0, 100_000], [4_000_000]] and sum(len(b) for b in got) == len(recs)
    assert read_all(str(p), max_bases=0, max_records=7) == recs


def test_missing_and_unsupported_files(tmp_path):
    with pytest.raises(OSError):
        read_all(str(tmp_path / "missing.fa"))
    # lz4 / zstd: sniffed like the reference does, and refused the same way when the Python module is not installed
    for magic, module in ((b"\x04\x22\x4d\x18", "lz4"), (b"\x28\xb5\x2f\xfd", "zstandard")):
        try:
            __import__(module)
            continue
        except ImportError:
            pass
        p = tmp_path / ("x." + module)
        p.write_bytes(magic + b"\0" * 32)
        with pytest.raises(RuntimeError):
            read_all(str(p))


def test_a_broken_compressed_stream_is_an_error_not_a_short_file(tmp_path):
    """What the decompressor raises inside the read callback must reach the caller: a bz2 file cut in the middle gives a ValueError
    whose cause is the decompressor's own exception -- not a silently truncated record set (ctypes would turn the exception into
    "end of stream")."""
    import bz2
    rng = np.random.default_rng(5)
    text = b"".join(b">r%d some words\n%s\n" % (k, bytes(rng.choice(list(b"ACGT"), 5000).astype(np.uint8))) for k in range(400))
    whole = bz2.compress(text)
    good = tmp_path / "good.fa.bz2"
    good.write_bytes(whole)
    assert len(read_all(str(good))) == 400
    cut = tmp_path / "cut.fa.bz2"
    cut.write_bytes(whole[:len(whole) // 2])
    with pytest.raises(ValueError) as info:
        read_all(str(cut))
    assert isinstance(info.value.__cause__, (EOFError, OSError))


@pytest.mark.gpu
def test_packed_batches_hold_the_same_records(tmp_path):
    """pga_fasta_next_packed (pinned staging arenas): same records, from the mapped and from the stream source."""
    from pyrodigal_amd import _cabi
    for name in ("fasta/multi.fa", "fasta/multi.fa.gz", "fasta/multi.fa.bz2", "fasta/blank_and_indented.fa", "MIIJ01000039.fna.gz"):
        for budget in (0, 1000):
            got = []
            with _cabi.FastaReader(golden_path(name)) as r:
                for pb in r.packed_batches(max_bases=budget):
                    got += [(pb.ids[i], pb.descriptions[i], pb.sequence(i).decode("ascii")) for i in range(pb.n)]
                    pb.release()
            check(name, got)


@pytest.mark.gpu
def test_fasta_file_to_genes_through_pinned_staging(tmp_path):
    """multi-record FASTA (gzip and plain) -> reader (pinned staging arenas, filled in turn) -> one DMA per batch ->
    genes; every record must get exactly the genes of a call on that record alone, whatever the batch it travelled in."""
    import gzip
    from pyrodigal_amd import _cabi, benchdata, pipeline
    from tests.util import read_fasta
    recs = []
    for name in ("SRR492066", "KK037166", "MIIJ01000039", "GCF_001457455.1_NCTC11397_genomic_100kb"):
        recs += [(h.split()[0], s) for h, s in read_fasta(name + ".fna.gz")]
    recs += [("synthetic_%d" % c, benchdata.synthetic_contig(3000 + 977 * c, 0.30 + 0.40 * (c % 41) / 40, 5000 + c).decode()) for c in range(60)]
    recs.insert(3, ("empty_record", ""))
    models = [b for _, b in benchdata.load_model_set()]
    ctx = _cabi.Context(0)
    try:
        ctx.set_models(models)
        want = {}
        for rid, seq in recs[::5] + recs[:6]:
            res = ctx.find_genes_batch([seq], meta=True)
            want[rid] = (len(seq), int(res.contigs[0]["model"]), res.genes[["begin", "end", "strand", "start_ndx", "stop_ndx"]].tolist())
        for suffix, opener in ((".fna.gz", gzip.open), (".fna", open)):
            path = tmp_path / ("mixed" + suffix)
            with opener(path, "wt") as f:
                for rid, seq in recs:
                    f.write(">%s some description\n" % rid)
                    for k in range(0, len(seq), 70):
                        f.write(seq[k:k + 70] + "\n")
            got = {}
            n_batches = 0
            for ids, descs, lens, res in pipeline.find_genes_fasta(str(path), models, n_contexts=2, max_bases=200_000, meta=True):
                n_batches += 1
                assert all(d == "some description" for d in descs)
                for i, rid in enumerate(ids):
                    g = res.genes_of(i)
                    got[rid] = (int(lens[i]), int(res.contigs[i]["model"]), g[["begin", "end", "strand", "start_ndx", "stop_ndx"]].tolist())
            assert n_batches >= 4 and list(got) == [r[0] for r in recs]
            for rid in want:
                assert got[rid] == want[rid], (suffix, rid)
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
