"""Golden records for the library's FASTA reader, produced by the REFERENCE's own parser.

Run in the build container (where /root/reference exists):   python tests/golden/make_fasta_records.py
It imports /root/reference/src/pyrodigal/tests/fasta.py (the reader the reference's tests and CLI feed GeneFinder with), writes
a set of small edge-case inputs under tests/golden/fasta/ (plain, CRLF, gzip, multi-member gzip, bz2, xz), parses them and the
reference's own sequence fixtures with that parser, and stores what it yields in tests/golden/fasta_records.json:
    {file: {"records": [[id, description, sequence], ...]}}            small inputs, verbatim
    {file: {"n": .., "ids": [..], "lens": [..], "sha256": ..}}          large fixtures: ids, lengths and a digest of the sequences
    {file: {"error": "ValueError"}}                                     inputs the reference rejects
Only data travels: the JSON and the input files; the reference's Python file is never copied."""
import bz2
import gzip
import hashlib
import importlib.util
import json
import lzma
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/pyrodigal/tests/fasta.py"
OUT_DIR = os.path.join(HERE, "fasta")

CASES = {
    "crlf.fa": b">a one two\r\nACGT\r\nacgt\r\n\r\n>b\r\nNNNN\r\n",
    "no_final_newline.fa": b">x\nACG\nTTT",
    "blank_and_indented.fa": b"\n\n>id   spaced   description  \n  ACGT  \n\n\tGG\n>empty\n>last\nA\n",
    "junk_before_header.fa": b"junk line\nmore\n>r1\nAC\n",
    "only_header.fa": b">solo",
    "gt_inside.fa": b">a\nAC>GT\n",
    "bare_gt.fa": b">\nACGT\n> spaced id\nTT\n",
    "interior_blanks.fa": b">s keeps interior blanks\nAC GT\n  N N  \n",
    "not_fasta.txt": b"ACGT\nACGT\n",
    "whitespace_only.fa": b"\n  \n\t\n",
}


def synthetic(n, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].tobytes()


def main():
    spec = importlib.util.spec_from_file_location("ref_fasta", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    os.makedirs(OUT_DIR, exist_ok=True)
    # a multi-record file with wrapped lines, one very long line and an empty record: plain, gzip, two-member gzip, bz2, xz
    recs = [("c%d" % i, "len=%d" % n, synthetic(n, i).decode()) for i, n in enumerate([5, 70, 71, 3000, 200_000, 1, 33])]
    recs.insert(3, ("void", "nothing here", ""))
    text = "".join(">%s %s\n%s" % (i, d, (s + "\n" if len(s) > 100_000 else "".join(s[k:k + 70] + "\n" for k in range(0, len(s), 70)))) for i, d, s in recs).encode()
    files = dict(CASES)
    files["multi.fa"] = text
    files["multi.fa.gz"] = gzip.compress(text, mtime=0)
    half = text.index(b">c3")
    files["multi_two_members.fa.gz"] = gzip.compress(text[:half], mtime=0) + gzip.compress(text[half:], mtime=0)
    files["multi.fa.bz2"] = bz2.compress(text)
    files["multi.fa.xz"] = lzma.compress(text)
    for name, data in files.items():
        with open(os.path.join(OUT_DIR, name), "wb") as f:
            f.write(data)
    out = {}
    def digest(rs):
        h = hashlib.sha256()
        for r in rs:
            h.update(r.seq.encode())
        return {"n": len(rs), "ids": [r.id for r in rs], "descriptions": [r.description for r in rs], "lens": [len(r.seq) for r in rs],
                "sha256": h.hexdigest()}

    for name in sorted(files):
        try:
            rs = list(ref.parse(os.path.join(OUT_DIR, name)))
            out["fasta/" + name] = {"records": [[r.id, r.description, r.seq] for r in rs]} if sum(len(r.seq) for r in rs) <= 4096 else digest(rs)
        except ValueError:
            out["fasta/" + name] = {"error": "ValueError"}
    for name in ("SRR492066.fna.gz", "KK037166.fna.gz", "MIIJ01000039.fna.gz", "GCF_001457455.1_NCTC11397_genomic.fna.gz", "SRR492066.single.faa.gz"):
        out[name] = digest(list(ref.parse(os.path.join(HERE, name))))
    with open(os.path.join(HERE, "fasta_records.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", len(out), "entries")


if __name__ == "__main__":
    main()
