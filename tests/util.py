"""Shared helpers for the test-suite: fixtures loading, FASTA parsing, synthetic contigs."""
import gzip
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_fasta(name):
    """Returns [(header, sequence)] of a (gzipped) FASTA fixture."""
    path = os.path.join(GOLDEN, name)
    opener = gzip.open if path.endswith(".gz") else open
    recs, hdr, buf = [], None, []
    with opener(path, "rt") as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                if hdr is not None:
                    recs.append((hdr, "".join(buf)))
                hdr, buf = line[1:], []
            elif line:
                buf.append(line)
    if hdr is not None:
        recs.append((hdr, "".join(buf)))
    return recs


def golden_path(name):
    return os.path.join(GOLDEN, name)


def parse_prodigal_header(hdr):
    """'>id # begin # end # strand # ID=..;partial=..;start_type=..;rbs_motif=..;rbs_spacer=..;gc_cont=..'"""
    f = [x.strip() for x in hdr.split(" # ")]
    attrs = dict(kv.split("=", 1) for kv in f[4].split(";") if kv)
    return (int(f[1]), int(f[2]), int(f[3]), attrs["partial"], attrs["start_type"],
            attrs["rbs_motif"], attrs["rbs_spacer"], attrs["gc_cont"])


_COMP = str.maketrans("ACGTN", "TGCAN")


def gene_sequence(seq, begin, end, strand):
    """Nucleotides of a gene (1-based inclusive coords), unknown letters rendered N."""
    s = "".join(c if c in "ACGT" else "N" for c in seq[begin - 1:end].upper())
    return s if strand == 1 else s.translate(_COMP)[::-1]


def synthetic_contig(length, gc, seed):
    """i.i.d. bases, P(G)=P(C)=gc/2 (SURVEY.md section 8d generator)."""
    rng = np.random.default_rng(seed)
    p = [(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, size=length, p=p)].tobytes()
