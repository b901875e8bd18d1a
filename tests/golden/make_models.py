"""Generates the 13 synthetic-genome TrainingInfo fixtures used as metagenomic bins by the
benchmark and the batch tests (the reference's 50 built-in models are not
in the checkout, so meta mode runs on custom bins).

Each model is trained by the CPU oracle's `train()` (itself pinned byte-for-byte on the
reference's TrainingInfo fixtures) on a 2 Mbp planted-ORF genome:
  intergenic spacer ~ geometric(mean 120 bp), i.i.d. bases at the target GC;
  ORF = ATG + L sense codons (L ~ geometric(mean 300), codons drawn from a GC-matched
  distribution without stop codons) + TAA; strand +/- with p = 0.5.
Run from the repo root:  python tests/golden/make_models.py
"""
import gzip
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "models")
COMP = bytes.maketrans(b"ACGT", b"TGCA")


from pyrodigal_amd.benchdata import planted_contig as planted_genome  # noqa: E402  (the generator lives with the other workload generators)


def main():
    os.makedirs(OUT, exist_ok=True)
    gcs = [0.30 + 0.40 * k / 12 for k in range(13)]
    for k, gc in enumerate(gcs):
        tt = 4 if k == 6 else 11           # one translation-table-4 model exercises the re-extraction branch
        seq = planted_genome(2_000_000, gc, 4242 + k)
        t = orc.Oracle(seq).train(tt=tt)
        name = "synth_gc%02d_tt%d.tinf.bin.gz" % (round(gc * 100), tt)
        with gzip.GzipFile(os.path.join(OUT, name), "wb", compresslevel=9, mtime=0) as f:
            f.write(t.tobytes())
        print(name, "gc=%.4f uses_sd=%d" % (t.gc, t.uses_sd), os.path.getsize(os.path.join(OUT, name)))


if __name__ == "__main__":
    main()
