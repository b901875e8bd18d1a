"""Randomised cross-check of the two independent DP implementations (tree kernels vs the window-scanning kernel) and of
the device / host tails on gene-dense synthetic contigs (planted ORFs on both strands, overlapping genes, runs of N),
plus an oracle comparison on a sample.  Any divergence between the implementations fails the test."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import synthetic_contig

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _planted():
    spec = importlib.util.spec_from_file_location("make_models", os.path.join(ROOT, "tests", "golden", "make_models.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.planted_genome


def _contigs():
    planted = _planted()
    rng = np.random.default_rng(2024)
    out = []
    for k in range(160):
        L = int(rng.choice([400, 1500, 5000, 12000, 40000, 90000], p=[0.1, 0.15, 0.25, 0.25, 0.2, 0.05]))
        gc = float(rng.uniform(0.25, 0.75))
        s = bytearray(planted(L, gc, 5000 + k) if k % 3 else synthetic_contig(L, gc, 5000 + k))
        if isinstance(s, (str,)):
            s = bytearray(s.encode())
        if k % 5 == 0 and L > 1000:                       # unknown bases, short and long runs
            for _ in range(3):
                at = int(rng.integers(0, L - 200)); n = int(rng.choice([1, 7, 60, 150]))
                s[at:at + n] = b"N" * n
        out.append(bytes(s))
    return out


def test_dp_kernels_and_tails_agree_on_random_gene_dense_contigs(monkeypatch):
    from pyrodigal_amd import _cabi, benchdata
    models = [b for _, b in benchdata.load_model_set()]
    seqs = _contigs()
    ctx = _cabi.Context(0)
    ctx.set_models(models)
    runs = {}
    for name, env in (("tree+auto", {}), ("scan+host", {"PGA_DP_KERNEL": "scan", "PGA_TAIL": "host"}),
                      ("tree1+device", {"PGA_DP_KERNEL": "tree1", "PGA_TAIL": "device"}), ("tree3+host", {"PGA_DP_KERNEL": "tree3", "PGA_TAIL": "host"}),
                      # the kernels every headline number comes from, forced onto this small launch: the wave-batch connection
                      # scorer and the LDS-table form of the coding score
                      ("wave+ldscs", {"PGA_DP_KERNEL": "wave", "PGA_CS_LDS": "2"}), ("wavedyn", {"PGA_DP_KERNEL": "wave", "PGA_DPW_SCHED": "0"}),
                      # the coding score by per-lane table gathers (the fallback when a contig's models are not neighbours in the table)
                      ("wave+topowalk", {"PGA_DP_KERNEL": "wave", "PGA_DPW_TOPO_WALK": "1"}),
                      # the topology per node from global memory instead of per contig from LDS copies
                      ("wave+topoglobal", {"PGA_DP_KERNEL": "wave", "PGA_DPW_TOPO_LDS": "0"}),
                      ("tree+globalcs", {"PGA_CS_LDS": "0"}), ("scan+globalcs", {"PGA_DP_KERNEL": "scan", "PGA_TAIL": "host", "PGA_CS_LDS": "0"}),
                      # the start scorer walking a workgroup's models three to a pass (its path for more than 512 models)
                      ("tree+3models/pass", {"PGA_SS_MODELS_PER_PASS": "3"}),
                      # the start scorer with a thread per node (what stage-level calls run) instead of over the list of start nodes
                      ("wave+thread-per-node", {"PGA_DP_KERNEL": "wave", "PGA_SS_STARTS_ONLY": "0"})):
        for k in ("PGA_DP_KERNEL", "PGA_TAIL", "PGA_CS_LDS", "PGA_SS_MODELS_PER_PASS", "PGA_DPW_TOPO_WALK", "PGA_DPW_SCHED", "PGA_DPW_TOPO_LDS", "PGA_SS_STARTS_ONLY"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        runs[name] = ctx.find_genes_batch(seqs, meta=True, want_nodes=True)
    base = runs["tree+auto"]
    assert len(base.genes) > 1500
    for name, r in runs.items():
        assert r.genes.tobytes() == base.genes.tobytes(), name
        assert np.array_equal(r.contigs["model"], base.contigs["model"]), name
        for a, b in zip(r.nodes, base.nodes):
            for f in ("traceb", "tracef", "ov_mark", "elim"):
                assert np.array_equal(a[f], b[f]), (name, f)
            assert np.array_equal(a["score"].view(np.uint64), b["score"].view(np.uint64)) and np.array_equal(a["sscore"].view(np.uint64), b["sscore"].view(np.uint64)), name
            for f in ("cscore", "rscore", "uscore", "tscore"):
                assert np.array_equal(a[f].view(np.uint64), b[f].view(np.uint64)), (name, f)
    # without the node arrays the winners are gathered without their final-pass fields and the gene records fetch them per gene
    for k in ("PGA_DP_KERNEL", "PGA_TAIL", "PGA_CS_LDS", "PGA_SS_MODELS_PER_PASS", "PGA_DPW_TOPO_WALK", "PGA_DPW_SCHED", "PGA_DPW_TOPO_LDS", "PGA_SS_STARTS_ONLY"):
        monkeypatch.delenv(k, raising=False)
    assert ctx.find_genes_batch(seqs, meta=True).genes.tobytes() == base.genes.tobytes()
    # single mode with masking as well
    for k in ("PGA_DP_KERNEL", "PGA_TAIL", "PGA_CS_LDS", "PGA_SS_MODELS_PER_PASS", "PGA_DPW_TOPO_WALK", "PGA_DPW_SCHED", "PGA_DPW_TOPO_LDS", "PGA_SS_STARTS_ONLY"):
        monkeypatch.delenv(k, raising=False)
    ctx.set_models(models[7:8])
    s1 = ctx.find_genes_batch(seqs, meta=False, mask=True, closed=True)
    monkeypatch.setenv("PGA_DP_KERNEL", "scan"); monkeypatch.setenv("PGA_TAIL", "host")
    s2 = ctx.find_genes_batch(seqs, meta=False, mask=True, closed=True)
    assert s1.genes.tobytes() == s2.genes.tobytes() and len(s1.genes) > 500
    # ... and with the other form of the coding score (per-lane table gathers)
    monkeypatch.delenv("PGA_DP_KERNEL"); monkeypatch.delenv("PGA_TAIL"); monkeypatch.setenv("PGA_CS_LDS", "0")
    s3 = ctx.find_genes_batch(seqs, meta=False, mask=True, closed=True)
    monkeypatch.delenv("PGA_CS_LDS")
    assert s3.genes.tobytes() == s1.genes.tobytes()
    # and the oracle on a sample of the meta-mode run
    bins = [orc.Training(b) for b in models]
    for i in range(0, len(seqs), 9):
        o = orc.Oracle(seqs[i])
        assert o.find_genes_meta(bins) == base.contigs[i]["model"]
        og, gg = o.genes(), base.genes_of(i)
        assert len(og) == len(gg) and all(np.array_equal(og[k], gg[k]) for k in ("begin", "end", "start_ndx", "stop_ndx"))
    ctx.close()
