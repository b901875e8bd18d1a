"""World-size-2 gloo test of the only exchange in the multi-GPU path: the all-gather of packed
gene records (pyrodigal_amd/distributed.py), plus the contig sharding rule."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyrodigal_amd import _cabi, distributed
    n = 3 + 4 * rank                        # ragged: ranks hold different numbers of genes
    g = np.zeros(n, dtype=_cabi.GENE_DTYPE)
    g["contig"] = np.arange(n) * world + rank
    g["begin"] = 100 * rank + np.arange(n)
    g["cscore"] = rank + np.arange(n) / 7.0
    allg = distributed.gather_genes(g, dist)
    empty = distributed.gather_genes(np.zeros(0, dtype=_cabi.GENE_DTYPE), dist)
    np.save(os.path.join(out_dir, "r%d.npy" % rank), allg)
    assert len(empty) == 0
    only0 = distributed.gather_genes(g, dist, dst=0)              # gather to one rank: the others get nothing back
    assert (only0.tobytes() == allg.tobytes()) if rank == 0 else len(only0) == 0
    dist.barrier()
    dist.destroy_process_group()


def test_gather_genes_world2(tmp_path):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    world = 2
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    for name in a.dtype.names:                                      # every rank holds the whole job
        assert np.array_equal(a[name], b[name]), name               # (field-wise: np.save does not keep struct padding)
    assert len(a) == 3 + 7
    assert list(a["begin"][:3]) == [0, 1, 2] and list(a["begin"][3:6]) == [100, 101, 102]
    assert a["cscore"][4] == 1 + 1 / 7.0


def _worker8(rank, world, port, out_dir):
    """One rank of the job as the 8-GPU run shards it: every rank computes the same packing of the job from (length, GC) alone, takes
    its share, "finds" one record per contig of its share, renumbers to job-wide contig ids, and the records are gathered to rank 0."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyrodigal_amd import _cabi, benchdata, distributed
    lengths, gcs, _ = benchdata.config4_spec(37)                       # fewer contigs than make every rank busy at world 8 x ~5
    lengths = np.array(lengths); lengths[::5] = 0                      # empty contigs: no work, no records
    work = distributed.estimate_work_known(lengths, gcs, list(np.linspace(0.30, 0.70, 16)))
    parts = distributed.pack_contigs(work, world)
    mine = np.asarray(parts[rank], np.int32)
    if rank == 3:
        mine = mine[:0]                                                # a rank with nothing at all
    local = np.repeat(np.arange(len(mine)), [0 if lengths[c] == 0 else 1 + c % 3 for c in mine]).astype(np.int32)   # ragged: 1-3 genes per contig
    g = np.zeros(len(local), dtype=_cabi.GENE_DTYPE)
    if len(g):
        g["contig"] = mine[local]
        g["begin"] = 1000 * g["contig"] + np.arange(len(g))
        g["cscore"] = g["contig"] / 3.0
    got = distributed.gather_genes(g, dist, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "all.npy"), got)
        np.save(os.path.join(out_dir, "parts.npy"), np.array([len(p) for p in parts]))
    else:
        assert len(got) == 0
    allg = distributed.gather_genes(g, dist)                           # and the all-gather form
    assert len(allg) == sum(0 if (lengths[c] == 0 or r == 3) else 1 + c % 3 for r in range(world) for c in parts[r])
    dist.barrier()
    dist.destroy_process_group()


def test_pack_and_gather_world8_with_ragged_and_empty_ranks(tmp_path):
    """SURVEY 8(e) at the scale the driver launches: 8 ranks, the packing computed identically on each, ragged record counts,
    empty contigs, one rank with no contigs at all; rank 0 ends up with every record of the job, in rank order."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    a = np.load(tmp_path / "all.npy")
    from pyrodigal_amd import benchdata, distributed
    lengths, gcs, _ = benchdata.config4_spec(37)
    lengths = np.array(lengths); lengths[::5] = 0
    parts = distributed.pack_contigs(distributed.estimate_work_known(lengths, gcs, list(np.linspace(0.30, 0.70, 16))), 8)
    want = [c for r, p in enumerate(parts) if r != 3 for c in p for _ in range(0 if lengths[c] == 0 else 1 + c % 3)]
    assert list(a["contig"]) == want and len(want) > 20
    assert np.array_equal(a["cscore"], a["contig"] / 3.0) and np.all(a["begin"] // 1000 == a["contig"])
    assert sorted(c for p in parts for c in p) == list(range(37))


def test_shard_contigs_partition():
    from pyrodigal_amd import distributed
    for world in (1, 2, 4, 8):
        seen = sorted(c for r in range(world) for c in distributed.shard_contigs(1003, r, world))
        assert seen == list(range(1003))


def test_benchdata_is_deterministic_and_matches_spec():
    from pyrodigal_amd import benchdata
    a, b = benchdata.synthetic_contig(10000, 0.5, 1234), benchdata.synthetic_contig(10000, 0.5, 1234)
    assert a == b and set(a) <= set(b"ACGT")
    gc = sum(a.count(x) for x in b"GC") / len(a)
    assert abs(gc - 0.5) < 0.03
    models = benchdata.load_model_set()
    assert len(models) == 16 and all(len(m[1]) == 558392 for m in models)
    gcs = [np.frombuffer(m[1][:8], np.float64)[0] for m in models]
    assert gcs == sorted(gcs) and 0.29 < gcs[0] < 0.32 and 0.69 < gcs[-1] < 0.72


def test_pack_contigs_balances_by_estimated_work():
    """static greedy bin packing of contigs by estimated work, largest first, identical on every rank."""
    from pyrodigal_amd import benchdata, distributed
    rng = np.random.default_rng(5)
    lengths = (rng.pareto(1.5, 400) * 3000 + 500).astype(int)
    seqs = [benchdata.synthetic_contig(int(n), 0.30 + 0.40 * (i % 41) / 40, 900 + i) for i, n in enumerate(lengths)]
    seqs.append(b"")
    model_gcs = np.linspace(0.30, 0.70, 16)
    work = distributed.estimate_work(seqs, model_gcs)
    assert work[-1] == 0 and np.all(work[:-1] > 0)
    # more work for a GC-rich contig of the same length (denser nodes), and for more models in the window
    a, b = benchdata.synthetic_contig(20000, 0.35, 1), benchdata.synthetic_contig(20000, 0.65, 2)
    assert distributed.estimate_work([b])[0] > distributed.estimate_work([a])[0]
    assert distributed.estimate_work([a], model_gcs)[0] > distributed.estimate_work([a], model_gcs[:2])[0]
    for world in (1, 2, 4, 8):
        parts = distributed.pack_contigs(work, world)
        assert sorted(i for p in parts for i in p) == list(range(len(seqs)))
        assert all(p == sorted(p) for p in parts)
        loads = np.array([work[p].sum() for p in parts])
        assert loads.max() <= work.sum() / world + work.max() + 1e-9          # the LPT guarantee
        assert parts == distributed.pack_contigs(work, world)                  # deterministic
    rr = np.array([work[distributed.shard_contigs(len(seqs), r, 8)].sum() for r in range(8)])
    lpt = np.array([work[p].sum() for p in distributed.pack_contigs(work, 8)])
    assert lpt.max() <= rr.max()


def test_estimate_work_known_matches_the_sampled_estimate():
    from pyrodigal_amd import benchdata, distributed
    model_gcs = np.linspace(0.30, 0.70, 16)
    gcs = [0.30 + 0.40 * (c % 41) / 40 for c in range(60)]
    seqs = [benchdata.synthetic_contig(20_000, gc, 1_000_000 + c) for c, gc in enumerate(gcs)]
    known = distributed.estimate_work_known([20_000] * 60, gcs, model_gcs)
    sampled = distributed.estimate_work(seqs, model_gcs)
    assert np.all(known > 0) and np.median(np.abs(known - sampled) / sampled) < 0.05
