"""Two ranks (gloo rendezvous, both on cuda:0 -- the test box has one GPU) shard a batch by estimated work,
call genes for their share through the C-ABI and gather: every rank must end up with exactly the genes a
single process finds for the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _batch():
    from pyrodigal_amd import benchdata
    return [benchdata.synthetic_contig(4000 + 1500 * (c % 7), 0.32 + 0.36 * (c % 13) / 12, 300 + c) for c in range(24)]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyrodigal_amd import _cabi, benchdata, distributed
    models = benchdata.load_model_set()
    ctx = _cabi.Context(0)
    ctx.set_models([b for _, b in models])
    gcs = [float(np.frombuffer(b[:8], np.float64)[0]) for _, b in models]
    genes, res, mine = distributed.find_genes_sharded(ctx, _batch(), dist, model_gcs=gcs, meta=True)
    np.save(os.path.join(out_dir, "g%d.npy" % rank), genes)
    np.save(os.path.join(out_dir, "m%d.npy" % rank), np.asarray(mine))
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process(tmp_path):
    from pyrodigal_amd import _cabi, benchdata
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    m0, m1 = np.load(tmp_path / "m0.npy"), np.load(tmp_path / "m1.npy")
    assert sorted(list(m0) + list(m1)) == list(range(24)) and len(m0) > 0 and len(m1) > 0
    ctx = _cabi.Context(0)
    ctx.set_models([b for _, b in benchdata.load_model_set()])
    ref = ctx.find_genes_batch(_batch(), meta=True).genes
    ctx.close()
    key = ["contig", "begin", "end", "strand", "start_ndx", "stop_ndx", "cscore", "sscore"]
    for g in (g0, g1):
        order = np.lexsort((g["begin"], g["contig"]))
        assert len(g) == len(ref) > 0
        for k in key:
            assert np.array_equal(g[k][order], ref[k]), k


def test_streamed_batches_over_two_contexts_equal_sequential_calls():
    from pyrodigal_amd import _cabi, benchdata, pipeline
    models = [b for _, b in benchdata.load_model_set()]
    batches = [[benchdata.synthetic_contig(3000 + 700 * ((7 * k + c) % 9), 0.35 + 0.03 * ((k + c) % 10), 50 * k + c) for c in range(20)]
               for k in range(7)] + [[]]
    ctx = _cabi.Context(0)
    ctx.set_models(models)
    want = [ctx.find_genes_batch(b, meta=True).genes for b in batches]
    ctx.close()
    got = list(pipeline.find_genes_stream(iter(batches), models, n_contexts=2, meta=True))
    assert len(got) == len(batches)
    for (b, res), w, orig in zip(got, want, batches):
        assert b is orig and res.genes.tobytes() == w.tobytes()
    with pytest.raises(ValueError):          # an error inside a worker reaches the consumer
        list(pipeline.find_genes_stream([["ACGT" * 100]], models, n_contexts=2, meta=True, min_gene=-5))
