"""Multi-GPU plumbing: contigs are independent, so the only exchange is a gather of gene records.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, over xGMI inside a node).
Each rank calls genes for its own contigs; `gather_genes` all-gathers the fixed-size packed
`pga_gene` records so that rank 0 (in fact every rank) holds the whole job's genes, contig ids
re-based to global numbering by the caller.  Volume is tiny (88 B per gene), latency-bound.
"""
import numpy as np


def shard_contigs(n_contigs, rank, world):
    """Round-robin contig -> rank assignment (contig c goes to rank c % world)."""
    return list(range(rank, n_contigs, world))


# nodes per bp of i.i.d. DNA at 30 / 40 / 50 / 60 / 65 / 70 % GC (probe): what the DP time follows
_NODE_DENSITY_GC = np.array([0.30, 0.40, 0.50, 0.60, 0.65, 0.70])
_NODE_DENSITY = np.array([0.014, 0.024, 0.037, 0.048, 0.052, 0.054])


def estimate_work(seqs, model_gcs=None, sample=8192):
    """Estimated node-passes of each contig: length x node density(gc) x models in its GC window.

    GC comes from a strided sample of at most ``sample`` bases per contig, so the estimate costs O(sample)
    per contig whatever its length.  ``model_gcs``: the GC of each loaded model (meta mode); None = 1 model."""
    work = np.zeros(len(seqs))
    mg = None if model_gcs is None else np.asarray(model_gcs, float)
    for i, s in enumerate(seqs):
        n = len(s)
        if n == 0:
            continue
        b = np.frombuffer(s if isinstance(s, (bytes, bytearray)) else s.encode("ascii"), np.uint8)
        if n > sample:
            b = b[:: n // sample]
        up = b & 0xDF                                       # upper case
        gc = float(np.count_nonzero((up == ord("G")) | (up == ord("C")))) / len(b)
        bins = 1
        if mg is not None:                                  # ref: lib.pyx:5335-5336
            low, high = min(0.65, 0.88495 * gc - 0.0102337), max(0.35, 0.86596 * gc + 0.1131991)
            bins = max(1, int(np.count_nonzero((mg >= low) & (mg <= high))))
        work[i] = n * float(np.interp(gc, _NODE_DENSITY_GC, _NODE_DENSITY)) * bins
    return work


def estimate_work_known(lengths, gcs, model_gcs=None):
    """`estimate_work` for contigs whose length and GC are known without looking at the bases (a synthetic job, or an index
    file next to the FASTA): every rank can plan the whole job before reading or generating a single contig."""
    lengths, gcs = np.asarray(lengths, float), np.asarray(gcs, float)
    bins = np.ones(len(lengths))
    if model_gcs is not None:                               # ref: lib.pyx:5335-5336
        mg = np.asarray(model_gcs, float)
        low = np.minimum(0.65, 0.88495 * gcs - 0.0102337)
        high = np.maximum(0.35, 0.86596 * gcs + 0.1131991)
        bins = np.maximum(1, ((mg[None, :] >= low[:, None]) & (mg[None, :] <= high[:, None])).sum(axis=1))
    return lengths * np.interp(gcs, _NODE_DENSITY_GC, _NODE_DENSITY) * bins


def pack_contigs(work, world):
    """Static greedy bin packing, largest first (LPT): returns, per rank, the ascending list of contig indices.

    Deterministic (ties broken by index) so that every rank computes the same assignment without talking."""
    work = np.asarray(work, float)
    order = sorted(range(len(work)), key=lambda i: (-work[i], i))
    load = [0.0] * world
    parts = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += work[i]
    return [sorted(p) for p in parts]


def find_genes_sharded(ctx, seqs, dist=None, device=None, model_gcs=None, **kw):
    """``ctx.find_genes_batch`` over this rank's share of ``seqs`` (the same list on every rank), then one gather.

    Returns (genes with global contig indices from every rank, this rank's BatchResult, this rank's contig indices)."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    mine = pack_contigs(estimate_work(seqs, model_gcs), world)[rank]
    res = ctx.find_genes_batch([seqs[i] for i in mine], **kw)
    genes = res.genes.copy()
    if len(genes):
        genes["contig"] = np.asarray(mine, np.int32)[genes["contig"]]
    return gather_genes(genes, dist, device), res, mine


def gather_genes(genes, dist=None, device=None, dst=None):
    """Gather a structured ``pga_gene`` array across ranks; returns the concatenation in rank order.

    ``dst=None``: all-gather, every rank holds the whole job's genes.  ``dst=r``: only rank ``r`` receives them (one
    gather over xGMI, and one device-to-host copy on that rank only); the other ranks get an empty array back."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return genes
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    on_gpu = dev.type == "cuda"
    n = torch.tensor([len(genes)], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    width = genes.dtype.itemsize
    cap = max(max(counts), 1)
    raw = np.ascontiguousarray(genes).view(np.uint8).reshape(-1)
    if on_gpu:
        # pinned staging on both sides of the exchange: the records cross PCIe at link speed
        stage = torch.empty(cap * width, dtype=torch.uint8, pin_memory=True)
        stage[: raw.size] = torch.from_numpy(raw)
        buf = stage.to(dev, non_blocking=True)
    else:
        buf = torch.zeros(cap * width, dtype=torch.uint8)
        buf[: raw.size] = torch.from_numpy(raw.copy())
    if dst is None:
        parts = [torch.empty(cap * width, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(parts, buf)
    else:
        parts = [torch.empty(cap * width, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
        dist.gather(buf, parts, dst=dst)
        if rank != dst:
            return genes[:0]
    total = sum(counts)
    out = np.empty(total, dtype=genes.dtype)
    flat = out.view(np.uint8).reshape(-1)
    host = torch.empty(total * width, dtype=torch.uint8, pin_memory=on_gpu)
    at = 0
    for p, c in zip(parts, counts):
        host[at: at + c * width].copy_(p[: c * width], non_blocking=on_gpu)
        at += c * width
    if on_gpu:
        torch.cuda.current_stream().synchronize()
    flat[:] = host.numpy()
    return out
