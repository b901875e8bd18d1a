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


def gather_genes(genes, dist=None, device=None):
    """All-gather a structured ``pga_gene`` array across ranks; returns the concatenation in rank order."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return genes
    import torch
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([len(genes)], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    width = genes.dtype.itemsize
    cap = max(max(counts), 1)
    buf = torch.zeros(cap * width, dtype=torch.uint8, device=dev)
    if len(genes):
        raw = np.ascontiguousarray(genes).view(np.uint8).reshape(-1)
        buf[: raw.size] = torch.from_numpy(raw.copy()).to(dev)
    parts = [torch.zeros(cap * width, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = [p[: c * width].cpu().numpy().view(genes.dtype) for p, c in zip(parts, counts)]
    return np.concatenate(out) if out else genes
