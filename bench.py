"""Headline benchmark: Mbp/s gene-called in meta mode (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the gene-finding path over this rank's batch, which is already
resident in HBM (`pga_batch_create`): digitise -> node extraction -> node scoring -> connection
scoring DP for every model in the contig's GC window -> winner -> genes in host memory.
Workload (config.workload): BASELINE.json configs[1], one 5 Mbp synthetic contig at 50 % GC per
rank (seed 1234 + rank), 16 custom metagenomic bins (see pyrodigal_amd/benchdata.py).
N > 1 (launched by torch.distributed.run): contigs are independent, so ranks share nothing in the
data path; rank 0 receives every rank's gene records through one RCCL all_gather ("weak" scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_NODE_PASS = 64.0     # compulsory SoA bytes per DP node-pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config2", choices=["config2", "config3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev                      # one process per GPU (ranks wrap only in single-GPU smoke tests)
    torch.cuda.set_device(dev_index)
    backend = os.environ.get("PGA_BENCH_BACKEND", "nccl")    # "nccl" is RCCL on ROCm; "gloo" for a single-GPU dry run
    xdev = torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=xdev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from pyrodigal_amd import _cabi, benchdata, distributed
    models = benchdata.load_model_set()
    ctx = _cabi.Context(dev_index)
    ctx.set_models([m[1] for m in models])
    if args.workload == "config2":
        seqs = benchdata.config2(rank)
        wname = "1x5Mbp_gc50_meta_per_gpu"
    else:
        seqs = benchdata.config3(1000, 50_000, first=1000 * rank)
        wname = "1000x50kbp_gc30-70_meta_per_gpu"
    bases = sum(len(s) for s in seqs)
    batch = ctx.upload(seqs)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # Bring the device out of its idle power state before the warmup steps proper: after a pause (the host was busy
    # generating the synthetic contigs) the first ~100 ms of work run at ramping clocks and would otherwise leak into
    # the timed steps when W is small.  Untimed, like the warmup.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.5:
        ctx.find_genes(batch, meta=True)
    res = None
    for _ in range(args.warmup):
        res = ctx.find_genes(batch, meta=True)
        distributed.gather_genes(res.genes, dist, device=xdev)
    sync()
    t0 = time.perf_counter()
    dp_ms, passes = 0.0, 0
    for _ in range(args.steps):
        res = ctx.find_genes(batch, meta=True)
        all_genes = distributed.gather_genes(res.genes, dist, device=xdev)
        dp_ms += res.t_dp_ms
        passes += res.node_passes
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * bases * args.steps / elapsed / 1e6
        achieved = BYTES_PER_NODE_PASS * passes / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0
        out = {
            "metric": "Mbp/sec gene-called (meta mode)", "value": round(value, 3), "unit": "Mbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wname, "contigs_per_gpu": len(seqs), "bases_per_gpu": bases, "models": len(models),
                       "node_passes_per_step": res.node_passes, "genes_rank0": int(len(res.genes)),
                       "genes_all_ranks": int(len(all_genes)), "parallelism": "contig-sharded x%d" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                         "kernel": "k_dp_tree_mw" if res.n_chains < 2048 else "k_dp_tree",
                         "kernel_ms_per_step": round(dp_ms / args.steps, 3), "chains": res.n_chains,
                         "bytes_per_node_pass": BYTES_PER_NODE_PASS},
        }
        seg = ctx.dp_stats()
        if seg["chains"] > 0:
            # few long chains: the connection scoring is one group of kernels (speculative segment walks, exact
            # re-scoring, verification; dp.hip "segmented chains"), timed as a whole by the same pair of events
            out["roofline"]["kernel"] = "connection scoring, segmented (k_dp_tree_mw + k_dp_rescore + k_dp_verify + helpers)"
            out["roofline"]["kernels"] = SEGMENTED_DP_KERNELS
            out["roofline"]["segments"] = seg["segments"]
            out["roofline"]["rejected_by_verification"] = seg["rejected"]
            out["roofline"]["chains_walked_serially"] = seg["serial"]
        out["roofline"]["traffic"] = pmc_traffic(wname)
        # PCIe-inclusive rate (upload + find), reported next to `value`, never as `value`
        t1 = time.perf_counter()
        ctx.find_genes_batch(seqs, meta=True)
        out["config"]["pcie_inclusive_Mbp_s"] = round(bases / (time.perf_counter() - t1) / 1e6, 3)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(seqs, models, res)
    batch.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


# every kernel of a segmented connection-scoring launch (pga_launch_dp with a plan), for the rocprof summaries
SEGMENTED_DP_KERNELS = ["k_dp_tree_mw", "k_seg_gather", "k_seg_weights", "k_seg_height", "k_spine_count",
                        "k_spine_scan", "k_spine_fill", "k_dp_rescore", "k_seg_leaves", "k_seg_build_far", "k_seg_build_upper",
                        "k_dp_verify"]


def pmc_traffic(workload):
    """HBM bytes per launch of the DP kernel from the separate rocprofv3 --pmc passes of this round
    (profiles/r01_pmc_traffic.json: FETCH_SIZE x 2 (gfx950 correction for wide loads) + WRITE_SIZE, in bytes)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return json.load(f).get(workload, {}).get("hbm_bytes_per_launch")
    except OSError:
        return None


def cpu_baseline(seqs, models, gpu_res):
    """The CPU oracle (a C port of pyrodigal's CPU path: byte pre-filter + split scorers), one thread,
    on a bounded sample of the same workload; also the gene-call parity check of this run."""
    from oracle import oracle as orc
    bins = [orc.Training(m[1]) for m in models]
    budget_bases = 5_000_000
    done, t_cpu, match, total_genes = 0, 0.0, True, 0
    for i, s in enumerate(seqs):
        if done >= budget_bases:
            break
        o = orc.Oracle(s)
        t0 = time.perf_counter()
        phase = o.find_genes_meta(bins)
        t_cpu += time.perf_counter() - t0
        done += len(s)
        og, gg = o.genes(), gpu_res.genes_of(i)
        total_genes += len(og)
        ok = gpu_res.contigs[i]["model"] == phase and len(og) == len(gg) and all(
            np.array_equal(og[k], gg[k]) for k in ("begin", "end", "start_ndx", "stop_ndx"))
        match = match and bool(ok)
    out = {"value": round(done / t_cpu / 1e6, 3), "unit": "Mbp/s", "cores": 1, "kind": "port",
           "sample": "%d contig(s), %d bp, same 16 models, meta mode, 1 thread (%s)" % (i + (done >= budget_bases), done, _cpu_name()),
           "gene_calls_identical_to_gpu": match, "genes_in_sample": total_genes, "host_cpus": os.cpu_count()}
    if len(seqs) > 1:
        out["all_cores"] = cpu_baseline_all_cores(seqs, bins)
    return out


def cpu_baseline_all_cores(seqs, bins):
    """The same oracle with one contig per thread on every host core (pyrodigal's ThreadPool model, ref: cli.py:289-302),
    on a sample sized for a few seconds.  A single contig does not spread over cores, so this is only reported for
    multi-contig workloads."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    sample = seqs[:min(len(seqs), 4 * cores)]

    def one(s):
        o = orc.Oracle(s)            # ctypes releases the GIL inside the C calls
        o.find_genes_meta(bins)
        return o.num_genes

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        genes = sum(ex.map(one, sample))
    dt = time.perf_counter() - t0
    bases = sum(len(s) for s in sample)
    return {"value": round(bases / dt / 1e6, 3), "unit": "Mbp/s", "cores": cores,
            "sample": "%d contig(s), %d bp, one contig per thread" % (len(sample), bases), "genes_in_sample": int(genes)}


def _cpu_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown cpu"


if __name__ == "__main__":
    main()
