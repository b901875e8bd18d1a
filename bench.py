"""Headline benchmark: Mbp/s gene-called in meta mode (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload config4|config3|config2|config5]

Workload (config.workload), the same fixed job at every N ("strong" scaling): BASELINE.json configs[3], the
100 000 x 20 kbp metagenome-like contigs (seed 1 000 000 + c, GC 30..70 %), meta mode over 16 custom
metagenomic bins (pyrodigal_amd/benchdata.py).  Every rank plans the whole job from the contigs' (length, GC) alone
(static greedy packing by estimated node-passes, pyrodigal_amd/distributed.py), generates only its own contigs and keeps
them resident in HBM (`pga_batch_create`, sub-batches of --sub-batch contigs).  One "step" = one full pass of the
gene-finding path over the whole job: per sub-batch digitise -> node extraction -> node scoring -> connection scoring
for every model in the contig's GC window -> winner -> genes in host memory, then ONE gather of the packed gene records
to rank 0 (RCCL over xGMI; contigs are independent, so there is no collective in the data path).
`value` = bases of the whole job / max-over-ranks step time of the HOST-TO-HOST loop (SURVEY 8d's definition of the metric:
ASCII contigs in host memory -> packing into pinned memory -> H2D -> path -> gene records in host memory -> gather), timed over
all --steps.  `config.resident_Mbp_s` is the same loop with the contigs already resident in HBM (no packing, no upload).
At N = 1 the other single-GPU configurations of BASELINE.json (configs[1], [2], [4]) are timed as well (`secondary`),
each with its own connection-scoring roofline.
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
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 2.0   # wave64 VALU instructions per second the chip can issue (256 CUs x 4 SIMD-32)
PMC_FILES = [os.path.join(ROOT, "profiles", n) for n in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json")]     # newest first

# every kernel of a segmented connection-scoring launch (pga_launch_dp with a plan), for the rocprof summaries
SEGMENTED_DP_KERNELS = ["k_dp_tree_mw", "k_dp_wave", "k_seg_gather", "k_seg_weights", "k_seg_height", "k_spine_count",
                        "k_spine_scan", "k_spine_fill", "k_dp_rescore", "k_seg_leaves", "k_seg_build_far", "k_seg_build_upper",
                        "k_dp_verify"]


def roofline(ctx, dp_ms, passes, calls, n_chains, wname, launch_key=None, aux_ms=None):
    """Connection scoring against the HBM roofline: 64 B x node-passes / kernel time (HIP events on the library's stream,
    summed over the calls of the timed region)."""
    achieved = BYTES_PER_NODE_PASS * passes / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0
    pmc = pmc_entry(launch_key or wname)
    r = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": pmc.get("hbm_bytes_per_launch"),
         "traffic_source": ("%s (separate rocprofv3 --pmc passes, collected at commit %s)" % (pmc.get("_file"), pmc.get("collected_at_commit")))
                           if pmc.get("hbm_bytes_per_launch") else None,
         "kernel": "k_dp_tree_mw" if n_chains < 2048 else ctx.dp_kernel_name(),
         "kernel_ms_per_launch": round(dp_ms / max(calls, 1), 4), "launches": calls,
         "node_passes_per_launch": int(passes // max(calls, 1)), "chains_per_launch": n_chains,
         "bytes_per_node_pass": BYTES_PER_NODE_PASS}
    # SURVEY 8(d) ii: the kernel is scan / compare work, so next to the HBM fraction goes the issue-rate figure: VALU
    # wave-instructions per second (SQ_INSTS_VALU of the --pmc pass over this run's kernel time) against the chip's VALU issue
    # peak, 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md "Wave scheduling", v_fma_f32 row)
    if pmc.get("valu_insts_per_launch") and dp_ms > 0:
        per_s = pmc["valu_insts_per_launch"] * max(calls, 1) / (dp_ms * 1e-3)
        r["valu_issue_frac"] = round(per_s / VALU_ISSUE_PEAK, 4)
        r["valu_insts_per_node_pass"] = round(pmc["valu_insts_per_launch"] * max(calls, 1) / max(passes, 1), 2)
    # ... and what the pipes say themselves (SQ_ACTIVE_INST_VALU / _SCA of a --pmc pass, x 4 = SIMD-cycles with a vector / scalar
    # instruction of the kernel executing) over the SIMD-cycles of this run's kernel time: a wave64 f64 or cross-lane instruction
    # holds the pipe longer than the two cycles the issue peak above assumes, so this is the figure that says "issue-bound"
    if pmc.get("valu_busy_simd_cycles_per_launch") and dp_ms > 0:
        simd_cycles = 1024 * 2.4e9 * dp_ms * 1e-3
        r["valu_busy_frac"] = round(pmc["valu_busy_simd_cycles_per_launch"] * max(calls, 1) / simd_cycles, 4)
        r["scalar_busy_frac"] = round(pmc["scalar_busy_simd_cycles_per_launch"] * max(calls, 1) / simd_cycles, 4)
        r["salu_insts_per_node_pass"] = round(pmc["salu_insts_per_launch"] * max(calls, 1) / max(passes, 1), 2)
        r["branch_insts_per_node_pass"] = round(pmc["branch_insts_per_launch"] * max(calls, 1) / max(passes, 1), 2)
    if aux_ms is not None and dp_ms > 0:
        # The lane masks of the pair steps are compiled once per (contig, translation table) by k_dpw_sched, from the topology arrays of
        # k_dpw_topo(_lds): both run earlier in the call, outside the events of the scoring launch, and both are connection scoring
        # (ref: ConnectionScorer.index, lib.pyx:1126-1176).  The same fraction with their time (HIP events, pga_dp_timings) counted in:
        topo_ms, sched_ms = aux_ms
        r["topology_kernels_ms_per_launch"] = round(topo_ms / max(calls, 1), 4)
        r["schedule_kernels_ms_per_launch"] = round(sched_ms / max(calls, 1), 4)
        r["frac_incl_schedule"] = round(BYTES_PER_NODE_PASS * passes / ((dp_ms + topo_ms + sched_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)
    seg = ctx.dp_stats()
    if seg["chains"] > 0:
        # few long chains: the connection scoring is one group of kernels (speculative segment walks, exact
        # re-scoring, verification; dp.hip "segmented chains"), timed as a whole by the same pair of events
        r["kernel"] = "connection scoring, segmented (k_dp_tree_mw or, for very long chains, k_dp_wave over the segments + k_dp_rescore + k_dp_verify + helpers)"
        r["kernels"] = SEGMENTED_DP_KERNELS
        r["segments"] = seg["segments"]
        r["rejected_by_verification"] = seg["rejected"]
        r["chains_walked_serially"] = seg["serial"]
    return r


def pmc_entry(workload):
    """What the separate rocprofv3 --pmc passes of this round measured for the connection-scoring launch of this workload
    (profiles/r03_pmc_traffic.json: HBM bytes and VALU wave-instructions per launch, the commit they were collected at and how
    they were corrected).  Counters cannot be collected inside a timed run, so this is a lookup: it is only valid while the
    kernel has not changed since `collected_at_commit`, which the JSON line repeats."""
    for path in PMC_FILES:
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if workload not in d:
            continue
        e = dict(d[workload])
        e.setdefault("collected_at_commit", d.get("collected_at_commit"))        # per entry since the file is refreshed one workload at a time
        e["_file"] = os.path.relpath(path, ROOT)
        return e
    return {}


def pipeline_entry(workload):
    """SURVEY 8(d)'s whole-pipeline figures of one device call of this workload -- HBM bytes per base over ALL kernels of the call
    and every kernel's milliseconds per call -- as tools/pipeline_traffic.py computed them from the kernel trace and the
    FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh.  A lookup like `pmc_entry`, valid for `collected_at_commit`."""
    e = pmc_entry("pipeline:" + workload)
    if not e:
        return None
    kern = e.get("kernels", {})
    return {"pipeline_hbm_bytes_per_bp": e.get("pipeline_hbm_bytes_per_bp"), "kernel_ms_per_call": e.get("kernel_ms_per_call"),
            "hbm_GB_per_call": e.get("pipeline_hbm_GB_per_call"), "bases_per_call": e.get("bases_per_call"),
            "kernel_ms": {k: v["ms_per_call"] for k, v in kern.items() if v["ms_per_call"] >= 0.02},
            "kernel_hbm_MB": {k: v["hbm_MB_per_call"] for k, v in kern.items() if v["hbm_MB_per_call"] >= 50.0},
            "source": "%s (%s, collected at commit %s)" % (e.get("_file"), e.get("method", ""), e.get("collected_at_commit"))}


class Lanes:
    """The device calls of one pass, dealt to C contexts (one HIP stream and one set of scratch buffers each) and issued from C host
    threads: while one call is in a host-side phase (upload, chain planning, unpacking) the other contexts' kernels run.
    Batch k belongs to context k % C; results come back in batch order."""

    def __init__(self, ctxs):
        from concurrent.futures import ThreadPoolExecutor
        self.ctxs = ctxs
        self.pool = ThreadPoolExecutor(len(ctxs)) if len(ctxs) > 1 else None

    def run(self, n, call):
        """``call(ctx, k)`` for k in range(n)."""
        C = len(self.ctxs)
        out = [None] * n
        if self.pool is None:
            for k in range(n):
                out[k] = call(self.ctxs[0], k)
            return out

        def lane(c):
            for k in range(c, n, C):
                out[k] = call(self.ctxs[c], k)
        for f in [self.pool.submit(lane, c) for c in range(C)]:
            f.result()
        return out

    def run_back_to_back(self, steps, n, call, done):
        """`steps` passes of n calls without a join between them: context c goes on with its first call of pass s + 1 as soon as it
        has made its last of pass s, and ``done(results of pass s)`` runs (in pass order, on the caller's thread) as soon as the
        last call of pass s has returned -- the pipeline of calls is filled once, not once per pass."""
        import threading
        C = len(self.ctxs)
        out = [[None] * n for _ in range(steps)]
        left = [n] * steps
        cv = threading.Condition()

        failed = []

        def lane(c):
            try:
                for j in range(c, steps * n, C):
                    if failed:
                        return
                    s_, k = divmod(j, n)
                    r = call(self.ctxs[c % C], k)
                    with cv:
                        out[s_][k] = r
                        left[s_] -= 1
                        if left[s_] == 0:
                            cv.notify_all()
            except BaseException as err:          # a failed call must not leave the caller waiting for its pass
                with cv:
                    failed.append(err)
                    cv.notify_all()
        if self.pool is None:
            for s_ in range(steps):
                done([call(self.ctxs[0], k) for k in range(n)])
            return
        futs = [self.pool.submit(lane, c) for c in range(C)]
        for s_ in range(steps):
            with cv:
                while left[s_] > 0 and not failed:
                    cv.wait()
            if failed:
                break
            done(out[s_])
            out[s_] = None
        for f in futs:
            f.result()
        if failed:
            raise failed[0]

    def close(self):
        if self.pool is not None:
            self.pool.shutdown()


def timed_steps(lanes, batches, steps, warmup, sync, gather, post=None, **kw):
    """W untimed + K timed passes over the resident batches; returns (seconds, dp_ms, node_passes, calls, last results).
    post(result, k): what the calling thread does with a call's result right behind the call."""
    one = (lambda ctx, k: ctx.find_genes(batches[k], **kw)) if post is None else (lambda ctx, k: post(ctx.find_genes(batches[k], **kw), k))
    res = []
    for _ in range(warmup):
        res = lanes.run(len(batches), one)
        gather(res)
    sync()
    t0 = time.perf_counter()
    dp_ms, passes, calls = 0.0, 0, 0
    genes = None
    for _ in range(steps):
        res = lanes.run(len(batches), one)
        genes = gather(res)
        for r in res:
            dp_ms += r.t_dp_ms; passes += r.node_passes; calls += 1
    sync()
    return time.perf_counter() - t0, dp_ms, passes, calls, res, genes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="config4", choices=["config4", "config3", "config2", "config5"])
    ap.add_argument("--contigs", type=int, default=100_000, help="config4: contigs of the whole job")
    ap.add_argument("--series", default="iid", choices=["iid", "planted"],
                    help="config4: i.i.d. bases (BASELINE.json's generator) or planted ORFs (SURVEY 8d's metagenome-like series)")
    ap.add_argument("--sub-batch", type=int, default=6_250, help="contigs per device call")
    ap.add_argument("--contexts", type=int, default=4, help="device contexts (streams) the calls of a pass are dealt to (four: DESIGN 5.1, round 5)")
    ap.add_argument("--gen-procs", type=int, default=0, help="worker processes generating the synthetic contigs (0: up to 32; 1: none, e.g. under rocprofv3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")

    # torch first: it brings its own copy of the HIP runtime, and the C-ABI library must bind to that one (the same
    # process cannot run two)
    import torch
    from pyrodigal_amd import benchdata, distributed
    models = benchdata.load_model_set()
    model_gcs = [float(np.frombuffer(m[1][:8], np.float64)[0]) for m in models]
    # ---- the job and this rank's share of it; generated by spawned worker processes
    single = args.workload == "config5"
    if args.workload == "config4":
        lengths, gcs, seeds = benchdata.config4_spec(args.contigs)
        wname = "%dx20kbp_gc30-70_meta" % args.contigs
    elif args.workload == "config3":
        c = np.arange(1000)
        lengths, gcs, seeds = np.full(1000, 50_000), 0.30 + 0.40 * (c % 41) / 40, 10_000 + c
        wname = "1000x50kbp_gc30-70_meta"
    elif args.workload == "config2":
        lengths, gcs, seeds = np.array([5_000_000]), np.array([0.50]), np.array([1234])
        wname = "1x5Mbp_gc50_meta"
    else:
        lengths, gcs, seeds = np.array([200_000_000]), np.array([0.65]), np.array([5])
        wname = "1x200Mbp_gc65_single"
    work = distributed.estimate_work_known(lengths, gcs, None if single else model_gcs)
    mine = distributed.pack_contigs(work, world)[rank]        # a single contig does not shard: ranks > 0 idle on configs 2 / 5
    t_gen = time.perf_counter()
    planted = args.workload == "config4" and args.series == "planted"
    if planted:
        wname += "_planted"
    # (worker processes of the generator: the CPUs this process may use -- the container's quota when it has one -- shared by the ranks of
    #  the node, at most 32: eight ranks spawning 32 interpreters each would spend longer starting them than generating)
    gen_procs = args.gen_procs or max(2, min(32, int((_cpu_quota() or os.cpu_count() or 1) * (2 if world == 1 else 1)) // max(1, world)))
    seqs = benchdata.generate(lengths[mine], gcs[mine], seeds[mine], procs=gen_procs, planted=planted)
    t_gen = time.perf_counter() - t_gen
    job_bases = int(np.sum(lengths))

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

    from pyrodigal_amd import _cabi
    sub = max(1, args.sub_batch)
    if args.workload == "config4" and args.contexts > 1 and len(seqs) > 1:
        # at least one device call per context: a rank whose share is a single sub-batch (the job on 8 GPUs) would otherwise
        # have nothing to overlap the host-side phases of its one call with
        # ... but not calls of fewer than 3125 contigs: below that the launches stop filling the chip
        sub = min(sub, max(min(sub, 3125), -(-len(seqs) // args.contexts)))
    groups = [seqs[i:i + sub] for i in range(0, len(seqs), sub)]
    n_ctx = max(1, min(args.contexts, len(groups)))
    # The call plan of a pass: the first calls (one per context) hold 1/4, 1/2, 3/4 and 1 sub-batch, the last ones the same in falling
    # order.  A pass starts with an idle device and every context packing and uploading its first call: with equal calls the first kernel
    # waits for a whole 125 MB upload, the contexts stay in step (their host phases coincide) and the last four calls end together.
    # Measured (round 6): 118.4 -> 112.1 ms per step.  PGA_BENCH_RAMP="f0,f1,..": other fractions; PGA_BENCH_RAMP=0: equal calls.
    ramp = os.environ.get("PGA_BENCH_RAMP", "0.25,0.5,0.75,1")
    call_plan = None
    if ramp not in ("", "0", "none") and args.workload == "config4" and len(groups) > 2 * n_ctx:
        fr = [float(x) for x in ramp.split(",")]
        head = [max(1, int(sub * f)) for f in fr]
        tail = head[::-1]
        mid = len(seqs) - sum(head) - sum(tail)
        if mid >= sub:              # (a share too small for a ramp keeps its equal calls)
            nmid = -(-mid // sub)
            sizes = head + [mid // nmid + (1 if k < mid % nmid else 0) for k in range(nmid)] + tail
            assert sum(sizes) == len(seqs) and min(sizes) > 0
            cuts = np.cumsum([0] + sizes)
            groups = [seqs[cuts[k]:cuts[k + 1]] for k in range(len(sizes))]
            n_ctx = max(1, min(args.contexts, len(groups)))
            call_plan = sizes
    ctxs = [_cabi.Context(dev_index) for _ in range(n_ctx)]
    ctx = ctxs[0]
    if single:
        from tests.util import golden_path
        import gzip
        with gzip.open(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz")) as f:
            blob = f.read()
        for c in ctxs:
            c.set_models([blob])
        kw = dict(meta=False, closed=True)
    else:
        for c in ctxs:
            c.set_models([m[1] for m in models])
        kw = dict(meta=True)
    lanes = Lanes(ctxs)
    base_of = np.cumsum([0] + [len(g) for g in groups])
    mine_arr = np.asarray(mine, np.int32)
    t_gather = [0.0]

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def renumber(r, k):
        """Job-wide contig numbers for the gene records of device call k (in place, in the result's own memory).  Runs on the thread
        that made the call, right behind it: the other contexts' kernels are running meanwhile, where the same loop at the end of
        the step (rounds 2-4: 13 ms per 2 Gbp step for 2.9 M records) had the device wait."""
        g = r.genes
        if len(g) and not getattr(r, "_renumbered", False):
            g["contig"] = mine_arr[base_of[k] + g["contig"]]
        r._renumbered = True
        return r

    def gather(results):
        """This rank's gene records with job-wide contig numbers (each call's records are renumbered as soon as the call returns,
        `renumber`), then the one exchange of the job: a gather to rank 0.  With one rank the records already are where the job
        wants them -- one array per device call, in host memory -- and nothing is copied."""
        t0 = time.perf_counter()
        parts = [renumber(r, k).genes for k, r in enumerate(results)]
        if dist is None:
            t_gather[0] += time.perf_counter() - t0
            return parts
        g = np.concatenate(parts) if len(parts) != 1 else parts[0]
        out = [distributed.gather_genes(g, dist, device=xdev, dst=0)]
        t_gather[0] += time.perf_counter() - t0
        return out

    # Bring the device out of its idle power state before the warmup steps proper: after a pause (the host was busy
    # generating the synthetic contigs) the first ~100 ms of work run at ramping clocks.  Untimed, like the warmup.
    t_pre = time.perf_counter()
    while groups and time.perf_counter() - t_pre < 0.4:
        ctx.find_genes_batch(groups[0], **kw)            # full-size calls: every launch a profiler sees has the size of the timed ones

    # ---- the headline: host to host.  A step = the whole job from ASCII contigs in host memory: per device call
    #      packing into pinned memory, H2D, the path, gene records back in host memory; then the gather.
    h2h_call = lambda c, k: renumber(c.find_genes_batch(groups[k], **kw), k)
    for _ in range(args.warmup):
        gather(lanes.run(len(groups), h2h_call))
    sync()
    t_gather[0] = 0.0
    t0 = time.perf_counter()
    dp_ms_shared, passes_shared, calls_shared = 0.0, 0, 0
    res, all_genes = [], None
    streamed = dist is not None and n_ctx > 1 and os.environ.get("PGA_BENCH_JOIN", "0") != "1"
    if streamed:
        # N > 1: a rank's share is a handful of calls (four for an eighth of the job), and what a join per step costs -- the first upload
        # with an idle device, the last calls running alone, about 14 ms per step on one GPU -- does not shrink with N.  So the K steps of
        # a rank run back to back: a context goes on with its call of step s + 1 as soon as it has made its call of step s, and the
        # gather of step s (in step order, on this thread; the one collective of the job) runs under the kernels of step s + 1.  Still K
        # whole passes over the whole job, K gathers, one barrier on either side.  (PGA_BENCH_JOIN=1: a join per step as with one rank.)
        acc = {"res": [], "genes": None}

        def step_done(res_):
            acc["genes"] = gather(res_)
            acc["res"] = res_
            for r in res_:
                acc["dp"] = acc.get("dp", 0.0) + r.t_dp_ms; acc["np"] = acc.get("np", 0) + r.node_passes; acc["nc"] = acc.get("nc", 0) + 1
        lanes.run_back_to_back(args.steps, len(groups), h2h_call, step_done)
        res, all_genes = acc["res"], acc["genes"]
        dp_ms_shared, passes_shared, calls_shared = acc.get("dp", 0.0), acc.get("np", 0), acc.get("nc", 0)
    else:
        for _ in range(args.steps):
            res = lanes.run(len(groups), h2h_call)
            all_genes = gather(res)
            for r in res:
                dp_ms_shared += r.t_dp_ms; passes_shared += r.node_passes; calls_shared += 1
    t_local = time.perf_counter() - t0            # this rank's own time, before waiting for the others
    sync()
    elapsed = time.perf_counter() - t0
    gather_s = t_gather[0]
    per_rank = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # what makes the first real multi-GPU run self-explaining: every rank's own step time, its share of the gather, and the
        # work the packing gave it
        mine_t = torch.tensor([1e3 * t_local / args.steps, 1e3 * gather_s / args.steps, float(np.sum(work[mine])), float(len(mine))],
                              dtype=torch.float64, device=xdev)
        allt = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(allt, mine_t)
        rows = [[float(x) for x in t_.tolist()] for t_ in allt]
        wsum = [r_[2] for r_ in rows]
        per_rank = {"step_ms": [round(r_[0], 3) for r_ in rows], "gather_ms": [round(r_[1], 3) for r_ in rows],
                    "contigs": [int(r_[3]) for r_ in rows],
                    "estimated_work_share": [round(w_ / max(sum(wsum), 1e-9), 4) for w_ in wsum],
                    "lpt_imbalance": round(max(wsum) / max(sum(wsum) / world, 1e-9), 4)}

    # ---- the same K passes issued back to back (one rank only; reported next to `value`, never as it): a step of the loop above ends
    #      with a join of all contexts, so the pipeline of calls drains and refills once per step; a caller that streams batch after
    #      batch does not pay that.  Every pass still ends with its own gather, in order.
    b2b_elapsed = None
    if dist is None and n_ctx > 1 and len(groups) > 1:
        sync()
        t0b = time.perf_counter()
        lanes.run_back_to_back(args.steps, len(groups), h2h_call, gather)
        sync()
        b2b_elapsed = time.perf_counter() - t0b

    # ---- the same loop with the contigs resident in HBM (no packing, no upload): the rate of the path alone
    batches = [ctxs[k % n_ctx].upload(g) for k, g in enumerate(groups)]
    res_steps = max(1, min(args.steps, 5))
    r_elapsed, _, _, _, _, _ = timed_steps(lanes, batches, res_steps, 1, sync, gather, post=renumber, **kw)
    if dist is not None:
        t = torch.tensor([r_elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        r_elapsed = float(t.item())
    # With several contexts a kernel shares the device with the other contexts' kernels and its own duration says little about
    # the kernel: the connection-scoring roofline is taken from the same calls issued one after the other (same batches, same
    # HIP events on the library's stream), right after the timed region; the overlapped figure is reported next to it.
    dp_ms, passes, calls = 0.0, 0, 0
    topo_ms = sched_ms = 0.0
    # (with a ramped call plan: the FULL-SIZE calls only -- the launch the PMC passes profile and the earlier rounds report)
    full_size = max(len(g) for g in groups)
    for _ in range(max(1, min(args.steps, 3))):
        for k, b in enumerate(batches):
            if len(groups[k]) != full_size:
                continue
            r = ctxs[k % n_ctx].find_genes(b, **kw)
            dp_ms += r.t_dp_ms; passes += r.node_passes; calls += 1
            tmg = ctxs[k % n_ctx].dp_timings()
            topo_ms += tmg["topo_ms"]; sched_ms += tmg["sched_ms"]
    sync()

    out = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = job_bases * args.steps / elapsed / 1e6
        n_chains = max([r.n_chains for r in res], default=0)
        out = {
            "metric": "Mbp/sec gene-called (meta mode)" if not single else "Mbp/sec gene-called (single mode)",
            "value": round(value, 3), "unit": "Mbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wname, "contigs": int(len(lengths)), "bases": job_bases, "models": 1 if single else len(models),
                       "contigs_rank0": len(seqs), "device_calls_per_step_rank0": len(groups), "sub_batch_contigs": sub, "contigs_per_call": call_plan,
                       "contexts_per_gpu": n_ctx,
                       "node_passes_per_step_rank0": int(passes_shared // max(args.steps, 1)),
                       "genes_all_ranks": int(sum(len(g) for g in all_genes)) if all_genes is not None else 0,
                       "parallelism": "contigs packed by estimated work over %d GPU(s), one gather of gene records to rank 0" % world,
                       "steps_issued": "back to back (a rank's contexts go on with step s + 1 while step s is gathered)" if streamed else "a join of the contexts and a gather per step",
                       "timed": "host to host (SURVEY 8d): ASCII contigs in host memory -> pinned packing -> H2D -> path -> gene records in host memory -> gather",
                       "resident_Mbp_s": round(job_bases * res_steps / r_elapsed / 1e6, 3), "resident_steps": res_steps,
                       "resident_ms_per_step": round(1e3 * r_elapsed / res_steps, 3),
                       "gather_ms_per_step_rank0": round(1e3 * gather_s / args.steps, 3),
                       "host_to_host_back_to_back_Mbp_s": round(job_bases * args.steps / b2b_elapsed / 1e6, 3) if b2b_elapsed else None,
                       "generate_s_rank0": round(t_gen, 2)},
            # one launch = one device call = one sub-batch: the PMC passes profile exactly that (tools/collect_profiles.sh)
            "roofline": roofline(ctx, dp_ms, passes, calls, n_chains, wname,
                                 "%dx20kbp_gc30-70_meta" % min(sub, len(seqs)) if args.workload == "config4" else None,
                                 aux_ms=(topo_ms, sched_ms) if not single and args.workload in ("config4", "config3") else None),
        }
        if per_rank is not None:
            out["config"]["per_rank"] = per_rank
        pl = pipeline_entry("%dx20kbp_gc30-70_meta" % min(sub, len(seqs)) if args.workload == "config4" else wname)
        if pl is not None:
            out["pipeline_hbm_bytes_per_bp"] = pl["pipeline_hbm_bytes_per_bp"]         # all kernels of a device call
            out["pipeline"] = pl
        if True:
            out["roofline"]["measured"] = "the full-size calls (%d contigs) issued one after the other right after the timed region (kernel alone on the device)" % full_size
            out["roofline"]["kernel_ms_per_launch_in_timed_region"] = round(dp_ms_shared / max(calls_shared, 1), 4)
            out["roofline"]["frac_in_timed_region"] = round(BYTES_PER_NODE_PASS * passes_shared / (dp_ms_shared * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if dp_ms_shared > 0 else 0.0
        if world == 1 and not args.no_cpu_baseline and not single:
            out["cpu_baseline"] = cpu_baseline(seqs, models, res[0] if res else None, args.contigs if args.workload == "config4" else 0)
            if len(seqs) > 1:
                # (i) one thread above, (ii) every host core here -- its own top-level object
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(seqs, models)
                # ... and its last row: the job that was timed, checked across every device call and context
                out["parity"] = parity_sample(seqs, models, res, base_of, n_ctx)
    share_line = None
    if rank == 0 and world == 1 and args.workload == "config4" and not args.no_secondary and len(seqs) >= 8 * 3125:
        # What one rank of the 8-GPU run will do, timed alone here: rank 0's share of the LPT packing (12 500 of the 100 000 contigs) with the
        # call plan bench.py gives a rank (calls of at least 3 125 contigs on up to --contexts contexts), host to host, a join and a gather
        # per step.  predicted_efficiency_at_8 = (this GPU's whole-job step / 8) / the share's step: what per-step fixed costs (first
        # upload, drain and refill of the call pipeline) leave of linear scaling BEFORE the gather over xGMI.  No scaling curve: a prediction.
        share = distributed.pack_contigs(work, 8)[0]
        where = {int(c): i for i, c in enumerate(mine)}
        sseqs = [seqs[where[int(c)]] for c in share]
        ssub = min(args.sub_batch, max(min(args.sub_batch, 3125), -(-len(sseqs) // args.contexts)))
        sgroups = [sseqs[i:i + ssub] for i in range(0, len(sseqs), ssub)]
        slanes = Lanes(ctxs[:max(1, min(n_ctx, len(sgroups)))])
        scall = lambda c, k: c.find_genes_batch(sgroups[k], **kw)
        for _ in range(2):
            slanes.run(len(sgroups), scall)
        sync()
        ssteps = max(3, min(args.steps, 10))
        t0s = time.perf_counter()
        for _ in range(ssteps):
            [r.genes for r in slanes.run(len(sgroups), scall)]
        sync()
        share_ms = 1e3 * (time.perf_counter() - t0s) / ssteps
        t0s = time.perf_counter()
        slanes.run_back_to_back(ssteps, len(sgroups), scall, lambda res_: None)
        sync()
        share_b2b_ms = 1e3 * (time.perf_counter() - t0s) / ssteps
        slanes.close()
        whole_ms = 1e3 * elapsed / args.steps
        share_line = {"contigs": len(sseqs), "bases": int(sum(len(x) for x in sseqs)), "device_calls_per_step": len(sgroups), "sub_batch_contigs": ssub,
                      "contexts": len(slanes.ctxs), "steps": ssteps, "ms_per_step": round(share_ms, 3),
                      "ms_per_step_back_to_back": round(share_b2b_ms, 3),
                      "whole_job_ms_per_step_one_gpu": round(whole_ms, 3),
                      "predicted_efficiency_at_8": round((whole_ms / 8.0) / share_ms, 4),
                      "predicted_efficiency_at_8_back_to_back": round((whole_ms / 8.0) / share_b2b_ms, 4),
                      "what": "rank 0's share of the job on 8 GPUs (LPT packing), timed alone on this GPU with a rank's own call plan; a prediction of "
                              "per-step fixed costs, not a scaling measurement: no multi-GPU run exists"}
    fasta_line = pool_line = None
    try:
        free_b, total_b = torch.cuda.mem_get_info(dev_index)
        out["config"]["hbm_in_use_GB"] = round((total_b - free_b) / 1e9, 1)        # the contexts' buffers + the resident batches, at their peak
    except Exception:
        pass
    # (the job's contexts hold tens of GB -- 160 GB in round 3 --; the two workloads below add about 30 GB while they run.  Closing the job's
    # contexts first makes the FASTA pass that creates its contexts inside the timed region wait for 160 GB of frees.)
    if rank == 0 and world == 1 and not single and not args.no_secondary:
        fasta_line = fasta_to_genes(seqs[:min(len(seqs), 20000)], models, dev_index, kw)
        pool_line = threadpool_find_genes(seqs[:min(len(seqs), 8000)], models, dev_index)
    for b in batches:
        b.close()
    lanes.close()
    for c in ctxs[1:]:
        c.close()
    if rank == 0 and world == 1 and not args.no_secondary:
        out["secondary"] = secondary(ctx, _cabi, benchdata, models, args.workload, sync)
        if fasta_line:
            out["secondary"]["fasta_file_to_genes"] = fasta_line
        if pool_line:
            out["secondary"]["threadpool_find_genes"] = pool_line
        if share_line:
            out["secondary"]["rank_share_of_8"] = share_line
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def secondary(ctx, _cabi, benchdata, models, headline, sync):
    """The other single-GPU configurations of BASELINE.json, a few steps each, inputs resident in HBM."""
    import gzip
    from tests.util import golden_path
    out = {}
    def planted4():
        lengths, gcs, seeds = benchdata.config4_spec(6250)
        return benchdata.generate(lengths, gcs, seeds, planted=True)

    # config4_planted: one device call of the headline job's size on the planted-ORF series (about 0.06 nodes per base: real density)
    def ragged4():
        # one device call's worth of contigs with ragged lengths (5 .. 60 kbp, seeded): no two tiles end alike, the extraction's staging
        # slack and the per-contig tails are timed on something other than 6 250 equal contigs
        rng = np.random.default_rng(20260601)
        n = 3800
        lengths = rng.integers(5_000, 60_001, n)
        c = np.arange(n)
        return benchdata.generate(lengths, 0.30 + 0.40 * (c % 41) / 40, 3_000_000 + c)

    plans = [("config4_planted", "6250x20kbp_gc30-70_meta_planted", planted4, dict(meta=True), 10),
             ("config4_ragged", "3800x5-60kbp_gc30-70_meta", ragged4, dict(meta=True), 10),
             ("config2", "1x5Mbp_gc50_meta", lambda: benchdata.config2(0), dict(meta=True), 10),
             ("config3", "1000x50kbp_gc30-70_meta", lambda: benchdata.generate(*_config3_spec()), dict(meta=True), 10),
             ("config5", "1x200Mbp_gc65_single", benchdata.config5, dict(meta=False, closed=True), 3)]
    for key, wname, make, kw, steps in plans:
        if key == headline:
            continue
        seqs = make()
        if key == "config5":
            with gzip.open(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz")) as f:
                ctx.set_models([f.read()])
        else:
            ctx.set_models([m[1] for m in models])
        b = ctx.upload(seqs)
        elapsed, dp_ms, passes, calls, res, _ = timed_steps(Lanes([ctx]), [b], steps, 2, sync, lambda r: None, **kw)
        bases = sum(len(s) for s in seqs)
        t1 = time.perf_counter()
        ctx.find_genes_batch(seqs, **kw)
        h2h = time.perf_counter() - t1
        out[key] = {"workload": wname, "value": round(bases * steps / elapsed / 1e6, 3), "unit": "Mbp/s", "steps": steps,
                    "ms_per_step": round(1e3 * elapsed / steps, 3), "genes": int(len(res[0].genes)),
                    "host_to_host_Mbp_s": round(bases / h2h / 1e6, 3),
                    "roofline": roofline(ctx, dp_ms, passes, calls, res[0].n_chains, wname)}
        if key == "config4_planted":
            out[key]["nodes_per_bp"] = round(float(np.sum(res[0].contigs["n_nodes"])) / bases, 4)
            out[key]["node_passes_per_call"] = int(passes // max(calls, 1))
        b.close()
    try:
        out.update(reference_benchmarks(ctx, models))
    except Exception as err:          # (never lose the line to a secondary figure)
        out["reference_benchmarks_error"] = repr(err)
    return out


def reference_benchmarks(ctx, models):
    """The two series the reference itself publishes, on the one genome of its test data that is here
    (GCF_001457455.1, 2.46 Mbp, 153 296 nodes): (i) benches/connection_scoring/bench.py:38-83 -- ConnectionScorer.index +
    score_connections(final=True) over a genome's sorted, scored nodes, one pass; (ii) benches/run_single/bench.py:37-104 --
    GeneFinder().train(genome) + find_genes(genome).  The published figures are a laptop CPU's (i7-10710U, one thread): context, not target."""
    import gzip
    from oracle import oracle as orc
    from tests.util import golden_path, read_fasta
    from pyrodigal_amd import lib
    seq = read_fasta("GCF_001457455.1_NCTC11397_genomic.fna.gz")[0][1]
    seq_b = seq.encode() if isinstance(seq, str) else seq
    tinf = orc.Training.load(golden_path("GCF_001457455.1_NCTC11397_genomic.tinf_closed.bin.gz"))
    o = orc.Oracle(seq_b)
    o.extract(tinf.trans_table, orc.Params(closed=True)); o.sort(); o.reset_scores()
    o.score_nodes(tinf, True, False)
    o.overlapping_starts(tinf, 1, 60)
    nd = o.nodes()
    n = len(nd)
    args_ = (nd["ndx"], nd["stop_val"], nd["type"], nd["strand"], nd["cscore"], nd["sscore"], nd["rscore"], nd["uscore"], nd["star_ptr"], tinf.st_wt, True)
    ctx.score_connections(*args_)
    h2h, kern = [], []
    for _ in range(10):
        t0 = time.perf_counter()
        score, traceb, ov, mi, ms = ctx.score_connections(*args_)
        h2h.append(time.perf_counter() - t0); kern.append(ms)
    t0 = time.perf_counter()
    o.dprog_raw(tinf, True)
    t_cpu = time.perf_counter() - t0
    ref = o.nodes()
    same = bool(np.array_equal(traceb, ref["traceb"]) and np.array_equal(score.view(np.uint64), ref["score"].view(np.uint64)))
    h2h.sort(); kern.sort()
    med, kmed = h2h[len(h2h) // 2], kern[len(kern) // 2] * 1e-3
    bases = len(seq_b)
    out = {"ref_connection_scoring": {
        "genome": "GCF_001457455.1", "bases": bases, "nodes": int(n), "passes": 1,
        "host_to_host_ms": round(1e3 * med, 3), "value": round(bases / med / 1e6, 2), "unit": "Mbp/s", "M_nodes_per_s": round(n / med / 1e6, 2),
        "kernels_ms": round(1e3 * kmed, 3), "kernels_Mbp_s": round(bases / kmed / 1e6, 1), "kernels_M_nodes_per_s": round(n / kmed / 1e6, 1),
        "identical_to_oracle": same,
        "cpu_oracle_one_core": {"ms": round(1e3 * t_cpu, 1), "Mbp_s": round(bases / t_cpu / 1e6, 2), "M_nodes_per_s": round(n / t_cpu / 1e6, 3)},
        "published": {"Mbp_s": 19.6, "M_nodes_per_s": 1.20, "what": "pyrodigal v3.0.0, AVX2 pre-filter, i7-10710U, one thread, 50 genomes "
                      "(benches/connection_scoring/v3.0.0.json): published on a laptop CPU -- context, not target"},
        "what": "pga_score_connections (= ConnectionScorer.index + score_connections(final=True)) over the genome's sorted, scored nodes in host "
                "memory: upload, topology, connection scoring, results back; median of 10"}}
    # (ii) train + find_genes through the drop-in host layer
    ts = []
    genes = 0
    for rep_ in range(4):
        t0 = time.perf_counter()
        finder = lib.GeneFinder(closed=True)
        finder.train(seq_b)
        g = finder.find_genes(seq_b)
        ts.append(time.perf_counter() - t0)
        genes = len(g)
        del finder
    ts = sorted(ts[1:])
    t_run = ts[len(ts) // 2]
    out["ref_run_single"] = {
        "genome": "GCF_001457455.1", "bases": bases, "genes": int(genes), "seconds": round(t_run, 4), "value": round(bases / t_run / 1e6, 2), "unit": "Mbp/s",
        "first_run_seconds": None,
        "published": {"Mbp_s": 2.16, "what": "pyrodigal v3.7.0, SSE2, i7-10710U, one thread, 50 genomes (benches/run_single/v3.7.0.json): published "
                      "on a laptop CPU -- context, not target"},
        "what": "pyrodigal_amd.lib: GeneFinder(closed=True).train(genome) + find_genes(genome), host to host, a new GeneFinder per run; median of "
                "the runs after the first (BASELINE.json configs[0])"}
    return out


def fasta_to_genes(seqs, models, dev_index, kw):
    """A FASTA file of (part of) the workload on local disk -> genes in host memory, through the library's reader (pinned
    staging arenas filled in turn, one DMA per batch, two contexts): the rate of the whole ingest path, parser included."""
    import tempfile
    from pyrodigal_amd import pipeline
    with tempfile.NamedTemporaryFile("wb", suffix=".fna", delete=False) as f:
        path = f.name
        for i, s in enumerate(seqs):
            f.write(b">contig_%d synthetic\n" % i)
            for k in range(0, len(s), 80):
                f.write(s[k:k + 80]); f.write(b"\n")
    try:
        bases = sum(len(s) for s in seqs)
        from pyrodigal_amd import _cabi
        blobs = [m[1] for m in models]
        rates = []
        for rep in range(2):            # cold: contexts, device buffers and pinned arenas are created inside the timed region
            t0 = time.perf_counter()
            genes = 0
            for ids, descs, lens, res in pipeline.find_genes_fasta(path, blobs, n_contexts=2, device=dev_index, max_bases=64 << 20, **kw):
                genes += len(res.genes)
            rates.append(bases / (time.perf_counter() - t0) / 1e6)
        # warm: a caller that processes file after file keeps its contexts (models loaded, buffers grown); the reader and its pinned
        # arenas are still per file
        ctxs = [_cabi.Context(dev_index) for _ in range(3)]
        for c in ctxs:
            c.set_models(blobs)
        warm = []
        for rep in range(3):
            t0 = time.perf_counter()
            for ids, descs, lens, res in pipeline.find_genes_fasta(path, blobs, device=dev_index, max_bases=64 << 20, contexts=ctxs, **kw):
                pass
            warm.append(bases / (time.perf_counter() - t0) / 1e6)
        for c in ctxs:
            c.close()
        return {"value": round(max(warm[1:]), 3), "unit": "Mbp/s", "cold_first_pass": round(rates[0], 3), "cold_second_pass": round(rates[1], 3),
                "warm_passes": [round(w, 3) for w in warm], "bases": bases, "records": len(seqs), "genes": int(genes),
                "what": "plain FASTA on local disk -> C reader (mapped file, parsed by several threads into pinned arenas) -> DMA -> path -> genes in "
                        "host memory.  value: three contexts kept across files (best of the later passes); cold: two contexts, their buffers and the "
                        "pinned arenas created inside the timed region"}
    finally:
        os.unlink(path)


def threadpool_find_genes(seqs, models, dev_index, threads=32):
    """The reference's own calling pattern (cli.py:289-302): a ThreadPool mapping `GeneFinder.find_genes` over the records, one
    contig per call, through the drop-in host layer (pyrodigal_amd.lib).  Concurrent calls are packed into shared device calls
    by the finder (context pool + request coalescer); every call still builds its own `Genes` under the GIL."""
    from concurrent.futures import ThreadPoolExecutor
    from pyrodigal_amd import lib
    bins = lib.MetagenomicBins([lib.MetagenomicBin(lib.TrainingInfo(raw=b), n) for n, b in models])
    bases = sum(len(s) for s in seqs)
    out = {"threads": threads, "contigs": len(seqs), "bases": bases, "unit": "Mbp/s",
           "what": "ThreadPoolExecutor(%d).map(finder.find_genes, contigs) on ONE GeneFinder(meta=True); host to host, one contig per call" % threads}
    for key, keep, nthreads in (("keep_nodes_false", False, threads), ("default", True, threads), ("keep_nodes_false_128_threads", False, 128)):
        finder = lib.GeneFinder(meta=True, metagenomic_bins=bins, keep_nodes=keep, device=dev_index)
        with ThreadPoolExecutor(nthreads) as ex:
            list(ex.map(finder.find_genes, seqs[:256]))                  # contexts, models, buffers
            finder.stats.update(device_calls=0, sequences=0, max_calls_per_device_call=0)
            t0 = time.perf_counter()
            genes = sum(len(g) for g in ex.map(finder.find_genes, seqs))
            dt = time.perf_counter() - t0
        st = finder.stats
        out[key] = {"value": round(bases / dt / 1e6, 3), "contigs_per_s": round(len(seqs) / dt, 1), "genes": int(genes), "threads": nthreads,
                    "device_calls": st["device_calls"], "contigs_per_device_call": round(st["sequences"] / max(st["device_calls"], 1), 1),
                    "most_calls_in_one_device_call": st["max_calls_per_device_call"]}
        if keep and nthreads == threads:
            # the lone call: one 20 kbp contig, nobody to share a device call with
            lat = []
            for s in seqs[:30]:
                t1 = time.perf_counter()
                finder.find_genes(s)
                lat.append(time.perf_counter() - t1)
            lat.sort()
            out["lone_call_ms"] = {"median": round(1e3 * lat[len(lat) // 2], 3), "min": round(1e3 * lat[0], 3), "contig_bp": len(seqs[0])}
        del finder
    # the reference's harness itself: multiprocessing.pool.ThreadPool(jobs).map(process, records) with map's own chunking
    # (cli.py:289-302; a chunk of len / (4 jobs) records per task, so a thread makes its calls back to back)
    from multiprocessing.pool import ThreadPool
    for nthreads in (32, 128):
        finder = lib.GeneFinder(meta=True, metagenomic_bins=bins, keep_nodes=False, device=dev_index)
        with ThreadPool(nthreads) as pool:
            pool.map(finder.find_genes, seqs[:256])
            finder.stats.update(device_calls=0, sequences=0, max_calls_per_device_call=0)
            t0 = time.perf_counter()
            genes = sum(len(g) for g in pool.map(finder.find_genes, seqs))
            dt = time.perf_counter() - t0
        st = finder.stats
        out["threadpool_map_%d_threads" % nthreads] = {
            "value": round(bases / dt / 1e6, 3), "contigs_per_s": round(len(seqs) / dt, 1), "genes": int(genes), "threads": nthreads,
            "device_calls": st["device_calls"], "contigs_per_device_call": round(st["sequences"] / max(st["device_calls"], 1), 1),
            "what": "multiprocessing.pool.ThreadPool(%d).map(finder.find_genes, contigs), keep_nodes=False" % nthreads}
        del finder
    out["value"] = out["keep_nodes_false"]["value"]
    return out


def _config3_spec():
    c = np.arange(1000)
    return np.full(1000, 50_000), 0.30 + 0.40 * (c % 41) / 40, 10_000 + c


def cpu_baseline(seqs, models, gpu_res, n_job=0):
    """The CPU oracle (a C port of pyrodigal's CPU path: byte pre-filter + split scorers) on a bounded sample of the same
    workload: one thread (also the gene-call parity check of this run), then one process per host core (cpu_baseline_all_cores)."""
    from oracle import oracle as orc
    bins = [orc.Training(m[1]) for m in models]
    budget_bases = 12_000_000 if len(seqs) > 1 else 5_000_000
    done, t_cpu, match, total_genes, i = 0, 0.0, True, 0, -1
    for i, s in enumerate(seqs):
        if done >= budget_bases:
            break
        o = orc.Oracle(s)
        t0 = time.perf_counter()
        phase = o.find_genes_meta(bins)
        t_cpu += time.perf_counter() - t0
        done += len(s)
        og = o.genes()
        total_genes += len(og)
        if gpu_res is not None and i < len(gpu_res.contigs):
            gg = gpu_res.genes_of(i)
            ok = gpu_res.contigs[i]["model"] == phase and len(og) == len(gg) and all(
                np.array_equal(og[k], gg[k]) for k in ("begin", "end", "start_ndx", "stop_ndx"))
            match = match and bool(ok)
    out = {"value": round(done / t_cpu / 1e6, 3), "unit": "Mbp/s", "cores": 1, "kind": "port",
           "sample": "%d contig(s), %d bp, same 16 models, meta mode, 1 thread (%s)" % (i + (done >= budget_bases), done, _cpu_name()),
           "gene_calls_identical_to_gpu": match, "genes_in_sample": total_genes, "host_cpus": os.cpu_count()}
    out["avx2"] = True      # oracle/Makefile: -O2 -mavx2 -ftree-vectorize; gcc vectorises the byte pre-filter loop (32-byte vectors)
    return out


def parity_sample(seqs, models, res, base_of, n_ctx, every=10, threads=16):
    """SURVEY 8(d) "parity checks in the bench run": a stratified sample of the job that was just timed -- every `every`-th contig,
    hence contigs of EVERY device call and every context -- through the CPU checker (oracle/, one contig per call on a few Python
    threads; the C side runs without the interpreter lock), compared with the gene records the timed passes produced: the gene tuples
    (begin, end, strand, start and stop node, partial flags, start type), the chosen bin per contig, and the largest absolute
    difference over the five score fields of the genes' start nodes (0.0 = bit-identical)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    bins = [orc.Training(m[1]) for m in models]
    picks = list(range(0, len(seqs), every))

    score_fields = ("cscore", "sscore", "rscore", "uscore", "tscore")
    int_fields = ("begin", "end", "strand", "start_ndx", "stop_ndx", "partial_begin", "partial_end", "start_type")

    def one(i):
        # the checker's gene list, with each gene's attributes read off its start / stop nodes as Gene's properties read them
        # (ref: lib.pyx:2644-2830): strand, partial flags = the edge flags of the two nodes, start type (3 = Edge), the start node's scores
        o = orc.Oracle(seqs[i])
        phase = o.find_genes_meta(bins)
        og, on = o.genes(), o.nodes()
        rec = {k: og[k].copy() for k in ("begin", "end", "start_ndx", "stop_ndx")}
        if len(og):
            s_, e_ = on[og["start_ndx"]], on[og["stop_ndx"]]
            fwd = s_["strand"] == 1
            rec["strand"] = s_["strand"].astype(np.int64)
            rec["partial_begin"] = np.where(fwd, s_["edge"], e_["edge"]).astype(np.int64)
            rec["partial_end"] = np.where(fwd, e_["edge"], s_["edge"]).astype(np.int64)
            rec["start_type"] = np.where(s_["edge"] != 0, 3, s_["type"]).astype(np.int64)
            for f in score_fields:
                rec[f] = s_[f].copy()
        return i, phase, rec

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        got = list(ex.map(one, picks))
    calls, n_genes, bad_tuples, bad_models, max_diff = set(), 0, 0, 0, 0.0
    for i, phase, rec in got:
        k = int(np.searchsorted(base_of, i, side="right") - 1)
        calls.add(k)
        r = res[k]
        li = i - int(base_of[k])
        gg = r.genes_of(li)
        n = len(rec["begin"])
        n_genes += n
        if int(r.contigs[li]["model"]) != phase:
            bad_models += 1
        if n != len(gg) or (n and not all(np.array_equal(rec[f], gg[f].astype(np.int64)) for f in int_fields)):
            bad_tuples += 1
            continue
        for f in score_fields:
            if n:
                max_diff = max(max_diff, float(np.max(np.abs(rec[f] - gg[f]))))
    tuple_fields = int_fields
    return {"contigs": len(picks), "every": every, "device_calls_covered": len(calls), "contexts_covered": len({k % n_ctx for k in calls}),
            "genes": n_genes, "tuples_identical": bad_tuples == 0, "chosen_bin_identical": bad_models == 0,
            "contigs_with_different_tuples": bad_tuples, "contigs_with_different_bin": bad_models,
            "tuple_fields": list(tuple_fields), "score_fields": list(score_fields), "max_abs_score_diff": max_diff,
            "checker": "oracle/ (CPU restatement, pinned on the reference's fixtures)", "seconds": round(time.perf_counter() - t0, 2)}


def cpu_baseline_all_cores(seqs, models):
    """SURVEY 8(d)'s line (ii): the same CPU checker on the host's cores -- a pool of C threads sharing the models, one contig per call
    (pyrodigal's own pool model, ref: cli.py:289-302, without an interpreter lock in the way; oracle/prodigal_oracle.c
    `po_find_genes_meta_pool_pinned`) -- as a curve over the thread count: threads pinned one per physical core, spread over the
    sockets, hardware siblings only once the cores are used up; every point runs for about five seconds (the sample's contigs gone
    over as often as that takes).  `cpu_busy` = CPU time the threads got / (threads x wall time): below 1 the threads were not
    running (a CPU quota of the container, or waiting), at 1 with a falling rate per thread they were running slower (shared caches,
    memory, clocks).  `value` is the best point."""
    from oracle import oracle as orc
    bins = [orc.Training(m[1]) for m in models]
    order, physical = _pin_order()
    logical = len(order)
    out = {"unit": "Mbp/s", "kind": "port", "physical_cores": physical, "logical_cpus": logical, "os_cpu_count": os.cpu_count(),
           "cpu_quota_cores": _cpu_quota(), "cpu": _cpu_name(), "avx2": True, "pinned": True, "curve": []}
    sample = seqs[:min(len(seqs), 4096)]
    mean_len = sum(len(s) for s in sample) / max(len(sample), 1)
    orc.find_genes_meta_pool_pinned(sample[:64], bins, min(8, logical), cpus=order)          # page cache of the tables, the library
    quota = out["cpu_quota_cores"]
    points = {1, 8, 32, 64, 128, physical, logical} | ({max(1, int(quota))} if quota else set())
    t0 = time.perf_counter()
    orc.find_genes_meta_pool_pinned(sample[:24], bins, 1, cpus=order)
    one_thread = 24 / (time.perf_counter() - t0)             # contigs per second of one thread
    for threads in sorted(t for t in points if t <= logical):
        running = min(threads, quota) if quota else threads  # threads that can be on a CPU at once
        total = int(max(threads * 8, one_thread * running * 5.5))
        t0 = time.perf_counter()
        genes, cpu_s = orc.find_genes_meta_pool_pinned(sample, bins, threads, total=total, cpus=order)
        dt = time.perf_counter() - t0
        calls = total
        out["curve"].append({"threads": threads, "value": round(calls * mean_len / dt / 1e6, 3), "seconds": round(dt, 2), "calls": calls,
                             "Mbp_s_per_thread": round(calls * mean_len / dt / 1e6 / threads, 3), "cpu_busy": round(cpu_s / (dt * threads), 3),
                             "genes": genes})
    best = max(out["curve"], key=lambda r: r["value"])
    one = out["curve"][0]
    out.update(value=best["value"], cores=best["threads"], scaling_vs_one_thread=round(best["value"] / max(one["value"], 1e-9), 1),
               sample="%d contigs of the job (%.0f bp on average) gone over for about 5 s per point, one contig per call on a pool of pinned C threads "
                      "sharing the 16 models; built -O2 -mavx2 (the byte pre-filter of the connection scoring is auto-vectorised with 32-byte "
                      "vectors, like the reference's AVX2 backend)" % (len(sample), mean_len))
    return out


def _pin_order():
    """Logical CPUs this process may use, ordered for pinning: one per physical core first, alternating between the sockets, then the
    hardware siblings in the same order.  Returns (order, number of physical cores among them)."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cores = {}
    for c in allowed:
        try:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            key = (int(open(base + "physical_package_id").read()), int(open(base + "core_id").read()))
        except (OSError, ValueError):
            key = (0, c)
        cores.setdefault(key, []).append(c)
    by_pkg = {}
    for key in sorted(cores):
        by_pkg.setdefault(key[0], []).append(cores[key])
    firsts, rest = [], []
    lists = [by_pkg[k] for k in sorted(by_pkg)]
    for i in range(max(len(l) for l in lists)):
        for l in lists:
            if i < len(l):
                firsts.append(l[i][0]); rest.extend(l[i][1:])
    return firsts + rest, len(firsts)


def _cpu_quota():
    """CPUs the container's CPU controller lets this process use at once (cgroup v2 cpu.max or v1 cfs quota), None when unlimited."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(p), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / p, 2)
    except (OSError, ValueError):
        return None


def _physical_cores():
    try:
        cores = set()
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
        return len(cores) or None
    except OSError:
        return None


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
