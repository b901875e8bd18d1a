"""bench.py's N > 1 branch, launched the way the driver launches it (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py
--gpus 2`), as two ranks that share cuda:0 -- the test box has one GPU, so the rendezvous and the collectives run on gloo
(PGA_BENCH_BACKEND=gloo; two RCCL ranks cannot share a device).  Everything else is the code an 8-GPU node runs: the LPT packing of
the job over the ranks, each rank's contexts sized to its share, the gather of gene records to rank 0, the MAX over ranks of the step
time, the per-rank diagnostics.  The JSON line must describe the same job as a one-rank run and hold the same number of genes."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--contigs", "4000", "--steps", "2", "--warmup", "1", "--no-secondary", "--no-cpu-baseline", "--gen-procs", "1"]


def _line(cmd, env):
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]            # rank 0 prints the one line of the job
    return json.loads(lines[0])


def test_two_ranks_on_one_device_report_the_same_job():
    env = dict(os.environ, PGA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    two = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", str(port), "bench.py", "--gpus", "2"] + ARGS, env)
    one = _line([sys.executable, "bench.py", "--gpus", "1"] + ARGS, env)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["scaling"] == "strong" and two["metric"] == one["metric"] and two["unit"] == one["unit"]
    assert two["config"]["workload"] == one["config"]["workload"] and two["config"]["bases"] == one["config"]["bases"] == 4000 * 20000
    pr = two["config"]["per_rank"]
    assert len(pr["step_ms"]) == len(pr["gather_ms"]) == len(pr["contigs"]) == len(pr["estimated_work_share"]) == 2
    assert sum(pr["contigs"]) == 4000 and min(pr["contigs"]) > 1500              # packed by estimated work: about half each
    assert abs(sum(pr["estimated_work_share"]) - 1.0) < 1e-3 and pr["lpt_imbalance"] < 1.02
    assert "per_rank" not in one["config"]
    # the job's answer does not depend on how it was split
    assert two["config"]["genes_all_ranks"] == one["config"]["genes_all_ranks"] > 0
    # a rank's contexts follow its share: 2000 contigs are one device call, so one context
    assert two["config"]["contexts_per_gpu"] == 1 and two["config"]["contigs_rank0"] == pr["contigs"][0]
    assert two["value"] > 0 and two["ms_per_step"] > 0 and two["roofline"]["frac"] > 0


def test_eight_ranks_on_one_device_split_a_job_like_an_eight_gpu_node():
    """The shape of the driver's 8-GPU run (verdict r4, item 9): eight ranks, a 16 000-contig job, every rank its 2 000-contig share
    as one device call on one context; the packing within 2 % of even; the job's genes as a one-rank run finds them.  (Eight ranks
    sharing one GPU say nothing about scaling: no scaling number exists)"""
    env = dict(os.environ, PGA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--contigs", "16000", "--steps", "1", "--warmup", "1", "--no-secondary", "--no-cpu-baseline", "--gen-procs", "1"]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    eight = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                   "--master-port", str(port), "bench.py", "--gpus", "8"] + args, env)
    one = _line([sys.executable, "bench.py", "--gpus", "1"] + args, env)
    pr = eight["config"]["per_rank"]
    assert eight["n_gpus"] == 8 and all(len(pr[k]) == 8 for k in ("step_ms", "gather_ms", "contigs", "estimated_work_share"))
    assert sum(pr["contigs"]) == 16000 and min(pr["contigs"]) > 1700 and pr["lpt_imbalance"] < 1.02
    assert eight["config"]["contexts_per_gpu"] == 1 and eight["config"]["device_calls_per_step_rank0"] == 1
    assert eight["config"]["genes_all_ranks"] == one["config"]["genes_all_ranks"] > 0
    assert eight["config"]["bases"] == one["config"]["bases"] == 16000 * 20000


def test_two_ranks_with_several_contexts_run_their_steps_back_to_back():
    """A rank whose share is several device calls (three calls on three contexts here, four on four for an eighth of the headline job)
    runs its K steps back to back, every step's gather under the next step's kernels (bench.py, `streamed`): the same job, the same
    genes, K gathers in step order."""
    env = dict(os.environ, PGA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--contigs", "16000", "--steps", "3", "--warmup", "1", "--no-secondary", "--no-cpu-baseline", "--gen-procs", "1"]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    two = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", str(port), "bench.py", "--gpus", "2"] + args, env)
    one = _line([sys.executable, "bench.py", "--gpus", "1"] + args, env)
    assert two["n_gpus"] == 2 and two["config"]["contexts_per_gpu"] >= 2 and two["config"]["device_calls_per_step_rank0"] >= 2
    assert two["config"]["steps_issued"].startswith("back to back") and one["config"]["steps_issued"].startswith("a join")
    assert two["config"]["genes_all_ranks"] == one["config"]["genes_all_ranks"] > 0
    assert two["config"]["node_passes_per_step_rank0"] > 0 and two["roofline"]["frac"] > 0 and two["roofline"]["frac_incl_schedule"] > 0
    joined = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), "bench.py", "--gpus", "2"] + args, dict(env, PGA_BENCH_JOIN="1"))
    assert joined["config"]["steps_issued"].startswith("a join") and joined["config"]["genes_all_ranks"] == one["config"]["genes_all_ranks"]
