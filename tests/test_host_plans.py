"""Host arithmetic behind two launches, through the C-ABI helpers that expose it (no GPU needed):
`pga_dp_start_order` -- the wave-batch connection scorer starts its chains longest first (a launch ends when its last chain
does) -- and `pga_cs_task_summary` -- how the ORF walks of the coding score are cut into tasks for the kernel
that keeps the hexamer tables in LDS.  The reference has no counterpart (its loops are serial); what is pinned
here is what the kernels rely on: a permutation, every node in exactly one task, no task larger than one round."""
import numpy as np
import pytest

from pyrodigal_amd import _cabi


def test_start_order_is_a_permutation_longest_first_and_stable():
    rng = np.random.default_rng(5)
    n = rng.integers(0, 3000, 5000).tolist()
    order = _cabi.dp_start_order(n)
    assert sorted(order) == list(range(len(n)))
    batches = [n[k] >> 6 for k in order]
    assert all(a >= b for a, b in zip(batches, batches[1:]))                    # walk batches of 64 nodes, most first
    for a, b in zip(order, order[1:]):
        if n[a] >> 6 == n[b] >> 6:
            assert a < b                                                        # equals keep the launch order
    assert _cabi.dp_start_order([]) == [] and _cabi.dp_start_order([7]) == [0]
    assert _cabi.dp_start_order([10, 700, 10, 64, 63]) == [1, 3, 0, 2, 4]


def test_coding_score_tasks_cover_every_node_once_and_fit_a_round():
    rng = np.random.default_rng(6)
    nodes = rng.integers(200, 1200, 4000).tolist()
    first = rng.integers(0, 10, 4000).tolist()
    models = rng.integers(1, 7, 4000).tolist()
    for task_nodes in (4096, 8192, 1000):
        s = _cabi.cs_task_summary(nodes, first, models, task_nodes)
        walks = sum(nd * ((m + 3) // 4) for nd, m in zip(nodes, models))        # one walk of a contig's nodes per four models
        assert s["nodes"] == walks and s["largest_task"] <= task_nodes and s["high_columns_first"]
        assert s["tasks"] >= walks // task_nodes and s["entries"] >= sum((m + 3) // 4 for m in models)


def test_a_genome_is_cut_into_many_tasks():
    # one 5 Mbp contig under 16 models: it used to be one workgroup per four models (config 2 took four times as long)
    s = _cabi.cs_task_summary([182_418], [0], [15], 4096)
    assert s["tasks"] == 4 * -(-182_418 // 4096) and s["largest_task"] == 4096 and s["nodes"] == 4 * 182_418
    s = _cabi.cs_task_summary([10_478_082], [3], [1], 4096)
    assert s["tasks"] == -(-10_478_082 // 4096) and s["entries"] == s["tasks"]


def test_contigs_without_nodes_or_models_make_no_task():
    assert _cabi.cs_task_summary([], [], [], 4096)["tasks"] == 0
    s = _cabi.cs_task_summary([0, 500, 300], [0, 0, 2], [3, 0, 2], 4096)
    assert s["nodes"] == 300 and s["tasks"] == 1
    with pytest.raises(ValueError):
        _cabi.cs_task_summary([100], [64], [1], 4096)                           # a column a task cannot name: the global-memory form is used


def test_xcd_order_keeps_the_chains_of_a_contig_on_one_xcd():
    """`pga_dp_xcd_order`: workgroup b runs on XCD b % 8; all chains of a key (a contig under one translation table) must land on the
    same XCD, every chain exactly once, queues in start order (longest first), loads balanced, fillers only at the queues' ends."""
    rng = np.random.default_rng(3)
    keys, nodes = [], []
    for contig in range(700):
        n = int(rng.integers(100, 1500))
        for _ in range(int(rng.integers(1, 7))):
            keys.append(contig); nodes.append(n)
    for contig in range(0, 700, 2):                    # a second translation table on every other contig: its own key
        keys.append(700 + contig); nodes.append(int(rng.integers(100, 1500)))
    out = _cabi.dp_xcd_order(nodes, keys)
    assert len(out) % 8 == 0 and sorted(c for c in out if c >= 0) == list(range(len(nodes)))
    xcd_of_key = {}
    load = [0] * 8
    for b, c in enumerate(out):
        if c < 0:
            assert all(o < 0 for o in out[b::8])       # a filler is never followed by a chain in its queue
            continue
        assert xcd_of_key.setdefault(keys[c], b % 8) == b % 8
        load[b % 8] += nodes[c]
    assert max(load) < 1.05 * min(load)
    for x in range(8):
        q = [nodes[c] >> 6 for c in out[x::8] if c >= 0]
        assert q == sorted(q, reverse=True)             # every queue starts its longest chains first
    assert _cabi.dp_xcd_order([], []) == []
    assert _cabi.dp_xcd_order([5], [0]) == [0, -1, -1, -1, -1, -1, -1, -1]
