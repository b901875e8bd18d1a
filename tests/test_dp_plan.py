"""Host logic of the segmented connection scoring (dp.hip `pga_dp_plan`, through `pga_dp_plan_summary`): which chains of a
launch are cut, into how many segments, with how much scratch.  Pure host arithmetic: runs without a GPU.
The reference has no counterpart (its dynamic programme is one serial loop, lib.pyx:1205-1237); what is pinned here is the
contract DESIGN.md 4.4 states: at most one workgroup per compute unit, every node of a cut chain in exactly one segment
(checked inside the call), short chains and launches with many chains left alone."""
import os

import pytest

from pyrodigal_amd import _cabi

SEG_ENV = ("PGA_DP_SEG", "PGA_DP_SEG_MIN", "PGA_DP_SEG_LEN", "PGA_DP_SEG_WARM", "PGA_DP_SEG_SLOTS", "PGA_DP_KERNEL")


@pytest.fixture(autouse=True)
def clean_env():
    saved = {k: os.environ.pop(k, None) for k in SEG_ENV}
    yield
    for k, v in saved.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


def test_short_chains_are_walked_whole():
    assert _cabi.dp_plan_summary([]) == {"chains": 0, "segments": 0, "max_sub_chain": 0, "scratch": 0}
    assert _cabi.dp_plan_summary([1900] * 100)["chains"] == 0
    assert _cabi.dp_plan_summary([16383])["chains"] == 0


def test_one_genome_fills_the_chip_but_not_more():
    for n in (20_000, 153_296, 1_000_000, 10_478_082):
        p = _cabi.dp_plan_summary([n])
        assert p["chains"] == 1 and 2 <= p["segments"] <= 252, (n, p)
        assert p["max_sub_chain"] <= n // p["segments"] + 64 + 4096 + n // (4 * p["segments"]) + 64     # segment (+ merged tail) + warm-up (PGA_DP_SEG_WARM, 4096)
        assert p["scratch"] >= n                                                                    # every segment keeps its own results


def test_config2_like_launch_stays_within_the_compute_units():
    chains = [182418, 182418, 182418, 182418, 269041]
    p = _cabi.dp_plan_summary(chains)
    assert p["chains"] == 5 and p["segments"] <= 252
    mixed = _cabi.dp_plan_summary(chains + [1900] * 40)
    assert mixed["chains"] == 5 and mixed["segments"] + 40 <= 252


def test_many_chains_or_switch_off_disable_it():
    assert _cabi.dp_plan_summary([20_000] * 2048)["chains"] == 0         # the one-wave kernel fills the chip already
    os.environ["PGA_DP_SEG"] = "0"
    assert _cabi.dp_plan_summary([1_000_000])["chains"] == 0
    os.environ.pop("PGA_DP_SEG")
    os.environ["PGA_DP_KERNEL"] = "scan"                                  # the cross-check kernels run whole chains
    assert _cabi.dp_plan_summary([1_000_000])["chains"] == 0


def test_environment_overrides():
    os.environ.update({"PGA_DP_SEG_MIN": "300", "PGA_DP_SEG_LEN": "256", "PGA_DP_SEG_WARM": "64"})
    p = _cabi.dp_plan_summary([10_000, 200, 700])
    assert p["chains"] == 2 and p["segments"] >= 10_000 // 256 + 2
    assert p["max_sub_chain"] <= 256 + 64 + 64
    with pytest.raises(ValueError):
        _cabi.dp_plan_summary([-1])


def test_a_genome_among_hundreds_of_small_contigs_is_still_cut_and_planned_quickly():
    import time
    t0 = time.time()
    p = _cabi.dp_plan_summary([10_000_000] + [1500] * 2000)
    assert time.time() - t0 < 0.5
    assert p["chains"] == 1 and 16 <= p["segments"] <= 64 + 8
    p = _cabi.dp_plan_summary([300_000] + [1500] * 100)
    assert p["chains"] == 1 and p["segments"] + 100 <= 252 + 8
