"""The pair steps of k_dp_wave are generated assembly (pyrodigal_amd/csrc/dpw_walk_gfx950.inc, tools/gen_dpw_walk.py): the checked-in
file must be what the generator writes today, and the generator's own check of the gfx950 wait-state rules must pass."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "tools", "gen_dpw_walk.py")
INC = os.path.join(ROOT, "pyrodigal_amd", "csrc", "dpw_walk_gfx950.inc")


def test_checked_in_assembly_is_the_generators_output():
    env = {k: v for k, v in os.environ.items() if k != "DPW_EXP"}
    out = subprocess.run([sys.executable, GEN], check=True, capture_output=True, text=True, env=env).stdout
    assert out == open(INC).read(), "regenerate: python tools/gen_dpw_walk.py > pyrodigal_amd/csrc/dpw_walk_gfx950.inc"


def test_wait_states_hold_over_the_control_flow_graph():
    r = subprocess.run([sys.executable, GEN, "--check"], capture_output=True, text=True)
    assert r.returncode == 0 and "wait states ok" in r.stdout, r.stderr


def test_the_check_sees_a_missing_wait_state(tmp_path):
    # the same generator with one pad taken out must fail its check (the checker is not vacuous)
    src = open(GEN).read()
    good = 'a("s_nop 0" if not near else "s_nop 1")'
    assert src.count(good) == 1
    bad = tmp_path / "gen_bad.py"
    bad.write_text(src.replace(good, 'a("s_nop 0")'))
    r = subprocess.run([sys.executable, str(bad), "--check"], capture_output=True, text=True)
    assert r.returncode != 0 and "needs 2" in r.stderr
