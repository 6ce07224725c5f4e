"""Vectors the REFERENCE ITSELF produced (tests/golden/ref_outputs/*.txt are trajectory files
from the reference repo, 7 columns `t s l ds dl dds ddl`, 3 decimals, trp_wrapper.cpp:298-301).

The weights of each saved run were not recorded, but two input/output pairs are reproduced to
print precision with the s weights of weights.txt (found by searching all_weights.txt):
  * c4.txt / c5.txt -> s4_slt_3d.txt, s4_cub_3d.txt, s5_slt_3d.txt.  OSQP hits max_iter there
    (status 2, "solved inaccurate"): the reference wrote OSQP's UNCONVERGED iterate, 0.34 m
    from the optimum.  The oracle's OSQP port lands on the same 3 decimals after 5000
    iterations -> this pins parser + corridor pipeline + assembly + ADMM port + sampling.
  * c2.txt -> s columns of s2_slt_3d_4.txt / s2_slt_3d_5.txt (OSQP converged): both the OSQP
    port and x* agree with the reference's file to print precision.
The other pairs only pin the row count 1 + sum floor(t_k/delta) and the first row."""
import os

import numpy as np
import pytest

from helpers import O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
W = np.loadtxt(os.path.join(GOLD, "inputs", "weights.txt"))
# all_weights.txt row 47 (0-based, numeric rows only): weights.txt with weight_l_ref = 49.71.  OSQP's
# stopping point couples the axes (joint termination test, joint adaptive rho), so reproducing its
# s-axis ITERATE needs the l weights of the saved run too; x* of the s axis does not depend on them.
W47 = np.array([35.73, 41.61, 25.57, 41.59, 0.12, 10.04, 49.71, 14.3, 7.27, 32.13])
PRINT = 5.0e-4 + 2e-5  # half a unit of the third decimal + slack for values that sit on a rounding edge


def run(name, variant, exact, tmp_path, weights=W47):
    out = str(tmp_path / "traj.txt")
    p = O.params_from_weights(weights)
    if not exact:  # the whole reference driver (parse -> corridors -> assemble -> OSQP port -> sample -> file)
        path = os.path.join(GOLD, "inputs", name + ".txt")
        cost, S, ctrl, cubes, info = O.find_traj(variant, path, out, p)
        inp = O.ParsedInput(path)
        rc, s = O.sample(cubes, inp.delta, ctrl, inp.init_s, inp.init_l)
        full = np.stack([np.arange(len(s[0])) * inp.delta, s[0], s[3], s[1], s[4], s[2], s[5]], 1)
        written = np.loadtxt(out)
        assert written.shape == full.shape and np.abs(written - full).max() <= PRINT  # file = 3-decimal rounding
        return cost, full, info
    inp = O.ParsedInput(os.path.join(GOLD, "inputs", name + ".txt"))
    n, cubes = O.pipeline(variant, inp)
    qp = O.AssembledQp(variant, cubes, p, inp)
    x, _, info = qp.solve_exact()
    rc, s = O.sample(cubes, inp.delta, x, inp.init_s, inp.init_l)
    return None, np.stack([np.arange(len(s[0])) * inp.delta, s[0], s[3], s[1], s[4], s[2], s[5]], 1), info


@pytest.mark.parametrize("name,variant,ref", [("c4", 0, "s4_slt_3d.txt"), ("c4", 1, "s4_cub_3d.txt"),
                                              ("c5", 0, "s5_slt_3d.txt"), ("c4", 0, "test_s4_slt_3d.txt")])
def test_oracle_reproduces_reference_file_c4(name, variant, ref, tmp_path):
    cost, got, info = run(name, variant, False, tmp_path)
    want = np.loadtxt(os.path.join(GOLD, "ref_outputs", ref))
    assert info.status == 2 and info.iter == 5000      # the reference accepted "solved inaccurate"
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= PRINT
    assert cost < 1e10


@pytest.mark.parametrize("ref", ["s2_slt_3d_4.txt", "s2_slt_3d_5.txt"])
@pytest.mark.parametrize("exact", [False, True])
def test_s_axis_of_scenario_2_matches_reference_file(ref, exact, tmp_path):
    _, got, info = run("c2", 0, exact, tmp_path, weights=W if exact else W47)
    want = np.loadtxt(os.path.join(GOLD, "ref_outputs", ref))
    assert info.status == 1
    assert got.shape == want.shape
    cols = [0, 1, 3, 5]  # t, s, ds, dds
    assert np.abs(got[:, cols] - want[:, cols]).max() <= PRINT


def test_xstar_differs_from_the_unconverged_reference_iterate(tmp_path):
    """Documents WHY parity is defined against x*: on c4 the reference's own answer is 0.34 m off."""
    _, got, info = run("c4", 0, True, tmp_path)
    want = np.loadtxt(os.path.join(GOLD, "ref_outputs", "s4_slt_3d.txt"))
    assert info.status == 1
    assert 0.2 < np.abs(got[:, 1] - want[:, 1]).max() < 0.5


PAIRS = [("c1", 0, "s1_slt_3d_30.txt"), ("c1", 0, "s1_slt_3d_31.txt"), ("c1", 0, "s1_slt_3d_500.txt"),
         ("c1", 1, "s1_cub_3d_3.txt"), ("c1", 1, "s1_cub_3d_4.txt"), ("c1", 1, "s1_cub_3d_30.txt"),
         ("c1", 1, "s1_cub_3d_31.txt"), ("c2", 1, "s2_cub_3d_3.txt"),
         ("c2", 1, "s2_cub_3d_5.txt"), ("c_road_s1", 0, "s1_slt_3d_200.txt")]


@pytest.mark.parametrize("name,variant,ref", PAIRS)
def test_row_count_and_first_row_of_saved_outputs(name, variant, ref):
    """Weak anchor for pairs whose weights are unknown: number of samples and the echoed
    initial state (solve_3d.cc:1279-1282, 1325-1331)."""
    inp = O.ParsedInput(os.path.join(GOLD, "inputs", name + ".txt"))
    n, cubes = O.pipeline(variant, inp)
    want = np.loadtxt(os.path.join(GOLD, "ref_outputs", ref))
    rows = 1 + sum(int(c.t / inp.delta) for c in cubes)
    assert rows == want.shape[0]
    first = [0.0, inp.init_s[0], inp.init_l[0], inp.init_s[1], inp.init_l[1], inp.init_s[2], inp.init_l[2]]
    assert np.abs(want[0] - first).max() <= PRINT


def test_w47_is_a_row_of_all_weights():
    rows = []
    for line in open(os.path.join(GOLD, "inputs", "all_weights.txt")):
        try:
            v = [float(t) for t in line.split()]
        except ValueError:
            continue
        if len(v) >= 10:
            rows.append(v[:10])
    assert np.allclose(rows[47], W47)
    assert np.allclose(np.delete(W47, 6), np.delete(W, 6))


def test_weight_search_result_is_what_the_goldens_pin():
    """tests/golden/find_weights.py ran every bundled input x both variants x all 204 numeric rows (288 lines) of all_weights.txt (+
    weights.txt) against every saved 7-column trajectory of the reference (58 files, ref_outputs/).  Its committed
    result: four files are reproduced in all seven columns (pinned above), the s columns of two more; for every other
    saved file NO row of the trial log reproduces it -- those runs used weights that were not kept (or other inputs)."""
    import json
    res = json.load(open(os.path.join(GOLD, "weight_search.json")))
    assert res["weight_rows"] == 205 and len(res["files"]) == 58
    full = {f for f, r in res["files"].items() if r["all"] and r["all"]["matches_to_print_precision"]}
    s_only = {f for f, r in res["files"].items() if r["s"] and r["s"]["matches_to_print_precision"]} - full
    assert full == {"s4_slt_3d.txt", "s4_cub_3d.txt", "s5_slt_3d.txt", "test_s4_slt_3d.txt"}
    assert s_only == {"s2_slt_3d_4.txt", "s2_slt_3d_5.txt"}
    for f in full:
        assert res["files"][f]["all"]["weight_row"] == 47 and res["files"][f]["all"]["mode"] == "osqp"
    # scenario_1 (c1 <-> s1_*), c3, c_road_s1 and the c7 family: best distances are 0.2 .. 5 m -- nothing to pin
    assert res["files"]["s1_slt_3d_30.txt"]["all"]["max_abs_diff"] > 0.1
    assert res["files"]["s7_slt_3d_3.txt"]["all"]["max_abs_diff"] > 0.1


def test_scenario1_cannot_be_pinned_by_a_reference_written_file(tmp_path):
    """VERDICT r5 item 2.  scenario_1 = src/c1.txt.  Every hypothesis that could tie one of the saved s1_* trajectories to it
    (tests/golden/pin_scenario1.py: the OSQP port's stopping point under the 204 logged weight rows; the old libbtrapz.so's
    c1.txt -> slt_3d.txt pair; the segment references read at four other places than solve_3d.cc:1159-1166 reads them) ends
    at "no": no file comes closer than 0.2 m in all seven columns.  The cause is in the reference: its harness rewrites the
    corridor file at every replanning step (cart_frenet.py:385-386), so a saved output's input is gone unless the run
    happened to be the last one -- c4 / c5 / c2 were (pinned above), c1 was not.  scenario_1's anchor therefore stays the row
    count + first row (PAIRS) and x* -- and the test below is a NOTE on how close a free fit gets, not evidence."""
    import json, subprocess, sys
    gold = os.path.join(os.path.dirname(__file__), "golden")
    rec = json.load(open(os.path.join(gold, "scenario1_pin_search.json")))
    assert rec["pinned_by_any_hypothesis"] == []
    assert len(rec["hypotheses"]) == 7
    for name, h in rec["hypotheses"].items():
        assert h["pinned"] == []
        for f, r in h.get("files", {}).items():
            if r["all"] is not None:
                assert r["all"]["max_abs_diff"] > 0.2, (name, f)     # 400 x the print precision
    h2 = rec["hypotheses"]["H2_old_library_pair"]
    assert not h2["slt_3d.txt"]["first_row_is_c1s"] and h2["slt_3d.txt"]["first_row"][3] == 5.0
    assert h2["s1_slt_3d.txt"]["first_row_is_c1s"] and h2["s1_slt_3d.txt"]["rows"] == 75
    assert h2["s1_slt_3d.txt"]["rows_from_c1_trapezoid"] == 70 and h2["s1_slt_3d.txt"]["rows_from_c1_cuboid"] == 74
    # the committed record is what the script writes (the whole search takes seconds)
    work = tmp_path / "golden"
    import shutil
    shutil.copytree(gold, str(work), ignore=shutil.ignore_patterns("*.npz", "corridors.json", "__pycache__"))
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(__file__)))
    src = open(os.path.join(str(work), "pin_scenario1.py")).read().replace("ROOT = os.path.dirname(os.path.dirname(HERE))",
                                                                           "ROOT = %r" % os.path.dirname(os.path.dirname(__file__)))
    open(os.path.join(str(work), "pin_scenario1.py"), "w").write(src)
    r = subprocess.run([sys.executable, os.path.join(str(work), "pin_scenario1.py")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    again = json.load(open(os.path.join(str(work), "scenario1_pin_search.json")))
    assert again == rec


FIT = [("s1_slt_3d_30.txt", "c1", 0, 0.0085), ("s1_cub_3d_3.txt", "c1", 1, 0.0150), ("s1_cub_3d_30.txt", "c1", 1, 0.0205)]


@pytest.mark.parametrize("ref,name,variant,s_residual", FIT)
def test_scenario1_lateral_columns_are_reproduced_by_fitted_weights(ref, name, variant, s_residual, tmp_path):
    """A NOTE, not a pin (ten free parameters that run to 1.6e5 are a fit): how close a free choice of weights
    brings the restatement to three files the reference wrote from some version of src/c1.txt.
    Round 3 (tests/golden/fit_weights.py -> weight_fit.json): no LOGGED weight row reproduces the saved scenario_1
    trajectories, but a continuous fit of the weights does on the lateral axis -- the l, dl, ddl columns of three files
    the reference wrote from src/c1.txt come back to print precision from the restatement's x* -- and brings the
    longitudinal columns from 0.05-0.7 to the stated residual (a floor every start of the fit ends at: those runs
    differ from the bundled input in more than the weights)."""
    import json
    fit = json.load(open(os.path.join(GOLD, "weight_fit.json")))["fits"][ref]
    want = np.loadtxt(os.path.join(GOLD, "ref_outputs", ref))
    _, got, info = run(name, variant, True, tmp_path, weights=np.array(fit["weights"]))
    assert info.status == 1 and got.shape == want.shape
    assert np.abs(got[:, [2, 4, 6]] - want[:, [2, 4, 6]]).max() <= PRINT
    assert fit["l"]["matches_to_print_precision"] and not fit["s"]["matches_to_print_precision"]
    assert np.abs(got[:, [1, 3, 5]] - want[:, [1, 3, 5]]).max() <= s_residual
    # the best starts of the fit ended at the same longitudinal residual: a floor, not a miss of the search
    r = fit["s"]["residuals_of_all_starts"]
    assert r[4] - r[0] <= 1e-3


def test_distance_of_the_references_own_answer_from_the_optimum_is_tabulated():
    """parity is held against the QP's optimum x*, not against OSQP's stopping point.  How far the two are
    apart is a property of the reference (eps 1e-5, no polish, 5000 iterations), pinned here per bundled input by the
    oracle's OSQP port: north_star's 1e-4 on control points holds against the reference's OUTPUT only where OSQP itself
    converged that far -- scenario_1 (c1) -- and cannot hold on c3 (4e-3), c_road_s1_3 (5e-3) or c4 / c5, whose saved
    reference trajectories are the unconverged iterate after 5000 iterations (0.34 control-point units off)."""
    import json
    gold = os.path.join(os.path.dirname(__file__), "golden")
    table = json.load(open(os.path.join(gold, "acceptance_table.json")))
    rows = {(r["input"], r["variant"]): r for r in table["rows"]}
    w = np.loadtxt(os.path.join(gold, "inputs", "weights.txt"))
    p = O.params_from_weights(w)
    for key in (("c1", 0), ("c2", 1), ("c3", 0), ("c4", 0)):
        path = os.path.join(gold, "inputs", key[0] + ".txt")
        _, _, xp, _, info = O.find_traj(key[1], path, None, p)
        inp = O.ParsedInput(path)
        n, cubes = O.pipeline(key[1], inp)
        x, _, ie = O.AssembledQp(key[1], cubes, p, inp).solve_exact()
        assert ie.status in (1, 2) and info.status == rows[key]["port_status"]
        rel = np.abs(np.asarray(xp) - x).max() / np.abs(x).max()
        assert abs(rel - rows[key]["port_vs_xstar_rel"]) <= 1e-3 * rows[key]["port_vs_xstar_rel"] + 1e-9
    both = [r for r in table["rows"] if r["port_vs_xstar_rel"] is not None]
    assert len(both) == 13
    assert rows[("c1", 0)]["port_vs_xstar_rel"] < 1e-5 and rows[("c1", 1)]["port_vs_xstar_rel"] < 1e-4
    assert sum(r["port_vs_xstar_rel"] <= 1e-4 for r in both) == 2           # only scenario_1 meets 1e-4 against the reference's output
    assert max(r["port_vs_xstar_rel"] for r in both if r["port_status"] == 1) < 5e-3
    assert all(abs(r["port_vs_xstar_abs"] - 0.341) < 1e-3 for r in both if r["port_status"] == 2 and r["input"] in ("c4", "c5"))


def test_committed_xstar_fixtures_are_what_the_generator_writes(tmp_path):
    """The x* fixtures the GPU parity tests load (synthetic_xstar / scenario1_xstar / scenario_xstar .npz, corridors.json)
    are products of the ORACLE's exact solver, not of the HIP solver: regenerating them with
    tests/golden/make_golden.py gives the committed arrays bit for bit (round 4: regenerated, byte-identical)."""
    import subprocess, sys
    gold = os.path.join(os.path.dirname(__file__), "golden")
    env = dict(os.environ, GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(gold, "make_golden.py")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    for f in ("scenario1_xstar.npz", "scenario_xstar.npz", "synthetic_xstar.npz"):
        a, b = np.load(os.path.join(gold, f)), np.load(str(tmp_path / f))
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            assert np.array_equal(a[k], b[k], equal_nan=True), (f, k)
    assert open(os.path.join(gold, "corridors.json")).read() == open(str(tmp_path / "corridors.json")).read()
