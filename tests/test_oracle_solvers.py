"""The oracle's two solvers against each other, against KKT certificates and against a
third-party QP solver (HiGHS, bundled with scipy) -- the pin on x* (SURVEY 8c)."""
import os

import numpy as np
import pytest

from helpers import O, oracle_qp_from_batch
from spectral_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
W = np.loadtxt(os.path.join(GOLD, "inputs", "weights.txt"))
FEASIBLE = [("c1", 0), ("c1", 1), ("c2", 0), ("c2", 1), ("c3", 0), ("c3", 1), ("c4", 0), ("c6", 0), ("c_road_s1", 0),
            ("c_road_s1_3", 0)]
INFEASIBLE = [("c7", 0), ("c7_7", 1), ("c_road_s1_2", 0), ("c_road_s1_3", 1)]


def load(name, variant):
    inp = O.ParsedInput(os.path.join(GOLD, "inputs", name + ".txt"))
    n, cubes = O.pipeline(variant, inp)
    return inp, cubes, O.AssembledQp(variant, cubes, O.params_from_weights(W), inp)


@pytest.mark.parametrize("name,variant", FEASIBLE)
def test_xstar_has_a_kkt_certificate(name, variant):
    _, _, qp = load(name, variant)
    x, y, info = qp.solve_exact()
    assert info.status == 1
    stat, viol, comp = qp.kkt(x, y)
    scale = 1 + np.abs(qp.q).max()
    assert stat < 1e-8 * scale * 1e2 and viol < 1e-9 and comp < 1e-6


@pytest.mark.parametrize("name,variant", FEASIBLE)
def test_xstar_matches_committed_golden(name, variant):
    _, _, qp = load(name, variant)
    x, _, info = qp.solve_exact()
    g = np.load(os.path.join(GOLD, "scenario_xstar.npz"))
    ref = g["%s/%d/xstar" % (name, variant)]
    assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max()


@pytest.mark.parametrize("name,variant", [("c1", 0), ("c2", 0), ("c3", 1), ("c_road_s1_3", 0)])
def test_tight_admm_converges_to_xstar(name, variant):
    """Independent algorithm (OSQP-style ADMM at eps 1e-9) reaches the same point."""
    _, _, qp = load(name, variant)
    x, _, info = qp.solve_exact()
    st = O.settings_tight(); st.polish = 0
    xt, _, it = qp.solve(st)
    assert it.status == 1
    assert np.abs(x - xt).max() <= 1e-6 * np.abs(x).max()


@pytest.mark.parametrize("name,variant", [("c1", 0), ("c2", 1)])
def test_xstar_matches_highs(name, variant):
    """Third-party check: HiGHS' QP solver (scipy's bundled highspy core)."""
    hs = pytest.importorskip("scipy.optimize._highspy._core")
    import scipy.sparse as sp
    _, _, qp = load(name, variant)
    x, _, info = qp.solve_exact()
    P, A = qp.dense()
    h = hs._Highs()
    h.setOptionValue("output_flag", False)
    lp = hs.HighsLp()
    n, m = qp.n, qp.m
    lp.num_col_, lp.num_row_ = n, m
    lp.col_cost_ = qp.q.tolist()
    lp.col_lower_ = [-hs.kHighsInf] * n; lp.col_upper_ = [hs.kHighsInf] * n
    lp.row_lower_ = qp.l.tolist(); lp.row_upper_ = qp.u.tolist()
    Ac = sp.csc_matrix(A)
    lp.a_matrix_.format_ = hs.MatrixFormat.kColwise
    lp.a_matrix_.start_ = Ac.indptr.tolist(); lp.a_matrix_.index_ = Ac.indices.tolist(); lp.a_matrix_.value_ = Ac.data.tolist()
    hess = hs.HighsHessian()
    Pl = sp.csc_matrix(np.tril(P))
    hess.dim_ = n; hess.format_ = hs.HessianFormat.kTriangular
    hess.start_ = Pl.indptr.tolist(); hess.index_ = Pl.indices.tolist(); hess.value_ = Pl.data.tolist()
    model = hs.HighsModel(); model.lp_ = lp; model.hessian_ = hess
    if h.passModel(model) != hs.HighsStatus.kOk:
        pytest.skip("this HiGHS build rejects the QP model")
    h.run()
    xh = np.array(h.getSolution().col_value)
    if xh.shape != x.shape or not np.isfinite(xh).all():
        pytest.skip("HiGHS returned no QP solution")
    assert np.abs(xh - x).max() <= 2e-5 * np.abs(x).max()


@pytest.mark.parametrize("name,variant", FEASIBLE)
def test_osqp_port_lands_in_its_tolerance_band(name, variant):
    """The reference's solver (eps 1e-5, max_iter 5000) stops near x*, not at it."""
    _, _, qp = load(name, variant)
    x, _, _ = qp.solve_exact()
    xo, _, io = qp.solve()
    assert io.status in (1, 2, -2)
    rel = np.abs(xo - x).max() / np.abs(x).max()
    if io.status == 1:
        assert rel < 2e-2
    g = np.load(os.path.join(GOLD, "scenario_xstar.npz"))
    assert np.abs(xo - g["%s/%d/osqp" % (name, variant)]).max() <= 1e-9 * np.abs(xo).max() + 1e-12


@pytest.mark.parametrize("name,variant", INFEASIBLE)
def test_infeasible_corridors_are_rejected_by_both(name, variant):
    _, _, qp = load(name, variant)
    _, _, info = qp.solve_exact()
    _, _, io = qp.solve()
    assert info.status not in (1,) and io.status not in (1,)


@pytest.mark.parametrize("cfg,S,variant", [(2, 10, 0), (3, 20, 0), (4, 20, 1)])
def test_synthetic_goldens(cfg, S, variant):
    g = np.load(os.path.join(GOLD, "synthetic_xstar.npz"))
    batch, sh = synth.make_batch(256, S, config=cfg, variant=variant)
    ctrl, obj, st, it = O.batch_solve(batch, sh, 0, 4, exact=True)
    assert (st == 1).all()
    assert np.abs(ctrl - g["cfg%d/xstar" % cfg][:4]).max() <= 1e-9 * np.abs(ctrl).max()
    qp = oracle_qp_from_batch(batch, sh, 2)
    x, y, info = qp.solve_exact()
    assert np.abs(x - ctrl[2]).max() <= 1e-9 * np.abs(x).max()
