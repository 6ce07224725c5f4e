"""Synthetic corridor generator: deterministic, right shapes, feasible."""
import numpy as np
import pytest

from helpers import O
from spectral_amd import layout as L
from spectral_amd import synth


@pytest.mark.parametrize("cfg,S,variant", [(2, 10, 0), (3, 20, 0), (4, 20, 1)])
def test_shapes_determinism_and_feasibility(cfg, S, variant):
    b1, sh = synth.make_batch(64, S, config=cfg, variant=variant)
    b2, _ = synth.make_batch(64, S, config=cfg, variant=variant)
    assert b1.seg.shape == (L.NUM_SEG_FIELDS, 64, S) and b1.init.shape == (64, 6)
    assert (b1.seg == b2.seg).all() and (b1.init == b2.init).all()
    assert (b1.seg[L.F_T] == 1.0).all()
    assert (b1.seg[L.F_UPP_BIAS] > b1.seg[L.F_DOWN_BIAS]).all()
    ctrl, obj, st, it = O.batch_solve(b1, sh, 0, 12, exact=True)
    assert (st == 1).all()                         # feasible by construction
    assert b1.algorithmic_bytes() == (17 * S + 18) * 8 + 96 * S + 12


def test_reference_weights_are_weights_txt():
    import os
    w = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", "inputs", "weights.txt"))
    assert np.allclose(w, synth.REFERENCE_WEIGHTS)
    sh = synth.shared_params()
    assert sh.w_s == (w[4], w[5], w[0], w[1]) and sh.w_l == (w[6], w[7], w[2], w[3])
