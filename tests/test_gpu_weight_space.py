"""The HIP path over the reference's WEIGHT space: the reference's real caller is an Optuna objective
that draws every one of the ten Params weights from U(0, 50) (src/trp_wrapper.py:56-97); its trial log all_weights.txt
holds 204 ten-column rows, from 0.0 to 49.99 per weight.  A slice of the two campaign drivers runs here, in the driver's
suite (the full sweeps: profiles/r05_fuzz_campaign.txt):

  tests/fuzz/weights_find_traj.py  find_traj -- both single-candidate kernels, in-memory and through CDLL(libtrp / libcub)
                                   -- on 3 bundled inputs x 2 variants x (every 5th trial row + 24 seeded draws + the
                                   degenerate rows: a *_ref / end weight of 0, all weights 1e-6, the corners of the box):
                                   accept decision and control points against the oracle's exact (or relaxed) solve;
  tests/fuzz/weights_batched.py    every batched form (lean / packed, one and two launches, ragged, warm, split) on 256
                                   candidates of each bench family under 3 rows chosen for spread + the degenerate rows +
                                   other header limits + the reference's default +-1e10 bounds
                                   (src/piecewise_jerk_problem.cc:9,25-35); then ragged batches made from the bundled
                                   corridor files' knots (4 forms each)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
FUZZ = os.path.join(os.path.dirname(__file__), "fuzz")


def test_find_traj_over_the_weight_space():
    env = dict(os.environ, WFT_FILE_ROW_STRIDE="5", PYTHONUNBUFFERED="1")
    p = subprocess.run([sys.executable, os.path.join(FUZZ, "weights_find_traj.py"), "7", "24", "1", "c1,c2,c7_7", "8"],
                       capture_output=True, text=True, env=env, timeout=900)
    tail = p.stdout[-3000:]
    assert p.returncode == 0, (tail, p.stderr[-1500:])
    m = re.search(r"'calls': (\d+), 'agree': (\d+), 'accepted': (\d+), 'rejected': (\d+), 'decisions_apart': (\d+), 'xstar_beyond': (\d+)", p.stdout)
    assert m, tail
    calls, agree, accepted, rejected, apart, beyond = map(int, m.groups())
    assert calls >= 1000 and agree == calls and apart == 0 and beyond == 0 and accepted >= 700, tail
    assert "'cdll_cost_differs': 0" in p.stdout and "'obj_beyond': 0" in p.stdout, tail


def test_batched_forms_over_the_weight_space():
    p = subprocess.run([sys.executable, os.path.join(FUZZ, "weights_batched.py"), "256", "3", "0", "-", "8"],
                       capture_output=True, text=True, env=dict(os.environ, PYTHONUNBUFFERED="1"), timeout=900)
    tail = p.stdout[-3000:]
    assert p.returncode == 0, (tail, p.stderr[-1500:])
    m = re.search(r"'cases': (\d+), 'forms': (\d+), 'candidates': (\d+), 'accept_differences': (\d+), 'beyond_tolerance': (\d+)", p.stdout)
    assert m, tail
    cases, forms, cand, acc, beyond = map(int, m.groups())
    assert cases >= 36 and forms >= 4 * cases and cand >= 100000 and acc == 0 and beyond == 0, tail   # (9 forms per bench family, 4 per ragged case)
    assert "'objective_beyond': 0" in p.stdout, tail
