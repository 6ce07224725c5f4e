"""More than 64 segments per candidate (the long form of the solve kernel: one axis problem per workgroup of several
wavefronts, MULTI in spectral_amd/csrc/btrapz_kernels.hip).  The reference has no limit on the segment count
(std::vector throughout, solve_3d.cc:323-486); its bundled inputs have at most 14.  Held to the oracle's x*."""
import os

import numpy as np
import pytest

from helpers import O, oracle_qp_from_batch
from spectral_amd import knots, native, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
W = np.loadtxt(os.path.join(GOLD, "inputs", "weights.txt"))


@pytest.mark.parametrize("gen,S,variant", [("generic", 65, 0), ("generic", 100, 0), ("scenario1", 128, 0), ("generic", 130, 0),
                                           ("scenario1", 200, 0), ("generic", 256, 0)])
def test_long_corridors_reach_the_oracle_optimum(gen, S, variant):
    import torch
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    B = 5
    batch, sh = (synth.make_scenario1_batch(B, S, variant) if gen == "scenario1" else synth.make_batch(B, S, config=3, variant=variant))
    o = solver.solve(solver.upload(batch), sh)
    torch.cuda.synchronize()
    st = o["status"].cpu().numpy(); ctrl = o["ctrl"].cpu().numpy(); cost = o["cost"].cpu().numpy()
    checked = 0
    for b in range(2 if S > 150 else 3):
        qp = oracle_qp_from_batch(batch, sh, b)
        x, _, info = qp.solve_exact(max_iter=120)
        assert (info.status in (1, 2)) == (st[b] > 0), (b, info.status, st[b])
        if st[b] > 0:
            assert np.abs(ctrl[b] - x).max() <= 1e-5 * np.abs(x).max()
            P, _ = qp.dense()
            assert abs(cost[b] - (0.5 * x @ P @ x + qp.q @ x)) <= 1e-6 * abs(cost[b])
            checked += 1
    assert checked >= 1


def test_long_form_limits_and_refusals():
    """Up to 256 segments, uniform cold solve only: the rescue pass, warm starts and ragged batches stay at 64."""
    import torch
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    batch, sh = synth.make_batch(2, 70, config=3)
    db = solver.upload(batch)
    with pytest.raises(native.BtrapzError):
        solver.solve(db, sh, elastic=1)
    with pytest.raises(native.BtrapzError):
        solver.solve(db, sh, keep_multipliers=True)
    big, sh2 = synth.make_batch(2, 257, config=3)
    with pytest.raises(native.BtrapzError):
        solver.solve(solver.upload(big), sh2)
    torch.cuda.synchronize()


def test_find_traj_on_a_long_horizon():
    """find_traj on a scene of more than 64 one-second pieces: host corridor stage, the long form through the batched
    entry point, sampling -- decision and trajectory of the oracle's restatement of the same call."""
    kb = synth.scenario1_knots(1, 80)                       # N = 801 knots, two lane corridors
    params = native.CParams(*[float(v) for v in synth.REFERENCE_WEIGHTS], 1)
    cost, traj, ctrl = native.find_traj_mem(0, params, kb, cap=4096)
    assert cost < 1e10
    S = len(ctrl) // 12
    assert S > 64 and native.find_traj_last_status()[0] in (1, 2)
    import tempfile
    path = os.path.join(tempfile.mkdtemp(), "long.txt")
    knots.write_corridor_file(path, kb, 0)
    inp = O.ParsedInput(path)
    n, cubes = O.pipeline(0, inp)
    assert n == S
    qp = O.AssembledQp(0, cubes, O.params_from_weights(synth.REFERENCE_WEIGHTS), inp)
    x, _, info = qp.solve_exact(max_iter=120)
    assert info.status in (1, 2) and np.abs(ctrl - x).max() <= 1e-5 * np.abs(x).max()
    rc, smp = O.sample(cubes, inp.delta, x, inp.init_s, inp.init_l)
    assert rc == 0 and traj.shape[1] == len(smp[0])
    assert np.abs(traj[1] - smp[0]).max() <= 1e-4 * max(1.0, np.abs(smp[0]).max())      # s column against x*'s samples
