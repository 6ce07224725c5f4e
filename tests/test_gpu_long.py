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
    """Up to 256 segments, uniform cold solve only (with its rescue pass up to 192): the all-elastic solve, warm starts
    and ragged batches stay at 64."""
    import torch
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    batch, sh = synth.make_batch(2, 70, config=3)
    db = solver.upload(batch)
    solver.solve(db, sh, elastic=1)                          # rescue pass: three wavefronts per problem at most
    with pytest.raises(native.BtrapzError):
        solver.solve(db, sh, elastic=2)
    big193, sh193 = synth.make_batch(2, 193, config=3)
    with pytest.raises(native.BtrapzError):
        solver.solve(solver.upload(big193), sh193, elastic=1)
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


def test_control_point_buffer_of_the_64_segment_contract_is_not_overrun():
    """btrapz_find_traj_mem's ctrl argument is [12 * 64] (its contract before horizons beyond 64 segments were solved): on
    a longer horizon the first 12 * 64 values are written, S is reported, and nothing behind the buffer is touched;
    btrapz_find_traj_mem_cap with the buffer's size returns them all."""
    import ctypes as C
    kb = synth.scenario1_knots(1, 80)
    params = native.CParams(*[float(v) for v in synth.REFERENCE_WEIGHTS], 1)
    call = native.TrajCall(0, params, kb, cap=4096)
    cost, traj, full = call()
    S = len(full) // 12
    assert cost < 1e10 and S > 64
    buf = np.full(12 * 64 + 64, -7.0)                       # 64 guard values behind the contract's size
    n, ns = C.c_int(0), C.c_int(0)
    t = np.zeros((7, 4096))
    c2 = native.lib().btrapz_find_traj_mem(0, C.byref(call.ti), C.byref(call.cp), 4096, t.ctypes.data, C.byref(n), buf.ctypes.data, C.byref(ns))
    assert c2 == cost and ns.value == S
    assert np.array_equal(buf[:12 * 64], full[:12 * 64]) and (buf[12 * 64:] == -7.0).all()
    small = np.full(100 + 8, -7.0)
    c3 = native.lib().btrapz_find_traj_mem_cap(0, C.byref(call.ti), C.byref(call.cp), 4096, t.ctypes.data, C.byref(n), small.ctypes.data, 100, C.byref(ns))
    assert c3 == cost and np.array_equal(small[:100], full[:100]) and (small[100:] == -7.0).all()
    assert native.lib().btrapz_find_traj_mem_cap(0, C.byref(call.ti), C.byref(call.cp), 4096, t.ctypes.data, C.byref(n), None, 5, C.byref(ns)) == 100000000000.0


@pytest.mark.parametrize("case,variant", [("s712_it11_v1", 1), ("s755_it2486_v1", 1)])
def test_rescue_pass_of_the_long_form(case, variant):
    """Two corridors of 76 and 65 segments the round-3 fuzz campaign found (tests/fuzz/cases/): no feasible trajectory,
    least violation within the rescue tolerance -- the oracle's relaxed solve accepts them, and since the long form has
    a rescue pass (ipm_solve_long_elastic_kernel) so does find_traj: same decision, control points of the relaxed
    problem's solution."""
    w = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", "inputs", "weights.txt"))
    path = os.path.join(os.path.dirname(__file__), "fuzz", "cases", case + ".txt")
    inp = O.ParsedInput(path)
    n, cubes = O.pipeline(variant, inp)
    qp = O.AssembledQp(variant, cubes, O.params_from_weights(w), inp)
    assert n > 64 and qp.solve_exact()[2].status not in (1, 2)
    x, _, info, viol = qp.solve_elastic()
    assert info.status in (1, 2) and viol <= 0.0125 - 0.0005
    params = native.CParams(*[float(v) for v in w], 3)
    cost, traj, ctrl = native.find_traj_mem(variant, params, knots.parse_corridor_file(path), cap=4096)
    st, v = native.find_traj_last_status()
    assert cost < 1e10 and st == 2 and len(ctrl) == 12 * n, (st, v)
    assert np.abs(ctrl - x).max() <= 1e-4 * np.abs(x).max(), np.abs(ctrl - x).max() / np.abs(x).max()
    os.environ["BTRAPZ_ELASTIC"] = "0"                      # strict mode: refused, as the exact solve refuses it
    try:
        assert native.find_traj_mem(variant, params, knots.parse_corridor_file(path), cap=4096)[0] >= 1e10
    finally:
        del os.environ["BTRAPZ_ELASTIC"]
