"""Error behaviour of the batched C-ABI on a device: bad arguments come back as BTRAPZ_EINVAL with a message
(never a crash -- the reference's CHECK_* abort the process, include/btrapz/logging.h:256-258), and a failed call
leaves the context usable."""
import ctypes as C

import numpy as np
import pytest

from spectral_amd import native, synth

pytestmark = pytest.mark.gpu
EINVAL = -1


def test_bad_arguments_are_reported_not_fatal():
    import torch
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    lib, h = native.lib(), solver.ctx._h
    batch, sh = synth.make_batch(8, 10, config=2)
    db = solver.upload(batch)
    o = solver.solve(db, sh)
    torch.cuda.synchronize()
    good = o["ctrl"].clone()
    csh = native.CShared.from_shared(sh)
    p = lambda t: C.c_void_p(t.data_ptr())
    args = lambda B, S, seg=db.seg, ctrl=o["ctrl"]: (h, C.byref(csh), None, B, S, p(seg), p(db.init), p(db.ref_end),
                                                     p(db.dl_bounds), p(ctrl), p(o["cost"]), p(o["status"]), p(o["iters"]), None)
    assert lib.btrapz_solve_batch_device(*args(0, 10)) == EINVAL            # B < 1
    assert lib.btrapz_solve_batch_device(*args(8, 0)) == EINVAL             # S < 1
    assert lib.btrapz_solve_batch_device(*args(8, 257)) == EINVAL           # S > 256 (65..256: the long form)
    assert b"invalid" in lib.btrapz_last_error(h)
    bad = list(args(8, 10)); bad[5] = None                                  # null seg
    assert lib.btrapz_solve_batch_device(*bad) == EINVAL
    bad = list(args(8, 10)); bad[1] = None                                  # null shared
    assert lib.btrapz_solve_batch_device(*bad) == EINVAL
    assert lib.btrapz_solve_batch_device(None, C.byref(csh), None, 8, 10, *[None] * 9) == EINVAL   # null context
    assert lib.btrapz_argmin_device(h, 8, 3, 0, p(o["cost"]), p(o["cost"]), p(o["cost"]), None) == EINVAL   # B % group
    assert lib.btrapz_sample_device(h, 8, 10, C.c_double(0.0), p(db.seg), p(db.init), p(o["ctrl"]), 1, p(o["cost"]), 8,
                                    p(o["cost"]), p(o["status"]), None) == EINVAL                           # delta <= 0
    assert lib.btrapz_eval_states_device(h, 8, 10, None, p(db.seg), p(o["ctrl"]), 0, p(o["cost"]), p(o["cost"]), None) == EINVAL
    assert lib.btrapz_corridor_batch_device(h, 0, 8, 2, 1, C.c_double(0.1), *[p(o["cost"])] * 6, 16, p(o["cost"]), p(o["status"]),
                                            p(o["cost"]), p(o["cost"]), None) == EINVAL                     # N < 3
    assert lib.btrapz_corridor_batch_device(h, 0, 8, 71, 1001, C.c_double(0.1), *[p(o["cost"])] * 6, 16, p(o["cost"]),
                                            p(o["status"]), p(o["cost"]), p(o["cost"]), None) == EINVAL     # num_obs > 1000 (the parser's bound; 65..1000: the serial kernel)
    assert lib.btrapz_corridor_batch_device(h, 0, 8, 100001, 2, C.c_double(0.1), *[p(o["cost"])] * 6, 16, p(o["cost"]),
                                            p(o["status"]), p(o["cost"]), p(o["cost"]), None) == EINVAL     # N > 100 000
    assert lib.btrapz_corridor_batch_device(h, 0, 8, 71, 2, C.c_double(0.1), *[p(o["cost"])] * 6, 257, p(o["cost"]),
                                            p(o["status"]), p(o["cost"]), p(o["cost"]), None) == EINVAL     # more slots than a solve takes
    # the fused prism + corridor entry point: the limits of the two calls it replaces
    road = native.CRoad.reference()
    pc = lambda P, N, O: lib.btrapz_prism_corridor_batch_device(h, 0, 8, P, N, C.byref(road), p(o["cost"]), O, C.c_double(0.1), *[p(o["cost"])] * 4,
                                                                16, p(o["cost"]), p(o["status"]), p(o["cost"]), p(o["cost"]), None, None)
    assert pc(17, 71, 5) == EINVAL and pc(0, 71, 5) == EINVAL          # 1..16 cars
    assert pc(2, 2, 5) == EINVAL and pc(2, 513, 5) == EINVAL           # 3..512 knots
    assert pc(2, 71, 0) == EINVAL and pc(2, 71, 65) == EINVAL          # 1..64 strips
    assert lib.btrapz_prism_corridor_batch_device(h, 0, 8, 2, 71, None, p(o["cost"]), 5, C.c_double(0.1), *[p(o["cost"])] * 4, 16, p(o["cost"]),
                                                  p(o["status"]), p(o["cost"]), p(o["cost"]), None, None) == EINVAL   # no road
    assert lib.btrapz_destroy(None) == EINVAL
    # the two experimental schedules: an error in the shipped build, not a silent default
    if not lib.btrapz_build_has_experiments():
        for kw in (dict(queue=1), dict(start=1)):
            opt = native._options(**kw)
            bad = list(args(8, 10)); bad[2] = C.byref(opt)
            assert lib.btrapz_solve_batch_device(*bad) == EINVAL and b"EXPERIMENTS" in lib.btrapz_last_error(h)
            with pytest.raises(native.BtrapzError):
                solver.solve(db, sh, **kw)
    # the context still works, and gives the same answer
    o2 = solver.solve(db, sh)
    torch.cuda.synchronize()
    assert torch.equal(o2["ctrl"], good)


def test_per_candidate_failures_are_statuses_not_errors():
    """Infeasible bounds, t <= 0 and NaN inputs fail the candidate, not the call or its neighbours."""
    import torch
    from spectral_amd import layout as L
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0)
    batch, sh = synth.make_batch(12, 20, config=3)
    ref = solver.solve(solver.upload(batch), sh)
    torch.cuda.synchronize()
    ref_ctrl = ref["ctrl"].cpu().numpy().copy()
    b2 = batch.slice(0, 12)
    seg = b2.seg.copy()
    seg[L.F_UPP_BIAS, 1, 3] = seg[L.F_DOWN_BIAS, 1, 3] - 5.0     # crossing position bounds
    seg[L.F_T, 4, 0] = 0.0                                       # a segment without duration
    seg[L.F_T, 5, 7] = -1.0
    seg[L.F_X_BIAS, 7, 2] = np.nan                               # NaN in the reference line
    seg[L.F_DS_HI, 9, :] = -1.0                                  # velocity upper bound below the lower one
    b2.seg = seg
    b2.init = b2.init.copy(); b2.init[10, 1] = np.inf            # infinite initial speed
    o = solver.solve(solver.upload(b2), sh)
    torch.cuda.synchronize()
    st = o["status"].cpu().numpy(); cost = o["cost"].cpu().numpy(); ctrl = o["ctrl"].cpu().numpy()
    for b in (1, 4, 5, 9):
        assert st[b] == -3 and np.isinf(cost[b]), (b, st[b])    # BTRAPZ_PRIMAL_INFEASIBLE
    for b in (7, 10):
        assert st[b] <= 0 and np.isinf(cost[b]), (b, st[b])     # not solved, loses the arg-min
    for b in (0, 2, 3, 6, 8, 11):                               # neighbours in the same wavefronts are untouched
        assert st[b] == 1 and np.array_equal(ctrl[b], ref_ctrl[b])
    bi, bc = solver.argmin(o["cost"])
    assert int(bi.item()) in (0, 2, 3, 6, 8, 11)
