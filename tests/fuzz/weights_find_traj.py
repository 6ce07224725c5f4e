"""find_traj over the reference's WEIGHT space: the reference's real caller is an Optuna objective
that draws each of the ten Params weights from U(0, 50) (src/trp_wrapper.py:56-97); its trial log all_weights.txt holds
204 ten-column rows.  For every input x variant x weight row, and for both single-candidate kernels (BTRAPZ_SPLIT=0/1):
accept decision and control points of the HIP path against the oracle's exact solve (and, with the rescue pass on, the
oracle's relaxed solve).

    python tests/fuzz/weights_find_traj.py [SEED] [RANDOM_ROWS] [ELASTIC] [INPUTS,comma] [WORKERS]     # needs a GPU

Rows: the 204 of tests/golden/inputs/all_weights.txt, RANDOM_ROWS seeded draws from U(0,50)^10, and the degenerate rows
(a *_ref or end weight exactly 0: P only semidefinite on that axis -- there the optimum need not be unique, so the
objective value, the status and feasibility are compared, control points only when the oracle's KKT matrix says the
optimum is unique).  Every 8th call also goes through CDLL(libtrp.so / libcub.so).find_traj with the file path in the
environment: same cost, bit for bit.  Oracle solves run first, in forked workers, before anything touches the GPU."""
import ctypes as C
import os
import sys
import time
import multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 0
NRAND = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ELASTIC = sys.argv[3] if len(sys.argv) > 3 else "1"
INPUTS = (sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] not in ("", "-") else "c1,c2,c_road_s1_3,c7_7").split(",")
WORKERS = int(sys.argv[5]) if len(sys.argv) > 5 else min(14, os.cpu_count() or 4)
os.environ["BTRAPZ_ELASTIC"] = ELASTIC
GOLD = os.path.join(ROOT, "tests", "golden", "inputs")
SENTINEL = 100000000000.0


def weight_rows(seed, nrand):
    rows = []
    for line in open(os.path.join(GOLD, "all_weights.txt")):
        t = line.split()
        if len(t) == 10:
            rows.append(("file", [float(v) for v in t]))
    every = int(os.environ.get("WFT_FILE_ROW_STRIDE", "1"))      # (the driver suite's slice: every n-th row of the trial log)
    rows = rows[::every]
    rng = np.random.default_rng(1000 + seed)
    for r in rng.uniform(0.0, 50.0, (nrand, 10)):
        rows.append(("rand", [float(v) for v in r]))
    base = [35.73, 41.61, 25.57, 41.59, 0.12, 10.04, 0.71, 14.3, 7.27, 32.13]
    deg = []
    for zero in ([4], [6], [8], [9], [4, 8], [6, 9], [4, 6, 8, 9], [4, 5], [6, 7], [0, 1], [2, 3], [0, 1, 2, 3], list(range(10))):
        w = list(base)
        for z in zero:
            w[z] = 0.0
        deg.append(("zero%s" % "_".join(map(str, zero)), w))
    for r in rng.uniform(0.0, 50.0, (6, 10)):      # random rows with one *_ref weight zeroed
        w = [float(v) for v in r]; w[int(rng.choice([4, 6]))] = 0.0
        deg.append(("randzero", w))
    # extremes of the box
    for lo, hi in ((1e-3, 50.0), (50.0, 1e-3), (1e-6, 1e-6), (50.0, 50.0)):
        deg.append(("corner", [lo, lo, lo, lo, hi, hi, hi, hi, lo, hi]))
    return rows + deg


def oracle_job(args):
    name, variant, rows = args
    from oracle import oracle as O
    path = os.path.join(GOLD, name + ".txt")
    inp = O.ParsedInput(path)
    n, cubes = O.pipeline(variant, inp)
    out = []
    for kind, w in rows:
        rec = dict(accept=False, x=None, obj=None, status=None, iters=None, elastic=False, viol=None, unique=True, skip=False)
        try:
            if 1 <= n <= 64 and all(c.t > 0 for c in cubes):
                qp = O.AssembledQp(variant, cubes, O.params_from_weights(w), inp)
                x, _, info = qp.solve_exact()
                rec["status"], rec["iters"] = info.status, info.iter
                ok = info.status in (1, 2)
                if not ok and ELASTIC == "1" and not (qp.l > qp.u + 1e-12).any() and np.isfinite(qp.l).all() and np.isfinite(qp.u).all():
                    x, _, info, viol = qp.solve_elastic()
                    rec["elastic"], rec["viol"] = True, float(viol)
                    rec["skip"] = abs(viol - 0.0125) < 0.0005
                    ok = info.status in (1, 2) and viol <= 0.0125
                    rec["estatus"] = info.status
                if ok:
                    rc, smp = O.sample(cubes, inp.delta, x, inp.init_s, inp.init_l)
                    ok = rc == 0
                if ok:
                    P, A = qp.dense()[0], qp.dense()[1]
                    rec["x"] = np.array(x, dtype=float)
                    rec["obj"] = float(0.5 * x @ P @ x + qp.q @ x)
                    if kind.startswith("zero") or kind in ("randzero", "corner"):
                        # unique optimum <=> P positive definite on the null space of the active rows
                        Ax = A @ x
                        act = (np.abs(Ax - qp.l) <= 1e-7 * (1 + np.abs(qp.l))) | (np.abs(Ax - qp.u) <= 1e-7 * (1 + np.abs(qp.u)))
                        Aa = A[act]
                        if Aa.shape[0] < A.shape[1]:
                            u, s, vt = np.linalg.svd(Aa, full_matrices=True)
                            r = int((s > 1e-9 * max(1.0, s.max() if s.size else 1.0)).sum())
                            Z = vt[r:].T
                            if Z.shape[1]:
                                ev = np.linalg.eigvalsh(Z.T @ P @ Z)
                                rec["unique"] = bool(ev.min() > 1e-7 * max(1.0, ev.max()))
                rec["accept"] = bool(ok)
        except Exception as e:
            rec["error"] = repr(e)[:200]
        out.append(rec)
    return name, variant, n, out


_qp_cache = {}


def special_qp(name, variant, w):
    from oracle import oracle as O
    key = (name, variant)
    if key not in _qp_cache:
        inp = O.ParsedInput(os.path.join(GOLD, name + ".txt"))
        _qp_cache[key] = (inp, O.pipeline(variant, inp)[1])
    inp, cubes = _qp_cache[key]
    return O.AssembledQp(variant, cubes, O.params_from_weights(w), inp)


def main():
    rows = weight_rows(SEED, NRAND)
    jobs = [(name, v, rows) for name in INPUTS for v in (0, 1)]
    t0 = time.time()
    # split each (input, variant) into chunks so the pool stays busy
    chunks = []
    CH = 32
    for name, v, rr in jobs:
        for i in range(0, len(rr), CH):
            chunks.append((name, v, rr[i:i + CH]))
    with mp.get_context("fork").Pool(WORKERS) as pool:
        res = pool.map(oracle_job, chunks, chunksize=1)
    want = {}
    segs = {}
    for (name, v, rr), (_, _, n, out) in zip(chunks, res):
        want.setdefault((name, v), []).extend(out); segs[(name, v)] = n
    print("oracle: %d solves in %.1f s on %d workers" % (sum(len(v) for v in want.values()), time.time() - t0, WORKERS), flush=True)

    from spectral_amd import knots, native, trp_wrapper
    tally = dict(calls=0, agree=0, accepted=0, rejected=0, decisions_apart=0, xstar_beyond=0, skipped=0, status2=0, cdll_calls=0, cdll_cost_differs=0,
                 degenerate_rows=0, degenerate_nonunique=0, obj_beyond=0, rescued=0)
    worst = dict(plain=0.0, status2=0.0, rescued=0.0, obj=0.0, degenerate_obj=0.0)
    worst_at = {}
    iters_hist = {}
    t0 = time.time()
    tmp = "/tmp/wft_%d_" % os.getpid()
    for split in ("0", "1"):
        os.environ["BTRAPZ_SPLIT"] = split
        for name in INPUTS:
            path = os.path.join(GOLD, name + ".txt")
            kb = knots.parse_corridor_file(path)
            for variant in (0, 1):
                lib = C.CDLL(os.path.join(native.LIB_DIR, "libtrp.so" if variant == 0 else "libcub.so"))
                lib.find_traj.argtypes = (C.POINTER(trp_wrapper.Params),); lib.find_traj.restype = C.c_double
                for i, ((kind, w), rec) in enumerate(zip(rows, want[(name, variant)])):
                    params = native.CParams(*w, 7)
                    cost, traj, ctrl = native.find_traj_mem(variant, params, kb, 0)
                    st, cv = native.find_traj_last_status()
                    it = native.lib().btrapz_find_traj_last_iterations()
                    tally["calls"] += 1
                    if i % 8 == 0:
                        os.environ["BTRAPZ_INPUT"] = path; os.environ["BTRAPZ_OUTPUT_PREFIX"] = tmp
                        c2 = lib.find_traj(trp_wrapper.Params(*w, 7))
                        tally["cdll_calls"] += 1
                        if not (c2 == cost):
                            tally["cdll_cost_differs"] += 1; print("CDLL", name, variant, kind, i, cost, c2, flush=True)
                    if rec.get("skip"):
                        tally["skipped"] += 1; continue
                    got = cost != SENTINEL
                    if got != rec["accept"]:
                        tally["decisions_apart"] += 1
                        print("DECISION split=%s %s v%d row %d (%s) hip %s (status %d, iters %d) oracle %s (status %s, iters %s, elastic %s viol %s) w=%s" %
                              (split, name, variant, i, kind, got, st, it, rec["accept"], rec["status"], rec["iters"], rec["elastic"], rec["viol"], [round(v, 3) for v in w]), flush=True)
                        continue
                    tally["agree"] += 1
                    if not got:
                        tally["rejected"] += 1; continue
                    tally["accepted"] += 1
                    iters_hist[it] = iters_hist.get(it, 0) + 1
                    x = rec["x"]
                    err = np.abs(ctrl - x).max() / max(1e-300, np.abs(x).max())
                    special = kind.startswith("zero") or kind in ("randzero", "corner")
                    # the objective the product would report: compare through the oracle's P, q
                    if special:
                        tally["degenerate_rows"] += 1
                    if rec["elastic"]:
                        tally["rescued"] += 1
                        tol = 1e-3       # (the relaxed problem carries a 1 / delta = 1e8 penalty: its own conditioning, not x*'s)
                        key = "rescued"
                    elif st == 2:
                        tally["status2"] += 1; tol = 1e-4; key = "status2"
                    else:
                        tol = 1e-5; key = "plain"
                    if special and rec["elastic"] and not rec["unique"]:
                        tally["degenerate_nonunique"] += 1
                        continue       # (a rescued problem whose objective is identically zero: any least-violation point will do -- the decision is what is compared)
                    if special and not rec["elastic"]:
                        # a *_ref / end weight of 0 (P semidefinite), or every weight at 1e-6 (the objective below the
                        # solver's absolute tolerance): objective value and feasibility, not control points
                        qp = special_qp(name, variant, w)
                        P, A = qp.dense()[0], qp.dense()[1]
                        f_h, f_o = float(0.5 * ctrl @ P @ ctrl + qp.q @ ctrl), rec["obj"]
                        Ax = A @ ctrl
                        viol = float((np.maximum(np.maximum(qp.l - Ax, Ax - qp.u), 0.0) / (1.0 + np.minimum(np.maximum(np.abs(qp.l), np.abs(qp.u)), 1e9))).max())
                        dobj = abs(f_h - f_o) / (1e-9 + abs(f_o))
                        worst["degenerate_obj"] = max(worst["degenerate_obj"], dobj); worst["degenerate_viol"] = max(worst.get("degenerate_viol", 0.0), viol)
                        worst["degenerate_ctrl"] = max(worst.get("degenerate_ctrl", 0.0), err)
                        if not rec["unique"]:
                            tally["degenerate_nonunique"] += 1
                        if dobj > 1e-6 or viol > 1e-7:
                            tally["obj_beyond"] += 1
                            print("OBJECTIVE split=%s %s v%d row %d (%s) hip %.12g oracle %.12g rel %.2e violation %.2e" % (split, name, variant, i, kind, f_h, f_o, dobj, viol), flush=True)
                        continue
                    if err > worst[key]:
                        worst[key] = err; worst_at[key] = (split, name, variant, i, kind)
                    if not (ctrl.shape == x.shape and err <= tol):
                        tally["xstar_beyond"] += 1
                        print("XSTAR split=%s %s v%d row %d (%s) err %.3e (tol %.0e) hip status %d iters %d oracle iters %s w=%s" %
                              (split, name, variant, i, kind, err, tol, st, it, rec["iters"], [round(v, 3) for v in w]), flush=True)
            print("  split=%s %s done: %d calls, %.1f s, decisions apart %d, beyond tolerance %d" %
                  (split, name, tally["calls"], time.time() - t0, tally["decisions_apart"], tally["xstar_beyond"]), flush=True)
    nfile = sum(1 for k, _ in rows if k == "file")
    print("weights sweep (find_traj): inputs", INPUTS, "rows", len(rows), "(%d file + %d random + %d degenerate/corner)" % (nfile, NRAND, len(rows) - nfile - NRAND),
          "elastic", ELASTIC)
    print("  ", tally)
    print("   worst relative deviation from x*:", {k: "%.2e" % v for k, v in worst.items()}, "at", worst_at)
    print("   iterations histogram (accepted):", dict(sorted(iters_hist.items())))
    print("   segments:", {"%s/v%d" % k: v for k, v in segs.items()})
    return 0 if tally["decisions_apart"] == 0 and tally["xstar_beyond"] == 0 and tally["cdll_cost_differs"] == 0 and tally["obj_beyond"] == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
