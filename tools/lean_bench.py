#!/usr/bin/env python3
"""btrapz_options.lean (the solve kernel at two wavefronts per SIMD, btrapz_lean.hip) against the one-wavefront packed
form on the bench batches: time of the whole solve call by HIP events (one launch and the two launches of cap_iter),
agreement of the results (status, iterations, control points), and a slice of every batch against the oracle's exact
solve.  One JSON object on stdout.

    python tools/lean_bench.py [--batch 65536] [--oracle 96]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(solver, dev, db, sh, reps, **kw):
    import torch
    for _ in range(2):
        o = solver.solve(db, sh, split=-1, **kw)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        o = solver.solve(db, sh, split=-1, **kw)
    e1.record(); torch.cuda.synchronize(dev)
    res = {k: o[k].cpu().numpy().copy() for k in ("ctrl", "cost", "status", "iters")}
    return e0.elapsed_time(e1) / reps, res, solver.ctx.last_solve_form()


def main(argv=None):
    import torch
    from spectral_amd import synth
    from spectral_amd.solver import BatchSolver
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--oracle", type=int, default=96, help="candidates per batch compared with the oracle's exact solve")
    ap.add_argument("--cases", default="0,1,2,3")
    ap.add_argument("--caps", default="", help="further hand-over points of the lean two-launch solve to time, e.g. 3,4,8 (strong-scaling shard sizes)")
    a = ap.parse_args(argv)
    solver = BatchSolver(0)
    dev = torch.device("cuda:0")
    out = {}
    cases = [("scenario1 x 20 trapezoid", lambda: synth.make_scenario1_batch(a.batch, 20, 0)),
             ("generic x 20 trapezoid", lambda: synth.make_batch(a.batch, 20, config=3)),
             ("scenario1 x 20 cuboid", lambda: synth.make_scenario1_batch(a.batch, 20, 1)),
             ("generic x 10 trapezoid", lambda: synth.make_batch(a.batch, 10, config=2))]
    for ci in [int(x) for x in a.cases.split(",")]:
        label, make = cases[ci]
        batch, sh = make()
        db = solver.upload(batch)
        rec = {}
        runs = {}
        for name, kw in (("packed_one_launch", dict(lean=-1, cap_iter=-1)), ("lean_one_launch", dict(lean=1, cap_iter=-1)),
                         ("packed_two_launches", dict(lean=-1, cap_iter=6)), ("lean_two_launches", dict(lean=1, cap_iter=6))):
            ms, res, form = timed(solver, dev, db, sh, a.reps, **kw)
            runs[name] = res
            rec[name] = {"solve_ms": ms, "form": form, "mean_iterations": float(res["iters"].mean() + 1),
                         "solved_fraction": float((res["status"] > 0).mean())}
        for cap in [int(x) for x in a.caps.split(",") if x]:
            ms, res, form = timed(solver, dev, db, sh, a.reps, lean=1, cap_iter=cap)
            rec["lean_two_launches_cap%d" % cap] = {"solve_ms": ms, "form": form}
        ref = runs["packed_one_launch"]
        ok = ref["status"] > 0
        for name in ("lean_one_launch", "lean_two_launches", "packed_two_launches"):
            r = runs[name]
            both = ok & (r["status"] > 0)
            scale = np.abs(ref["ctrl"][both]).max(axis=1, keepdims=True)
            rec[name]["status_differs"] = int((r["status"] != ref["status"]).sum())
            rec[name]["accept_differs"] = int(((r["status"] > 0) != ok).sum())
            rec[name]["iters_differs"] = int((r["iters"] != ref["iters"]).sum())
            rec[name]["max_rel_ctrl_vs_packed"] = float((np.abs(r["ctrl"][both] - ref["ctrl"][both]) / scale).max()) if both.any() else None
        l1, l2 = runs["lean_one_launch"], runs["lean_two_launches"]
        okl = l1["status"] > 0
        rec["lean_two_launches"]["bit_identical_to_lean_one_launch"] = bool(
            np.array_equal(l1["status"], l2["status"]) and np.array_equal(l1["iters"], l2["iters"]) and
            np.array_equal(l1["ctrl"][okl], l2["ctrl"][okl]) and np.array_equal(l1["cost"], l2["cost"]))
        rec["speedup_one_launch"] = rec["packed_one_launch"]["solve_ms"] / rec["lean_one_launch"]["solve_ms"]
        rec["speedup_two_launches"] = rec["packed_two_launches"]["solve_ms"] / rec["lean_two_launches"]["solve_ms"]
        if a.oracle > 0:
            from oracle import oracle as O
            n = min(a.oracle, a.batch)
            xs, obj, st, _ = O.batch_solve(batch, sh, 0, n, exact=True)
            good = (st[:n] > 0) & (l1["status"][:n] > 0)
            rel = np.abs(l1["ctrl"][:n][good] - xs[good]).max(axis=1) / np.abs(xs[good]).max(axis=1)
            rec["lean_vs_oracle"] = {"candidates": int(n), "both_solved": int(good.sum()),
                                     "accept_differs": int(((st[:n] > 0) != (l1["status"][:n] > 0)).sum()),
                                     "max_rel_ctrl": float(rel.max()) if good.any() else None}
        out[label] = rec
        print(label, json.dumps(rec), file=sys.stderr, flush=True)
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
