#!/usr/bin/env python3
"""Acceptance decisions of the reference's algorithm on every bundled input.

The reference accepts a solve iff OSQP's status_val is 1 or 2 (solve_3d.cc:1251-1253; trp_wrapper.cpp:191-200 then
returns a_cost instead of 1e11).  For every bundled corridor file x both variants, with src/weights.txt, this script
records what the oracle's OSQP port (the reference's settings: eps 1e-5, max_iter 5000) decides, what an exact method
finds (x* or "no solution"), and the least-squares row violation of the relaxed problem the product's rescue pass solves
(btrapz_options.elastic: every row relaxed in its own norm |g|).  The product's expected decision follows from the
last two:

    accept  <=>  x* exists  or  (the exact solve stalls and the least violation / |g| is <= elastic_tol = 0.0125)

For every input without a solution the table also holds the violation per class of rows, in the rows' own units, of
the product's least-violation answer (`class_violation`: position [m], velocity, acceleration, jerk rows) and -- where
the OSQP port returns an iterate -- of that iterate (`port_class_violation`), so that the two can be compared row
class by row class.

tests/test_gpu_acceptance.py holds the HIP path to these decisions; INTEGRATION.md prints the table.
Writes acceptance_table.json (and, with --markdown, the table for INTEGRATION.md on stdout)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

INPUTS = ["c1", "c2", "c3", "c4", "c4_2", "c5", "c6", "c7", "c7_7", "c7_10", "c_road_s1", "c_road_s1_2", "c_road_s1_3"]
ELASTIC_TOL = 0.0125


def classify(rec):
    """Agreement class of one row."""
    if rec["port_accepts"] == rec["hip_accepts"]:
        if rec["port_accepts"] and rec["port_status"] == 2:
            return "both accept; the reference returns its unconverged iterate, the product the optimum" if rec["exact_status"] > 0 \
                else "both accept; the reference returns an infeasible ADMM iterate, the product the least-violation solution"
        return "agree"
    if rec["hip_accepts"]:
        return "product accepts (the QP has an optimum; ADMM did not get there in 5000 iterations)" if rec["exact_status"] > 0 \
            else "product accepts (least violation %.4f |g| <= %.4f |g|; the reference's ADMM declared infeasibility)" % (rec["least_violation"], ELASTIC_TOL)
    return "reference accepts, product rejects"


def main():
    w = np.loadtxt(os.path.join(HERE, "inputs", "weights.txt"))
    p = O.params_from_weights(w)
    rows = []
    for name in INPUTS:
        for v in (0, 1):
            path = os.path.join(HERE, "inputs", name + ".txt")
            cost, S, ctrl, cubes, info = O.find_traj(v, path, None, p)
            rec = {"input": name, "variant": v, "segments": int(S), "port_status": int(info.status), "port_iters": int(info.iter),
                   "port_accepts": bool(info.status in (1, 2)), "exact_status": None, "exact_iters": None,
                   "least_violation": None, "class_violation": None, "port_class_violation": None,
                   "inconsistent_bounds": False, "port_vs_xstar_rel": None, "port_vs_xstar_abs": None}
            inp = O.ParsedInput(path)
            n, cb = O.pipeline(v, inp)
            if n >= 1:
                qp = O.AssembledQp(v, cb, p, inp)
                x, y, ie = qp.solve_exact()
                rec["exact_status"], rec["exact_iters"] = int(ie.status), int(ie.iter)
                # how far the reference's OWN answer (the OSQP port's stopping point: eps 1e-5, no polish) is from the
                # optimum it approximates, where both exist -- relative on the control points (north_star's measure) and
                # in the control points' own unit
                if info.status in (1, 2) and ie.status in (1, 2):
                    xp = np.asarray(ctrl, dtype=float)
                    rec["port_vs_xstar_rel"] = float(np.abs(xp - x).max() / np.abs(x).max())
                    rec["port_vs_xstar_abs"] = float(np.abs(xp - x).max())
                rec["inconsistent_bounds"] = bool((qp.l > qp.u + 1e-12).any())
                if ie.status not in (1, 2) and not rec["inconsistent_bounds"]:
                    xe, _, _, viol = qp.solve_elastic()
                    rec["least_violation"] = float(viol)
                    rec["class_violation"] = [round(v, 6) for v in qp.class_violations(xe)]
                    if info.status in (1, 2):
                        rec["port_class_violation"] = [round(v, 6) for v in qp.class_violations(qp.solve()[0])]
            rec["hip_accepts"] = bool(rec["exact_status"] in (1, 2) or
                                      (rec["least_violation"] is not None and rec["least_violation"] <= ELASTIC_TOL))
            rec["hip_status"] = 1 if rec["exact_status"] in (1, 2) else (2 if rec["hip_accepts"] else -3 if (rec["inconsistent_bounds"] or rec["least_violation"] is not None) else -5)
            rec["class"] = classify(rec)
            rows.append(rec)
    json.dump({"weights": [float(t) for t in w], "elastic_tol": ELASTIC_TOL, "rows": rows},
              open(os.path.join(HERE, "acceptance_table.json"), "w"), indent=1)
    if "--markdown" in sys.argv:
        print("| input | variant | S | OSQP port: status (iterations) | reference returns | port vs x*: rel (abs) | exact method | least violation / |g| (pos m, vel, acc, jerk) | product returns | |")
        print("|---|---|---|---|---|---|---|---|---|---|")
        for r in rows:
            print("| `%s.txt` | %s | %d | %d (%d) | %s | %s | %s | %s | %s | %s |" % (
                r["input"], "trapezoid" if r["variant"] == 0 else "cuboid", r["segments"], r["port_status"], r["port_iters"],
                "trajectory" if r["port_accepts"] else "`1e11`",
                "—" if r["port_vs_xstar_rel"] is None else "%.1e (%.2g)" % (r["port_vs_xstar_rel"], r["port_vs_xstar_abs"]),
                "x* in %d iterations" % r["exact_iters"] if r["exact_status"] in (1, 2) else ("`l > u` rows" if r["inconsistent_bounds"] else "no solution"),
                "—" if r["least_violation"] is None else "%.4f (%s)%s" % (
                    r["least_violation"], ", ".join("%.3g" % v for v in r["class_violation"]),
                    "" if not r["port_class_violation"] else "; the port's iterate: " + ", ".join("%.3g" % v for v in r["port_class_violation"])),
                "trajectory (status %d)" % r["hip_status"] if r["hip_accepts"] else "`1e11`", r["class"]))


if __name__ == "__main__":
    main()
