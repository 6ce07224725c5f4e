#!/usr/bin/env python3
"""Which weights wrote which saved trajectory?  The search behind the reference-generated goldens.

The reference saved ~58 trajectory files (src/s*_3d*.txt, ..., copied to ref_outputs/) without recording the
weights of the run; src/all_weights.txt holds the Optuna trial log (288 lines, 204 numeric rows) and src/weights.txt the row
the harness loads.  This script runs the oracle's find_traj restatement (parser -> corridor pipeline -> assembly ->
OSQP port at the reference's settings -> Bernstein sampling) for EVERY bundled input x BOTH variants x EVERY weight
row (+ weights.txt), and the exact optimum x* of the same QP beside it, and compares all seven columns with every
saved file of the same row count and first row.  Result: weight_search.json, one record per saved file with the best
match overall and per axis (the axes are independent QPs, but OSQP's stopping point couples them, so an s-axis match
can come from a row whose l weights differ from the saved run's).

    python tests/golden/find_weights.py [--jobs 8]        # ~10 min on 8 cores

tests/test_reference_goldens.py pins every (file, input, variant, row) this search matches to print precision and
asserts that the committed JSON says "no row matches" for the rest.
"""
import argparse
import json
import os
import sys
import warnings
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
PRINT = 5.0e-4 + 2e-5
INPUTS = ["c1", "c2", "c3", "c4", "c4_2", "c5", "c6", "c7", "c7_7", "c7_10", "c_road_s1", "c_road_s1_2", "c_road_s1_3",
          "bounds"]


def weight_rows():
    rows = []
    for line in open(os.path.join(HERE, "inputs", "all_weights.txt")):
        try:
            v = [float(t) for t in line.split()]
        except ValueError:
            continue
        if len(v) >= 10:
            rows.append(v[:10])
    rows.append([float(v) for v in np.loadtxt(os.path.join(HERE, "inputs", "weights.txt"))[:10]])   # index 288
    return rows


def saved_outputs():
    out = {}
    d = os.path.join(HERE, "ref_outputs")
    for f in sorted(os.listdir(d)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a = np.loadtxt(os.path.join(d, f))
        if a.ndim == 2 and a.shape[1] == 7:
            out[f] = a
    return out


def work(job):
    name, variant = job
    from oracle import oracle as O
    saved = saved_outputs()
    path = os.path.join(HERE, "inputs", name + ".txt")
    try:
        inp = O.ParsedInput(path)
        n, cubes = O.pipeline(variant, inp)
    except Exception:
        return []
    if n < 1 or n > 64:
        return []
    rows = 1 + sum(int(c.t / inp.delta) for c in cubes)
    first = np.array([0.0, inp.init_s[0], inp.init_l[0], inp.init_s[1], inp.init_l[1], inp.init_s[2], inp.init_l[2]])
    cands = {k: a for k, a in saved.items() if a.shape[0] == rows and np.abs(a[0] - first).max() <= PRINT}
    res = []
    if not cands:
        return res
    for wi, w in enumerate(weight_rows()):
        p = O.params_from_weights(w)
        try:
            qp = O.AssembledQp(variant, cubes, p, inp)
        except Exception:
            continue
        for mode in ("osqp", "exact"):
            x, y, info = qp.solve() if mode == "osqp" else qp.solve_exact()
            if info.status not in (1, 2):
                continue
            rc, s = O.sample(cubes, inp.delta, x, inp.init_s, inp.init_l)
            if rc != 0 or len(s[0]) != rows:
                continue
            traj = np.stack([np.arange(rows) * inp.delta, s[0], s[3], s[1], s[4], s[2], s[5]], 1)
            for k, a in cands.items():
                d = np.abs(a - traj)
                res.append((k, name, variant, wi, mode, int(info.status), int(info.iter), float(d.max()),
                            float(d[:, [1, 3, 5]].max()), float(d[:, [2, 4, 6]].max())))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=len(os.sched_getaffinity(0)))
    a = ap.parse_args()
    jobs = [(n, v) for n in INPUTS for v in (0, 1)]
    with ProcessPoolExecutor(a.jobs) as ex:
        allres = [r for part in ex.map(work, jobs) for r in part]
    saved = saved_outputs()
    table = {}
    for f in saved:
        rec = {"rows": int(saved[f].shape[0])}
        mine = [r for r in allres if r[0] == f]
        for label, col in (("all", 7), ("s", 8), ("l", 9)):
            if not mine:
                rec[label] = None
                continue
            b = min(mine, key=lambda r: (r[col], r[4] != "osqp", r[3]))
            rec[label] = {"input": b[1], "variant": b[2], "weight_row": b[3], "mode": b[4], "status": b[5], "iters": b[6],
                          "max_abs_diff": b[col], "matches_to_print_precision": bool(b[col] <= PRINT)}
            # every row that matches this label (different rows can share the weights that matter)
            rec[label]["matching_rows"] = sorted({(r[1], r[2], r[3], r[4]) for r in mine if r[col] <= PRINT})[:12]
        table[f] = rec
    json.dump({"print_tolerance": PRINT, "weight_rows": len(weight_rows()), "inputs": INPUTS, "files": table},
              open(os.path.join(HERE, "weight_search.json"), "w"), indent=1, sort_keys=True)
    for f, rec in sorted(table.items()):
        fmt = lambda r: "-" if r is None else "%s/%d row %d %s %.4f%s" % (r["input"], r["variant"], r["weight_row"], r["mode"],
                                                                        r["max_abs_diff"], " *" if r["matches_to_print_precision"] else "")
        print("%-22s all: %-38s s: %-38s l: %s" % (f, fmt(rec["all"]), fmt(rec["s"]), fmt(rec["l"])))


if __name__ == "__main__":
    main()
