#!/usr/bin/env python3
"""Can scenario_1 (src/c1.txt) be pinned by a file the reference itself wrote?  The hypotheses of VERDICT r5 item 2, each
run to a yes / no with its numbers -> tests/golden/scenario1_pin_search.json.

The reference saved trajectories without their weights, and its harness REWRITES its corridor file at every replanning
step (src/cart_frenet.py:385-386 opens .../c_road_s1_3.txt for writing) -- a saved output's input survives only by luck.
`find_weights.py` (round 3) already ran every bundled input x both variants x the 204 logged weight rows of
src/all_weights.txt (+ weights.txt) through the oracle's whole find_traj, OSQP port and exact optimum: for c1 the row
count and first row match seven saved files, and no row reproduces any of them (weight_search.json).  This script adds
what that search held fixed:

  H1  the OSQP port's stopping point under the logged rows              (from weight_search.json: the search's `osqp` mode)
  H2  src/slt_3d.txt / src/s1_slt_3d.txt as output of the OLD libbtrapz.so, whose compiled-in input is c1.txt
      (strings libbtrapz.so; its main is the stale src/trp_extend_3d.cc:32-94, another grammar: weights IN the file, one
      axis, two obstacles without l bounds)
  H3  the piecewise-linear reference of a segment read elsewhere than solve_3d.cc:1159-1166 reads it (x_ref[10k],
      x_ref[10k+1]): at the segment's real first knot; one knot later; one knot earlier
  H4  another knot spacing in the reference's slope (delta 0.1 hard-coded as in CorridorSplit, solve_3d.cc:753,764: here
      the files' delta IS 0.1, so this is H3 with the slope over TEN knots, i.e. the segment's chord)

    python tests/golden/pin_scenario1.py          # ~2 min on 8 cores

A hypothesis "pins" a file when all seven columns agree to print precision (5.2e-4) for one weight row.
"""
import json
import os
import sys
import warnings
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from find_weights import PRINT, weight_rows  # noqa: E402

# the saved files whose row count and first row c1.txt reproduces (weight_search.json), per variant
C1_FILES = {0: ["s1_slt_3d_30.txt", "s1_slt_3d_31.txt", "s1_slt_3d_500.txt"],
            1: ["s1_cub_3d_3.txt", "s1_cub_3d_4.txt", "s1_cub_3d_30.txt", "s1_cub_3d_31.txt", "s1_slt_3d_4.txt"]}
REF_MODES = ("as_reference", "at_segment_start", "one_knot_later", "one_knot_earlier", "chord_over_ten_knots")


class Refs:
    """An input whose x_ref / y_ref arrays are rewritten so that the reference's own indexing (entries 10k and 10k+1,
    solve_3d.cc:1159-1166) reads what the hypothesis says it should."""

    def __init__(self, inp, cubes, mode):
        for k in ("N", "delta", "init_s", "init_l", "ds_ref", "dl_ref", "dds", "ddds", "ddl", "dddl", "dx_bounds", "dy_bounds"):
            setattr(self, k, getattr(inp, k))
        N = inp.N
        out = []
        for ref in (inp.x_ref, inp.y_ref):
            r = ref.copy()
            g = lambda i: ref[min(max(i, 0), N - 1)]
            for k, c in enumerate(cubes):
                i0 = 10 * k
                if i0 + 1 >= N:
                    break
                if mode == "at_segment_start":
                    a, b = g(c.beg_t), g(c.beg_t + 1)
                elif mode == "one_knot_later":
                    a, b = g(i0 + 1), g(i0 + 2)
                elif mode == "one_knot_earlier":
                    a, b = g(i0 - 1), g(i0)
                elif mode == "chord_over_ten_knots":
                    a = g(i0); b = a + (g(i0 + 10) - a) / 10.0
                else:
                    a, b = g(i0), g(i0 + 1)
                if i0 != N - 1:
                    r[i0] = a
                if i0 + 1 != N - 1:        # (the end term reads x_ref[N-1]: solve_3d.cc:268,315 -- left alone)
                    r[i0 + 1] = b
            out.append(r)
        self.x_ref, self.y_ref = out


def work(job):
    variant, mode = job
    from oracle import oracle as O
    inp = O.ParsedInput(os.path.join(HERE, "inputs", "c1.txt"))
    n, cubes = O.pipeline(variant, inp)
    saved = {}
    for f in C1_FILES[variant]:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            saved[f] = np.loadtxt(os.path.join(HERE, "ref_outputs", f))
    src = Refs(inp, cubes, mode)
    best = {f: {"all": (9e9, -1), "s": (9e9, -1), "l": (9e9, -1)} for f in saved}
    for wi, w in enumerate(weight_rows()):
        qp = O.AssembledQp(variant, cubes, O.params_from_weights(w), src)
        x, y, info = qp.solve()
        if info.status not in (1, 2):
            continue
        rc, s = O.sample(cubes, inp.delta, x, inp.init_s, inp.init_l)
        traj = np.stack([np.arange(len(s[0])) * inp.delta, s[0], s[3], s[1], s[4], s[2], s[5]], 1)
        for f, a in saved.items():
            if a.shape != traj.shape:
                continue
            d = np.abs(a - traj)
            for label, v in (("all", d.max()), ("s", d[:, [1, 3, 5]].max()), ("l", d[:, [2, 4, 6]].max())):
                if v < best[f][label][0]:
                    best[f][label] = (float(v), wi)
    return variant, mode, best


def main():
    ws = json.load(open(os.path.join(HERE, "weight_search.json")))
    rec = {"print_tolerance": PRINT, "hypotheses": {}}
    # H1: what the round-3 search found for the c1-derived files (its `osqp` rows are the port's stopping point)
    h1 = {}
    for v, files in C1_FILES.items():
        for f in files:
            r = ws["files"][f]
            h1[f] = {k: (None if r[k] is None else {"input": r[k]["input"], "variant": r[k]["variant"], "weight_row": r[k]["weight_row"],
                                                    "mode": r[k]["mode"], "max_abs_diff": r[k]["max_abs_diff"]}) for k in ("all", "s", "l")}
    rec["hypotheses"]["H1_port_under_logged_rows"] = {"pinned": [f for f, r in h1.items() if r["all"] and r["all"]["max_abs_diff"] <= PRINT], "files": h1}
    # H2: the old library's pair.  First row of a trajectory file = the initial state of its input (trp_wrapper.cpp:288-301).
    from oracle import oracle as O
    inp = O.ParsedInput(os.path.join(HERE, "inputs", "c1.txt"))
    first = [0.0, inp.init_s[0], inp.init_l[0], inp.init_s[1], inp.init_l[1], inp.init_s[2], inp.init_l[2]]
    h2 = {"c1_first_row": first}
    for f in ("slt_3d.txt", "s1_slt_3d.txt"):
        a = np.loadtxt(os.path.join(HERE, "ref_outputs", f))
        rows = {v: 1 + sum(int(c.t / inp.delta) for c in O.pipeline(v, inp)[1]) for v in (0, 1)}
        h2[f] = {"first_row": a[0].tolist(), "rows": int(a.shape[0]), "first_row_is_c1s": bool(np.abs(a[0] - first).max() <= PRINT),
                 "rows_from_c1_trapezoid": rows[0], "rows_from_c1_cuboid": rows[1]}
    h2["pinned"] = []
    h2["why_not"] = ("slt_3d.txt starts at ds = 5.000 where c1.txt says 7: it was written from another c1.txt.  s1_slt_3d.txt starts "
                     "at c1's initial state but has 75 rows where today's pipeline cuts c1 into 70 (trapezoid) / 74 (cuboid): it was "
                     "written by another pipeline.  The old main (trp_extend_3d.cc:32-94) reads weights from the file and one axis only: "
                     "c1.txt's token stream under that grammar is not a problem statement.")
    rec["hypotheses"]["H2_old_library_pair"] = h2
    # H3 / H4
    jobs = [(v, m) for v in (0, 1) for m in REF_MODES]
    with ProcessPoolExecutor(min(len(jobs), len(os.sched_getaffinity(0)))) as ex:
        res = list(ex.map(work, jobs))
    h3 = {}
    for v, mode, best in res:
        for f, b in best.items():
            h3.setdefault(mode, {})[f] = {k: {"max_abs_diff": b[k][0], "weight_row": b[k][1]} for k in ("all", "s", "l")}
    for mode in REF_MODES:
        rec["hypotheses"]["H3_refs_" + mode if mode != "chord_over_ten_knots" else "H4_refs_" + mode] = {
            "pinned": [f for f, r in h3[mode].items() if r["all"]["max_abs_diff"] <= PRINT], "files": h3[mode]}
    rec["pinned_by_any_hypothesis"] = sorted({f for h in rec["hypotheses"].values() for f in h["pinned"]})
    json.dump(rec, open(os.path.join(HERE, "scenario1_pin_search.json"), "w"), indent=1, sort_keys=True)
    for name, h in rec["hypotheses"].items():
        print(name, "pinned:", h["pinned"])
        for f, r in sorted(h.get("files", {}).items()):
            print("   %-20s" % f, "  ".join("%s %.4f (row %s)" % (k, r[k]["max_abs_diff"], r[k].get("weight_row")) if r[k] else k + " -" for k in ("all", "s", "l")))


if __name__ == "__main__":
    main()
