#!/usr/bin/env python3
"""Golden vectors for the obstacle-prism -> bounds generator, produced by the REFERENCE'S OWN code.

`Car`, `get_bounds`, `lineFromPoints` and `delete_multiple_element` are taken out of /root/reference/src/cart_frenet.py
with `ast` at run time (the module itself cannot be imported: commonroad is absent and it loads a scenario at import)
and executed with the harness's globals (cart_frenet.py:54-64).  Nothing of that source is stored here: the output,
prism_goldens.json, holds scene inputs and the bounds the reference returned.  Scenes: the harness's own two-car
construction (cart_frenet.py:1536-1546) at several relative positions, and seeded random scenes of 1-3 cars.

    python tests/golden/make_prism_goldens.py          # needs /root/reference (build container only)
"""
import ast
import copy
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/src/cart_frenet.py"
GLOBALS = dict(s_u_l=50.0, s_l_l=0.0, d_u_l=8.0, d_l_l=-2.0, num_of_knots=71, homotopy="yield")


def reference_functions():
    tree = ast.parse(open(SRC).read())
    want = {"Car", "get_bounds", "lineFromPoints", "delete_multiple_element"}
    nodes = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in want]
    ns = dict(np=np, copy=copy, cur_s=[], cur_l=[], **GLOBALS)
    exec(compile(ast.Module(body=nodes, type_ignores=[]), SRC, "exec"), ns)
    return ns["Car"], ns["get_bounds"]


def main():
    Car, get_bounds = reference_functions()
    rng = np.random.default_rng(20260403)
    scenes = []
    # the harness's construction: car 2 (ref 11, 4.5 m/s) then car 1 (ref 10, 6 m/s), 4 s prisms; t0 = 0 when ahead of
    # the ego, else |ds| / 5 (cart_frenet.py:1536-1546)
    for ds2, l2, ds1, l1 in [(12.0, 3.5, 25.0, 0.5), (8.0, 3.5, -6.0, 0.5), (-4.0, 3.5, 18.0, 0.5), (30.0, 5.0, 10.0, 1.0),
                             (15.0, 3.5, 15.0, 0.4), (-10.0, 6.0, -3.0, 0.0)]:
        c2 = dict(centre=(abs(ds2), l2, 0 if ds2 > 0 else abs(ds2) / 5.0), vel_s=4.5, vel_l=0, time=4.0, ref=11)
        c1 = dict(centre=(ds1, l1, 0 if ds1 > 0 else abs(ds1) / 5.0), vel_s=6.0, vel_l=0.0, time=4.0, ref=10)
        scenes.append([c2, c1])
    for n in range(60):
        cars = []
        for r in range(int(rng.integers(1, 4))):
            ahead = rng.uniform() < 0.6
            cars.append(dict(centre=(round(float(rng.uniform(5, 40)), 1), round(float(rng.uniform(-1.0, 7.0)), 2),
                                     0 if ahead else round(float(rng.uniform(0.2, 3.0)), 1)),
                             vel_s=round(float(rng.uniform(1, 8)), 1), vel_l=round(float(rng.choice([0.0, 0.0, 0.2, -0.2])), 1),
                             time=float(rng.choice([3.0, 4.0])), ref=10 + r))
        scenes.append(cars)
    out = []
    for cars in scenes:
        Car._lateral = []
        for c in cars:
            Car(tuple(c["centre"]), vel_s=c["vel_s"], vel_l=c["vel_l"], time=c["time"], ref=c["ref"])
        try:
            b = get_bounds(Car._lateral)
            # per strip: the l bounds (constant over the knots) and the knots whose s bounds are not the free road's
            strips = []
            for s_b, l_b in b:
                assert all(list(p) == list(l_b[0]) for p in l_b) and len(s_b) == GLOBALS["num_of_knots"]
                free = [GLOBALS["s_l_l"], GLOBALS["s_u_l"]]
                strips.append(dict(l=[float(v) for v in l_b[0]],
                                   s=[[i, float(p[0]), float(p[1])] for i, p in enumerate(s_b) if [float(p[0]), float(p[1])] != free]))
            err = None
        except Exception as e:                  # the reference raises on some constellations (e.g. max() of nothing)
            strips, err = None, type(e).__name__
        out.append(dict(cars=[dict(c, centre=list(c["centre"])) for c in cars], strips=strips, error=err))
    json.dump(dict(globals=GLOBALS, scenes=out), open(os.path.join(HERE, "prism_goldens.json"), "w"), separators=(",", ":"))
    print("wrote prism_goldens.json:", len(out), "scenes,", sum(s["error"] is not None for s in out), "on which the reference raises")


if __name__ == "__main__":
    main()
