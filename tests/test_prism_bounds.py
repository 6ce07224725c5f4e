"""SURVEY 8(f) rank 4, CPU side: the restatement of the reference's obstacle-prism -> bounds generator
(oracle/prism_oracle.py) against vectors the reference's own `Car` / `get_bounds` produced
(tests/golden/prism_goldens.json, tests/golden/make_prism_goldens.py)."""
import json
import os

import numpy as np

from oracle import prism_oracle as P

GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = json.load(open(os.path.join(GOLD, "prism_goldens.json")))


def dense(strip, n, free):
    s = [list(free) for _ in range(n)]
    for i, lo, hi in strip["s"]:
        s[i] = [lo, hi]
    return s


def test_restatement_equals_the_reference_wherever_the_reference_is_well_formed():
    """Bit for bit on every stored scene whose reference output is a proper partition of the road: all single-car
    scenes, the harness's own two-car constructions, disjoint and most overlapping constellations.  The others are
    exactly the scenes where the reference's pairwise, construction-order-dependent edge bookkeeping emits an inverted
    (l_lo >= l_hi) or out-of-order strip: there the restatement's single definition applies."""
    gl = G["globals"]; n = gl["num_of_knots"]; free = (gl["s_l_l"], gl["s_u_l"])
    agree = malformed = 0
    for sc in G["scenes"]:
        assert sc["error"] is None
        mine = P.prism_bounds(sc["cars"], n, gl["s_l_l"], gl["s_u_l"], gl["d_l_l"], gl["d_u_l"])
        ref = sc["strips"]
        bad = any(r["l"][0] >= r["l"][1] for r in ref) or any(ref[i + 1]["l"][0] < ref[i]["l"][0] for i in range(len(ref) - 1))
        if bad:
            malformed += 1
            continue
        assert len(mine) == len(ref)
        for m, r in zip(mine, ref):
            assert m[1][0] == r["l"]
            assert m[0] == dense(r, n, free)
        agree += 1
    assert agree >= 50 and malformed <= agree // 3
    single = [sc for sc in G["scenes"] if len(sc["cars"]) == 1]
    assert len(single) >= 10            # (all of them are in the agreeing class: no `continue` above skips them)
    assert not any(any(r["l"][0] >= r["l"][1] for r in sc["strips"]) for sc in single)


def test_strips_partition_the_road_and_keep_clear_of_every_prism():
    """Properties that hold for every scene, well-formed reference output or not: the strips tile [min edge, max edge]
    without gaps or overlaps; inside a car's lateral extent and time window the s interval stops at the car's face."""
    rng = np.random.default_rng(4)
    for _ in range(200):
        cars = []
        for r in range(int(rng.integers(1, 5))):
            ahead = rng.uniform() < 0.5
            cars.append(dict(centre=(float(rng.uniform(5, 40)), float(rng.uniform(-3.0, 9.0)), 0 if ahead else float(rng.uniform(0.1, 3.0))),
                             vel_s=float(rng.uniform(0, 8)), vel_l=float(rng.choice([0.0, 0.3, -0.3])), time=float(rng.choice([3.0, 4.0]))))
        strips = P.prism_bounds(cars)
        ls = [s[1][0] for s in strips]
        assert all(a[1] == b[0] for a, b in zip(ls, ls[1:])) and all(a[0] < a[1] for a in ls)
        assert ls[0][0] <= -2.0 and ls[-1][1] >= 8.0
        for car in cars:
            cl, ch = P.lateral_extent(car)
            s0, l0, t0 = car["centre"]
            for (sb, lb) in strips:
                if not (cl <= lb[0][0] and lb[0][1] <= ch):
                    continue
                for i in range(71):
                    if i < t0 * 10 or i > (t0 + car["time"]) * 10:
                        continue
                    face = (s0 - P.L_SAFE if t0 == 0 else s0 + P.L_SAFE) + car["vel_s"] * (i / 10 - t0)
                    if t0 == 0:
                        assert sb[i][1] <= face + 0.006            # (2-decimal rounding of the face line)
                    else:
                        assert sb[i][0] >= face - 0.006
