"""CPU restatement of the obstacle-prism -> per-knot bounds generator of the reference's harness:
`Car.getCar` + `get_bounds` of src/cart_frenet.py:664-1030 (helpers lineFromPoints :818-830,
delete_multiple_element :808-815).

TEST INFRASTRUCTURE ONLY (see oracle.py).  Pure Python on purpose: a scene has a handful of cars.

What the reference does.  Every car is a prism in (s, l, t): centre (s0, l0, t0), constant velocities, duration T,
grown by the safety margins l_safe = 5/3 + 5/3, w_safe = 2/3 + 2/3 (:694-700).  Its lateral extent [l_min, l_max]
(:709-710, :764-765) cuts the road [d_l_l, d_u_l] into lateral strips; a strip covered by a car gets, inside the car's
time window (knots t0*10 .. (t0+T)*10, both inclusive, :909-913), the car's rear face as UPPER s bound when the car
starts at t0 = 0 ("yield": stay behind it, :905-915) and its front face as LOWER s bound otherwise (:935-945); faces
are the lines of lineFromPoints, rounded to 2 decimals.  Strips nobody covers are free ([s_l_l, s_u_l], :881-887,
:969-975, :1000-1006); strips with the same l bounds are merged by intersecting their s bounds (:987-995).

The reference gets there through a class-level list of (car, ref, l) edges filled pairwise while the cars are
constructed (:718-760), hash-ordered de-duplication (`list(set(...))`, :848; `__lt__` on hash(self), :683-684) and a
per-car walk over consecutive edges.  Its result therefore depends on construction order and on object hashes when
lateral extents coincide or nest in particular ways.  The restatement below gives the same geometry ONE definition:

    edges  = sorted distinct l_min, l_max of every car (not clipped to the road), plus d_l_l / d_u_l where the cars
             leave room below / above
    strip j = [e_j, e_(j+1)]; a car whose lateral extent contains the strip contributes, per knot, (s_l_l, rear(i))
              when it starts at t0 = 0 and (front(i), s_u_l) otherwise inside its window, (s_l_l, s_u_l) outside; the
              strip's bounds are the intersection of its cars' contributions, (s_l_l, s_u_l) when there is none

tests/golden/make_prism_goldens.py runs the reference's own functions (taken from /root/reference at generation time,
never copied) on seeded scenes and stores inputs and outputs; tests/test_prism_bounds.py holds this restatement to
every stored scene on which the two agree by construction (all but nested / coinciding extents) and counts the rest.
"""
L_SAFE = 5.0 / 3 + 5.0 / 3       # :698
W_SAFE = 2.0 / 3 + 2.0 / 3       # :699


def lateral_extent(car):
    """car = dict(centre=(s0, l0, t0), vel_s, vel_l, time).  (:705-710, :763-765)"""
    s0, l0, t0 = car["centre"]
    fl = l0 + car.get("vel_l", 0.0) * car.get("time", 3.0)
    if car.get("vel_l", 0.0) >= 0:
        return l0 - W_SAFE, fl + W_SAFE
    return fl - W_SAFE, l0 + W_SAFE


def face_line(x1, y1, x2, y2, num_of_knots):
    """lineFromPoints (:818-830): y rounded to 2 decimals at t = i / 10."""
    c = (y2 - y1) / (x2 - x1)
    return [round(c * i / 10 - c * x1 + y1, 2) for i in range(num_of_knots)]


def prism_bounds(cars, num_of_knots=71, s_l_l=0.0, s_u_l=50.0, d_l_l=-2.0, d_u_l=8.0):
    """-> list of strips [(s_bounds [N][2], l_bounds [N][2])], ascending in l."""
    ext = [lateral_extent(c) for c in cars]
    edges = set()
    for lo, hi in ext:
        edges.add(lo); edges.add(hi)
    # the road's own edges only where the cars leave room (:881-887, :969-975); car extents are not clipped to the road
    if not edges or min(edges) > d_l_l:
        edges.add(d_l_l)
    if max(edges) < d_u_l:
        edges.add(d_u_l)
    edges = sorted(edges)
    out = []
    for j in range(len(edges) - 1):
        e0, e1 = edges[j], edges[j + 1]
        lo = [s_l_l] * num_of_knots; hi = [s_u_l] * num_of_knots
        first = [True] * num_of_knots
        for car, (cl, ch) in zip(cars, ext):
            if not (cl <= e0 and e1 <= ch):
                continue
            s0, l0, t0 = car["centre"]
            T, vs = car.get("time", 3.0), car.get("vel_s", 0.0)
            ahead = t0 == 0
            fs = s0 + vs * T                                      # forw_state[0] (:704)
            y1, y2 = (s0 - L_SAFE, fs - L_SAFE) if ahead else (s0 + L_SAFE, fs + L_SAFE)   # corners 0 -> 4 / 2 -> 6 (:727-741)
            line = face_line(t0, y1, t0 + T, y2, num_of_knots)
            for i in range(num_of_knots):
                inside = not (i < t0 * 10 or i > (t0 + T) * 10)
                # a car's own bounds at a knot: its face inside the window, the road's limits outside (:908-915); the
                # face REPLACES the limit (it may lie beyond it); several cars over one strip intersect (:987-995)
                c_lo = line[i] if (inside and not ahead) else s_l_l
                c_hi = line[i] if (inside and ahead) else s_u_l
                if first[i]:
                    lo[i], hi[i], first[i] = c_lo, c_hi, False
                else:
                    lo[i], hi[i] = max(lo[i], c_lo), min(hi[i], c_hi)
        out.append(([[lo[i], hi[i]] for i in range(num_of_knots)], [[e0, e1] for _ in range(num_of_knots)]))
    return out
