"""Knot-level candidate batches: the input of the corridor stage (per-knot bounds + reference
trajectory), i.e. the content of the reference's corridor text files (grammar of
src/trp_wrapper.cpp:39-144, written by cart_frenet.py:384-453) for many candidates at once."""
from dataclasses import dataclass

import numpy as np


@dataclass
class KnotBatch:
    B: int
    N: int
    num_obs: int
    delta: float
    s_bounds: np.ndarray   # [B][num_obs][N][2]
    l_bounds: np.ndarray   # [B][num_obs][N][2]
    ds_bounds: np.ndarray  # [B][N][2]
    dl_bounds: np.ndarray  # [B][N][2]
    s_ref: np.ndarray      # [B][N]
    l_ref: np.ndarray      # [B][N]
    init: np.ndarray       # [B][6]
    header: dict           # ds_ref, dl_ref, dds, ddds, ddl, dddl of the file


def parse_corridor_file(path):
    """Tolerant token parser (a short last row leaves the tail at its previous value, as the
    reference's unchecked `ifs >>` does).  Returns a KnotBatch with B = 1."""
    tok = open(path).read().split()
    pos = [0]
    last = [0.0]

    def nxt():
        if pos[0] < len(tok):
            try:
                last[0] = float(tok[pos[0]])
            except ValueError:
                pos[0] = len(tok)
            pos[0] += 1
        return last[0]
    N = int(nxt()); delta = nxt()
    init = np.array([nxt() for _ in range(6)])
    O = int(nxt())
    ds_ref, dl_ref = nxt(), nxt()
    dds = (nxt(), nxt()); ddds = (nxt(), nxt()); ddl = (nxt(), nxt()); dddl = (nxt(), nxt())
    sb = np.zeros((O, N, 2)); lb = np.zeros((O, N, 2))
    for o in range(O):
        for i in range(N):
            sb[o, i] = (nxt(), nxt())
        for i in range(N):
            lb[o, i] = (nxt(), nxt())
    dsb = np.array([(nxt(), nxt()) for _ in range(N)]); dlb = np.array([(nxt(), nxt()) for _ in range(N)])
    s_ref = np.array([nxt() for _ in range(N)]); l_ref = np.array([nxt() for _ in range(N)])
    hdr = dict(ds_ref=ds_ref, dl_ref=dl_ref, dds=dds, ddds=ddds, ddl=ddl, dddl=dddl)
    return KnotBatch(1, N, O, delta, sb[None], lb[None], dsb[None], dlb[None], s_ref[None], l_ref[None], init[None], hdr)


def write_corridor_file(path, kb, b=0):
    """Candidate b of a KnotBatch as the text file find_traj reads (the inverse of parse_corridor_file; numbers at
    full precision, nan / inf as strtod reads them)."""
    h = kb.header
    r = lambda v: repr(float(v))
    pairs = lambda a: " ".join("%s %s" % (r(x), r(y)) for x, y in a)
    with open(path, "w") as f:
        f.write("%d %s\n" % (kb.N, r(kb.delta)))
        f.write(" ".join(r(v) for v in kb.init[b]) + "\n%d\n" % kb.num_obs)
        f.write("%s %s\n" % (r(h["ds_ref"]), r(h["dl_ref"])))
        f.write(" ".join("%s %s" % (r(h[k][0]), r(h[k][1])) for k in ("dds", "ddds", "ddl", "dddl")) + "\n")
        for o in range(kb.num_obs):
            f.write(pairs(kb.s_bounds[b, o]) + "\n" + pairs(kb.l_bounds[b, o]) + "\n")
        f.write(pairs(kb.ds_bounds[b]) + "\n" + pairs(kb.dl_bounds[b]) + "\n")
        f.write(" ".join(r(v) for v in kb.s_ref[b]) + "\n" + " ".join(r(v) for v in kb.l_ref[b]) + "\n")


def jittered(kb, B, seed=0, s_shift=0.4, l_shift=0.05):
    """B candidates around a parsed file: per-candidate shifts of the corridor bounds and of the
    reference (obstacle position / lane offset jitter).  Shifts are applied to whole obstacles so the
    slope structure -- hence the segmentation logic -- stays the file's."""
    rng = np.random.default_rng(seed)
    rep = lambda a: np.repeat(a, B, axis=0).copy()
    out = KnotBatch(B, kb.N, kb.num_obs, kb.delta, rep(kb.s_bounds), rep(kb.l_bounds), rep(kb.ds_bounds),
                    rep(kb.dl_bounds), rep(kb.s_ref), rep(kb.l_ref), rep(kb.init), dict(kb.header))
    ds = rng.uniform(-s_shift, s_shift, (B, kb.num_obs, 1, 1)); ds[0] = 0.0
    out.s_bounds[..., 1:] += ds          # upper s bound of each obstacle corridor
    dr = rng.uniform(-s_shift, s_shift, (B, 1)) * np.linspace(0, 1, kb.N)[None, :]; dr[0] = 0.0
    out.s_ref += dr
    dlr = rng.uniform(-l_shift, l_shift, (B, 1)); dlr[0] = 0.0
    out.l_ref += dlr
    return out
