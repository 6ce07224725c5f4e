"""ctypes loader for the CPU oracle (oracle/btrapz_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under spectral_amd/ may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libbtrapz_oracle.so")

TRAPEZOID, CUBOID = 0, 1
FAIL_SENTINEL = 100000000000.0


class Cube(C.Structure):
    _fields_ = [("beg_t", C.c_int), ("end_t", C.c_int), ("t", C.c_double),
                ("beg_l", C.c_double), ("end_l", C.c_double),
                ("upp_skew", C.c_double), ("upp_bias", C.c_double),
                ("down_skew", C.c_double), ("down_bias", C.c_double),
                ("l_upp_skew", C.c_double), ("l_upp_bias", C.c_double),
                ("l_down_skew", C.c_double), ("l_down_bias", C.c_double),
                ("count", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Params(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "s_acc_weight", "s_jerk_weight", "l_acc_weight", "l_jerk_weight",
        "weight_s_ref", "weight_ds_ref", "weight_l_ref", "weight_dl_ref",
        "weight_end_s", "weight_end_l")] + [("iteration", C.c_int)]


class Input(C.Structure):
    _fields_ = [("N", C.c_int), ("delta", C.c_double),
                ("init_s", C.c_double * 3), ("init_l", C.c_double * 3),
                ("num_obs", C.c_int), ("ds_ref", C.c_double), ("dl_ref", C.c_double),
                ("dds", C.c_double * 2), ("ddds", C.c_double * 2),
                ("ddl", C.c_double * 2), ("dddl", C.c_double * 2),
                ("x_bounds", C.POINTER(C.c_double)), ("y_bounds", C.POINTER(C.c_double)),
                ("dx_bounds", C.POINTER(C.c_double)), ("dy_bounds", C.POINTER(C.c_double)),
                ("x_ref", C.POINTER(C.c_double)), ("y_ref", C.POINTER(C.c_double)),
                ("x_kappa", C.POINTER(C.c_double)), ("y_kappa", C.POINTER(C.c_double))]


class QpParams(C.Structure):
    _fields_ = [("w_s", C.c_double * 4), ("w_l", C.c_double * 4),
                ("weight_end_s", C.c_double), ("weight_end_l", C.c_double),
                ("ds_ref", C.c_double), ("dl_ref", C.c_double),
                ("dds", C.c_double * 2), ("ddds", C.c_double * 2),
                ("ddl", C.c_double * 2), ("dddl", C.c_double * 2),
                ("init_s", C.c_double * 3), ("init_l", C.c_double * 3),
                ("N", C.c_int), ("delta", C.c_double),
                ("dx_bounds", C.POINTER(C.c_double)), ("dy_bounds", C.POINTER(C.c_double)),
                ("x_ref", C.POINTER(C.c_double)), ("y_ref", C.POINTER(C.c_double))]


class Qp(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int),
                ("P_p", C.POINTER(C.c_longlong)), ("P_i", C.POINTER(C.c_longlong)),
                ("P_x", C.POINTER(C.c_double)), ("P_nnz", C.c_int),
                ("A_p", C.POINTER(C.c_longlong)), ("A_i", C.POINTER(C.c_longlong)),
                ("A_x", C.POINTER(C.c_double)), ("A_nnz", C.c_int),
                ("q", C.POINTER(C.c_double)), ("l", C.POINTER(C.c_double)),
                ("u", C.POINTER(C.c_double))]


class Settings(C.Structure):
    _fields_ = [("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double),
                ("eps_abs", C.c_double), ("eps_rel", C.c_double),
                ("eps_prim_inf", C.c_double), ("eps_dual_inf", C.c_double),
                ("max_iter", C.c_int), ("scaling", C.c_int),
                ("scaled_termination", C.c_int), ("check_termination", C.c_int),
                ("adaptive_rho", C.c_int), ("adaptive_rho_interval", C.c_int),
                ("adaptive_rho_tolerance", C.c_double), ("polish", C.c_int)]


class Info(C.Structure):
    _fields_ = [("status", C.c_int), ("iter", C.c_int), ("rho_updates", C.c_int),
                ("obj_val", C.c_double), ("pri_res", C.c_double),
                ("dua_res", C.c_double), ("rho", C.c_double)]


def build(force=False):
    """Compile the oracle (building the checker is not using it)."""
    src = os.path.join(_HERE, "btrapz_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libbtrapz_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_input_read.argtypes = [C.c_char_p, C.POINTER(Input)]
        L.orc_input_read.restype = C.c_int
        L.orc_input_free.argtypes = [C.POINTER(Input)]
        L.orc_corridor_generation.argtypes = [C.c_int, C.c_int, C.c_double,
                                              C.POINTER(C.c_double), C.POINTER(C.c_double),
                                              C.POINTER(Cube), C.c_int]
        L.orc_corridor_generation.restype = C.c_int
        L.orc_collision_check.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(Cube),
                                          C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), C.POINTER(Cube), C.c_int]
        L.orc_collision_check.restype = C.c_int
        L.orc_assemble.argtypes = [C.c_int, C.c_int, C.POINTER(Cube), C.POINTER(QpParams),
                                   C.POINTER(Qp)]
        L.orc_assemble.restype = C.c_int
        L.orc_qp_free.argtypes = [C.POINTER(Qp)]
        L.orc_settings_reference.argtypes = [C.POINTER(Settings)]
        L.orc_settings_tight.argtypes = [C.POINTER(Settings)]
        L.orc_osqp_solve.argtypes = [C.POINTER(Qp), C.POINTER(Settings), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(Info)]
        L.orc_osqp_solve.restype = C.c_int
        L.orc_ipm_solve.argtypes = [C.POINTER(Qp), C.c_double, C.c_int, C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), C.POINTER(Info)]
        L.orc_ipm_solve.restype = C.c_int
        L.orc_elastic_solve.argtypes = [C.POINTER(Qp), C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.POINTER(Info)]
        L.orc_elastic_solve.restype = C.c_int
        L.orc_kkt_residuals.argtypes = [C.POINTER(Qp), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_sample.argtypes = [C.c_int, C.POINTER(Cube), C.c_double, C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), C.POINTER(C.c_double)] + \
            [C.POINTER(C.c_double)] * 6 + [C.c_int, C.POINTER(C.c_int)]
        L.orc_sample.restype = C.c_int
        L.orc_acost.argtypes = [C.c_int, C.POINTER(Params), C.POINTER(Input), C.c_int] + \
            [C.POINTER(C.c_double)] * 6
        L.orc_acost.restype = C.c_double
        L.orc_find_traj.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.POINTER(Params),
                                    C.POINTER(Settings), C.POINTER(C.c_int),
                                    C.POINTER(C.c_double), C.POINTER(Cube), C.POINTER(Info)]
        L.orc_find_traj.restype = C.c_double
        L.orc_batch_solve.argtypes = [C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_double)] * 5 + \
            [C.POINTER(Settings), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
             C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_batch_solve.restype = C.c_int
        _bind_mt(L)
        _lib = L
    return _lib


def _bind_mt(L):
    L.orc_batch_solve_mt.argtypes = [C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_double)] * 5 + \
        [C.POINTER(Settings), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
         C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_batch_solve_mt.restype = C.c_int
    L.orc_settings_reference.argtypes = [C.POINTER(Settings)]


_fast = None


def fast_lib():
    """The same source at -O3 -march=native for the CPU this runs on (bench.py's cpu_baseline leg only).  Built on
    first use, one file per CPU model so that a library built in the container is not run on the GPU box's host."""
    global _fast
    if _fast is None:
        import hashlib
        model = ""
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith(("model name", "flags")):
                    model += line
                    if line.startswith("flags"):
                        break
        except OSError:
            pass
        name = "libbtrapz_oracle_fast_%s.so" % hashlib.sha256(model.encode()).hexdigest()[:10]
        path = os.path.join(_HERE, name)
        src = os.path.join(_HERE, "btrapz_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "fast", "FAST=" + name], stdout=subprocess.DEVNULL)
        L = C.CDLL(path)
        _bind_mt(L)
        _fast = L
    return _fast


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def settings_reference():
    s = Settings()
    lib().orc_settings_reference(C.byref(s))
    return s


def settings_tight():
    s = Settings()
    lib().orc_settings_tight(C.byref(s))
    return s


def params_from_weights(w, iteration=3):
    """weights.txt order == Params field order (trp_wrapper.py:104-115)."""
    return Params(*[float(v) for v in w[:10]], int(iteration))


class ParsedInput:
    """Owns an orc_input; exposes numpy copies."""

    def __init__(self, path):
        self.raw = Input()
        rc = lib().orc_input_read(os.fsencode(path), C.byref(self.raw))
        if rc != 0:
            raise IOError("orc_input_read(%s) -> %d" % (path, rc))
        r = self.raw
        N, O = r.N, r.num_obs
        self.N, self.delta, self.num_obs = N, r.delta, O
        self.init_s = np.array(r.init_s[:]); self.init_l = np.array(r.init_l[:])
        self.ds_ref, self.dl_ref = r.ds_ref, r.dl_ref
        self.dds = np.array(r.dds[:]); self.ddds = np.array(r.ddds[:])
        self.ddl = np.array(r.ddl[:]); self.dddl = np.array(r.dddl[:])
        cp = lambda p, n: np.ctypeslib.as_array(p, shape=(n,)).copy()
        self.x_bounds = cp(r.x_bounds, O * N * 2).reshape(O, N, 2)
        self.y_bounds = cp(r.y_bounds, O * N * 2).reshape(O, N, 2)
        self.dx_bounds = cp(r.dx_bounds, N * 2).reshape(N, 2)
        self.dy_bounds = cp(r.dy_bounds, N * 2).reshape(N, 2)
        self.x_ref = cp(r.x_ref, N); self.y_ref = cp(r.y_ref, N)
        self.x_kappa = cp(r.x_kappa, N); self.y_kappa = cp(r.y_kappa, N)

    def __del__(self):
        try:
            lib().orc_input_free(C.byref(self.raw))
        except Exception:
            pass


def corridor_generation(variant, N, delta, xb, yb, cap=4096):
    xb = np.ascontiguousarray(xb, dtype=np.float64); yb = np.ascontiguousarray(yb, dtype=np.float64)
    out = (Cube * cap)()
    n = lib().orc_corridor_generation(variant, N, delta, _dp(xb), _dp(yb), out, cap)
    if n < 0:
        raise RuntimeError("corridor_generation -> %d" % n)
    return [out[i] for i in range(n)]


def collision_check(variant, N, delta, cube_lists, x_ref, y_ref, cap=4096):
    flat = [c for lst in cube_lists for c in lst]
    arr = (Cube * max(1, len(flat)))(*flat)
    counts = (C.c_int * max(1, len(cube_lists)))(*[len(l) for l in cube_lists])
    x_ref = np.ascontiguousarray(x_ref, dtype=np.float64); y_ref = np.ascontiguousarray(y_ref, dtype=np.float64)
    out = (Cube * cap)()
    n = lib().orc_collision_check(variant, N, delta, arr, counts, len(cube_lists), _dp(x_ref),
                                  _dp(y_ref), out, cap)
    return n, [out[i] for i in range(max(n, 0))]


def pipeline(variant, inp):
    """find_traj's corridor stage: per-obstacle generation then CollisionCheck."""
    lists = [corridor_generation(variant, inp.N, inp.delta, inp.x_bounds[o], inp.y_bounds[o])
             for o in range(inp.num_obs)]
    return collision_check(variant, inp.N, inp.delta, lists, inp.x_ref, inp.y_ref)


class AssembledQp:
    def __init__(self, variant, cubes, params, inp=None, **kw):
        """params: Params (weights); inp: ParsedInput or kwargs with the same fields."""
        src = inp if inp is not None else type("K", (), kw)
        self._keep = [np.ascontiguousarray(getattr(src, k), dtype=np.float64)
                      for k in ("dx_bounds", "dy_bounds", "x_ref", "y_ref")]
        qp_ = QpParams()
        qp_.w_s[:] = [params.weight_s_ref, params.weight_ds_ref, params.s_acc_weight, params.s_jerk_weight]
        qp_.w_l[:] = [params.weight_l_ref, params.weight_dl_ref, params.l_acc_weight, params.l_jerk_weight]
        qp_.weight_end_s, qp_.weight_end_l = params.weight_end_s, params.weight_end_l
        qp_.ds_ref, qp_.dl_ref = src.ds_ref, src.dl_ref
        qp_.dds[:] = list(src.dds); qp_.ddds[:] = list(src.ddds)
        qp_.ddl[:] = list(src.ddl); qp_.dddl[:] = list(src.dddl)
        qp_.init_s[:] = list(src.init_s); qp_.init_l[:] = list(src.init_l)
        qp_.N, qp_.delta = int(src.N), float(src.delta)
        qp_.dx_bounds, qp_.dy_bounds, qp_.x_ref, qp_.y_ref = [_dp(a) for a in self._keep]
        self.qpp = qp_
        self.S = len(cubes)
        self.cubes = (Cube * self.S)(*cubes)
        self.raw = Qp()
        rc = lib().orc_assemble(variant, self.S, self.cubes, C.byref(qp_), C.byref(self.raw))
        if rc != 0:
            raise RuntimeError("orc_assemble -> %d" % rc)
        r = self.raw
        self.n, self.m = r.n, r.m
        ci = lambda p, n: np.ctypeslib.as_array(p, shape=(n,)).copy()
        self.P_p = ci(r.P_p, r.n + 1); self.P_i = ci(r.P_i, r.P_nnz); self.P_x = ci(r.P_x, r.P_nnz)
        self.A_p = ci(r.A_p, r.n + 1); self.A_i = ci(r.A_i, r.A_nnz); self.A_x = ci(r.A_x, r.A_nnz)
        self.q = ci(r.q, r.n); self.l = ci(r.l, r.m); self.u = ci(r.u, r.m)

    def dense(self):
        import scipy.sparse as sp
        Pu = sp.csc_matrix((self.P_x, self.P_i, self.P_p), shape=(self.n, self.n)).toarray()
        P = Pu + Pu.T - np.diag(np.diag(Pu))
        A = sp.csc_matrix((self.A_x, self.A_i, self.A_p), shape=(self.m, self.n)).toarray()
        return P, A

    def solve(self, settings=None):
        s = settings if settings is not None else settings_reference()
        x = np.zeros(self.n); y = np.zeros(self.m); info = Info()
        lib().orc_osqp_solve(C.byref(self.raw), C.byref(s), _dp(x), _dp(y), C.byref(info))
        return x, y, info

    def solve_exact(self, eps=1e-9, max_iter=80):
        """x* by the oracle's dense interior-point method."""
        x = np.zeros(self.n); y = np.zeros(self.m); info = Info()
        lib().orc_ipm_solve(C.byref(self.raw), eps, max_iter, _dp(x), _dp(y), C.byref(info))
        return x, y, info

    def solve_elastic(self, delta=1e-8, eps=1e-9, max_iter=120, normalised=True):
        """Least-violation solution of the product's rescue pass (btrapz_options.elastic): x, y, info, largest row
        violation.  normalised (the product's problem since round 3): every inequality row is relaxed in its own norm,
        penalty (d / |a_i|)^2 / (2 delta) -- orc_elastic_solve on the QP with its inequality rows (and their bounds)
        divided by |a_i|; the violation returned is in that unit too, `self.class_violations(x)` gives the rows' own.
        normalised=False: round 2's problem, d^2 / (2 delta) on every row alike."""
        x = np.zeros(self.n); y = np.zeros(self.m); info = Info()
        P, A = self.dense()
        ineq = (self.u - self.l) > 1e-12
        nrm = np.ones(self.m)
        if normalised:
            nrm[ineq] = np.linalg.norm(A[ineq], axis=1)
            scaled = DenseQp(P, np.array(self.q, dtype=float), A / nrm[:, None], self.l / nrm, self.u / nrm)
            lib().orc_elastic_solve(C.byref(scaled.raw), float(delta), eps, max_iter, _dp(x), _dp(y), C.byref(info))
            y = y / nrm
        else:
            lib().orc_elastic_solve(C.byref(self.raw), float(delta), eps, max_iter, _dp(x), _dp(y), C.byref(info))
        Ax = A @ x
        viol = (np.abs(Ax - np.clip(Ax, self.l, self.u)) / nrm)[ineq].max() if ineq.any() else 0.0
        return x, y, info, float(viol)

    def class_violations(self, x):
        """Largest violation of a position / velocity / acceleration / jerk row (1, 2, 3, 4 coefficients per row:
        solve_3d.cc:823-888) by x, in the rows' own units."""
        _, A = self.dense()
        Ax = A @ x
        v = np.abs(Ax - np.clip(Ax, self.l, self.u))
        ineq = (self.u - self.l) > 1e-12
        nnz = (A != 0).sum(axis=1)
        return [float(v[ineq & (nnz == c)].max()) if (ineq & (nnz == c)).any() else 0.0 for c in (1, 2, 3, 4)]

    def kkt(self, x, y):
        res = np.zeros(3)
        x = np.ascontiguousarray(x); y = np.ascontiguousarray(y)
        lib().orc_kkt_residuals(C.byref(self.raw), _dp(x), _dp(y), _dp(res))
        return res

    def __del__(self):
        try:
            lib().orc_qp_free(C.byref(self.raw))
        except Exception:
            pass


class DenseQp(AssembledQp):
    """A general QP given as dense numpy arrays (tests of the oracle's own solvers)."""

    def __init__(self, P, q, A, l, u):
        import scipy.sparse as sp
        Pu = sp.csc_matrix(np.triu(P)); Ac = sp.csc_matrix(A)
        Pu.sort_indices(); Ac.sort_indices()
        self.n, self.m = len(q), len(l)
        ll = lambda a: np.ascontiguousarray(a, dtype=np.int64)
        self.P_p, self.P_i, self.P_x = ll(Pu.indptr), ll(Pu.indices), np.ascontiguousarray(Pu.data, dtype=np.float64)
        self.A_p, self.A_i, self.A_x = ll(Ac.indptr), ll(Ac.indices), np.ascontiguousarray(Ac.data, dtype=np.float64)
        self.q, self.l, self.u = [np.ascontiguousarray(v, dtype=np.float64) for v in (q, l, u)]
        lp = lambda a: a.ctypes.data_as(C.POINTER(C.c_longlong))
        self.raw = Qp(self.n, self.m, lp(self.P_p), lp(self.P_i), _dp(self.P_x), len(self.P_x), lp(self.A_p), lp(self.A_i),
                      _dp(self.A_x), len(self.A_x), _dp(self.q), _dp(self.l), _dp(self.u))

    def __del__(self):   # the arrays belong to numpy
        pass


def sample(cubes, delta, x, init_s, init_l, cap=4096):
    S = len(cubes)
    arr = (Cube * S)(*cubes)
    x = np.ascontiguousarray(x, dtype=np.float64)
    i_s = np.ascontiguousarray(init_s, dtype=np.float64); i_l = np.ascontiguousarray(init_l, dtype=np.float64)
    bufs = [np.zeros(cap) for _ in range(6)]
    npnt = C.c_int(0)
    rc = lib().orc_sample(S, arr, delta, _dp(x), _dp(i_s), _dp(i_l), *[_dp(b) for b in bufs], cap,
                          C.byref(npnt))
    return rc, [b[:max(npnt.value, 0)].copy() for b in bufs]


def find_traj(variant, input_path, output_path, params, settings=None):
    S = C.c_int(0)
    ctrl = np.zeros(12 * 64)
    cubes = (Cube * 64)()
    info = Info()
    cost = lib().orc_find_traj(variant, os.fsencode(input_path),
                               os.fsencode(output_path) if output_path else None,
                               C.byref(params), C.byref(settings) if settings is not None else None,
                               C.byref(S), _dp(ctrl), cubes, C.byref(info))
    s = S.value
    return cost, s, ctrl[:12 * max(s, 0)].copy(), [cubes[i] for i in range(max(min(s, 64), 0))], info


def batch_solve(batch, shared, b0=0, b1=None, exact=False, settings=None, threads=1, fast=False):
    """Solve candidates [b0,b1) of a spectral_amd.layout.Batch with the oracle (OSQP port, or
    x* when exact).  threads > 1: POSIX threads inside the C library drawing candidates from a shared
    counter (orc_batch_solve_mt).  fast: the -O3 -march=native build (timing only; same arithmetic)."""
    B, S = batch.B, batch.S
    b1 = B if b1 is None else b1
    seg = np.ascontiguousarray(batch.seg, dtype=np.float64); init = np.ascontiguousarray(batch.init, dtype=np.float64)
    ref_end = np.ascontiguousarray(batch.ref_end, dtype=np.float64); dlb = np.ascontiguousarray(batch.dl_bounds, dtype=np.float64)
    sh = np.ascontiguousarray(shared.as_array())
    ctrl = np.zeros((B, 12 * S)); obj = np.zeros(B); status = np.zeros(B, dtype=np.int32); iters = np.zeros(B, dtype=np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    sp = C.byref(settings) if settings is not None else None

    L = fast_lib() if fast else lib()
    rc = L.orc_batch_solve_mt(shared.variant, B, S, _dp(seg), _dp(init), _dp(ref_end), _dp(dlb), _dp(sh), sp,
                              1 if exact else 0, int(b0), int(b1), int(max(1, threads)), _dp(ctrl), _dp(obj), ip(status),
                              ip(iters))
    if rc != 0:
        raise RuntimeError("orc_batch_solve_mt -> %d" % rc)
    return ctrl[b0:b1], obj[b0:b1], status[b0:b1], iters[b0:b1]


def host_cpu_info():
    """What bounds the CPU baseline on this host: schedulable cores and the cgroup's CPU quota."""
    info = {"sched_affinity": len(os.sched_getaffinity(0)), "cgroup_cpu_max": None}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            info["cgroup_cpu_max"] = open(path).read().strip()
            break
        except OSError:
            continue
    q = info["cgroup_cpu_max"]
    eff = info["sched_affinity"]
    if q:
        parts = q.split()
        try:
            if len(parts) == 2 and parts[0] != "max":
                eff = min(eff, max(1, int(float(parts[0]) / float(parts[1]) + 0.5)))
            elif len(parts) == 1 and int(parts[0]) > 0:
                eff = min(eff, max(1, int(int(parts[0]) / 100000 + 0.5)))
        except ValueError:
            pass
    info["effective_cores"] = eff
    return info
