"""ctypes binding of libbtrapz_hip.so (include/btrapz_hip.h).

The library is the product; this module only marshals pointers.  There is no CPU or
PyTorch fallback: if the HIP library is missing, or no HIP device is visible, every
solve raises."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import layout as L

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
CSRC_DIR = os.path.join(_HERE, "csrc")
# BTRAPZ_HIP_LIB: another build of the library (tools: A/B runs of kernel variants on one GPU box); default: the in-tree build
LIB_PATH = os.environ.get("BTRAPZ_HIP_LIB") or os.path.join(LIB_DIR, "libbtrapz_hip.so")


class BtrapzError(RuntimeError):
    pass


class CShared(C.Structure):
    _fields_ = [("w_s", C.c_double * 4), ("w_l", C.c_double * 4),
                ("weight_end_s", C.c_double), ("weight_end_l", C.c_double),
                ("ds_ref", C.c_double), ("dl_ref", C.c_double),
                ("dds", C.c_double * 2), ("ddds", C.c_double * 2),
                ("ddl", C.c_double * 2), ("dddl", C.c_double * 2),
                ("delta", C.c_double), ("variant", C.c_int), ("reserved", C.c_int)]

    @classmethod
    def from_shared(cls, sh):
        s = cls()
        s.w_s[:] = sh.w_s; s.w_l[:] = sh.w_l
        s.weight_end_s, s.weight_end_l = sh.weight_end_s, sh.weight_end_l
        s.ds_ref, s.dl_ref = sh.ds_ref, sh.dl_ref
        s.dds[:] = sh.dds; s.ddds[:] = sh.ddds; s.ddl[:] = sh.ddl; s.dddl[:] = sh.dddl
        s.delta, s.variant = sh.delta, sh.variant
        return s


class COptions(C.Structure):
    """btrapz_options; struct_size is what btrapz_options_init() sets (the library rejects other layouts)."""
    _fields_ = [("struct_size", C.c_int), ("max_iter", C.c_int), ("eps", C.c_double), ("step_fraction", C.c_double),
                ("step_threshold", C.c_double), ("elastic", C.c_int), ("elastic_tol", C.c_double),
                ("elastic_delta", C.c_double), ("queue", C.c_int), ("split", C.c_int), ("start", C.c_int), ("cap_iter", C.c_int), ("lean", C.c_int), ("compact", C.c_int)]


def _options(max_iter=0, eps=0.0, elastic=0, elastic_tol=0.0, elastic_delta=0.0, queue=0, split=0, start=0, cap_iter=0, lean=0, compact=0):
    return COptions(C.sizeof(COptions), int(max_iter), float(eps), float(os.environ.get("BTRAPZ_STEP_FRACTION", "0")),
                    float(os.environ.get("BTRAPZ_STEP_THRESHOLD", "0")), int(elastic), float(elastic_tol),
                    float(elastic_delta), int(queue), int(split), int(start), int(cap_iter), int(lean), int(compact))


class CWarm(C.Structure):
    """btrapz_warm (include/btrapz_hip.h): optional warm start of a solve."""
    _fields_ = [("x0", C.c_void_p), ("lam0", C.c_void_p), ("lam_out", C.c_void_p),
                ("mu0", C.c_double), ("smin", C.c_double), ("hint", C.c_void_p)]


class CTrajInput(C.Structure):
    """btrapz_traj_input: the content of the corridor text file (trp_wrapper.cpp:39-144) as arrays."""
    _fields_ = [("N", C.c_int), ("num_obs", C.c_int), ("delta", C.c_double),
                ("init_s", C.c_double * 3), ("init_l", C.c_double * 3),
                ("ds_ref", C.c_double), ("dl_ref", C.c_double),
                ("dds", C.c_double * 2), ("ddds", C.c_double * 2), ("ddl", C.c_double * 2), ("dddl", C.c_double * 2),
                ("s_bounds", C.c_void_p), ("l_bounds", C.c_void_p), ("ds_bounds", C.c_void_p),
                ("dl_bounds", C.c_void_p), ("s_ref", C.c_void_p), ("l_ref", C.c_void_p)]


class CMultiShard(C.Structure):
    """btrapz_multi_shard: a shard that is on its device already."""
    _fields_ = [("B", C.c_int), ("index_base", C.c_longlong), ("seg", C.c_void_p), ("init", C.c_void_p),
                ("ref_end", C.c_void_p), ("dl_bounds", C.c_void_p)]


class CMultiView(C.Structure):
    """btrapz_multi_view: what lives on one device slot after a step (device pointers)."""
    _fields_ = [("device", C.c_int), ("B", C.c_int), ("index_base", C.c_longlong), ("stream", C.c_void_p), ("ctx", C.c_void_p),
                ("ctrl", C.c_void_p), ("cost", C.c_void_p), ("status", C.c_void_p), ("iters", C.c_void_p),
                ("best_idx", C.c_void_p), ("best_cost", C.c_void_p), ("best_ctrl", C.c_void_p)]


MULTI_AUTO, MULTI_COPIES, MULTI_RCCL = 0, 1, 2


class CParams(C.Structure):
    """include/btrapz/py_cpp_.h:6-21 == trp_wrapper.py:19-32."""
    _fields_ = [("s_acc_weight", C.c_double), ("s_jerk_weight", C.c_double),
                ("l_acc_weight", C.c_double), ("l_jerk_weight", C.c_double),
                ("weight_s_ref", C.c_double), ("weight_ds_ref", C.c_double),
                ("weight_l_ref", C.c_double), ("weight_dl_ref", C.c_double),
                ("weight_end_s", C.c_double), ("weight_end_l", C.c_double),
                ("iteration", C.c_int)]


class CSegment(C.Structure):
    """btrapz_segment == Cube (include/btrapz/cube_type.h:2-24)."""
    _fields_ = [("beg_t", C.c_int), ("end_t", C.c_int), ("t", C.c_double),
                ("beg_l", C.c_double), ("end_l", C.c_double),
                ("upp_skew", C.c_double), ("upp_bias", C.c_double),
                ("down_skew", C.c_double), ("down_bias", C.c_double),
                ("l_upp_skew", C.c_double), ("l_upp_bias", C.c_double),
                ("l_down_skew", C.c_double), ("l_down_bias", C.c_double),
                ("count", C.c_int)]


class CRoad(C.Structure):
    """btrapz_road: road limits and safety margins of the prism -> bounds stage (cart_frenet.py:54-58, 698-699)."""
    _fields_ = [("s_lo", C.c_double), ("s_hi", C.c_double), ("l_lo", C.c_double), ("l_hi", C.c_double),
                ("l_safe", C.c_double), ("w_safe", C.c_double), ("knots_per_second", C.c_double)]

    @classmethod
    def reference(cls):
        return cls(0.0, 50.0, -2.0, 8.0, 5.0 / 3 + 5.0 / 3, 2.0 / 3 + 2.0 / 3, 10.0)


EXPORTS = ("btrapz_corridor_from_file", "btrapz_find_traj", "btrapz_create", "btrapz_destroy", "btrapz_last_error",
           "btrapz_device_count", "btrapz_solve_batch_device", "btrapz_argmin_device",
           "btrapz_sample_device", "btrapz_solve_batch_host", "btrapz_solve_ragged_device",
           "btrapz_corridor_batch_device", "btrapz_sample_ragged_device", "btrapz_solve_warm_device",
           "btrapz_eval_states_device", "btrapz_find_traj_mem", "btrapz_find_traj_mem_cap", "btrapz_prism_bounds_device",
           "btrapz_prism_corridor_batch_device",
           "btrapz_find_traj_last_iterations", "btrapz_argmin_pairs_device", "btrapz_options_init",
           "btrapz_rescue_violations_device", "btrapz_find_traj_last_status", "btrapz_debug_mqm_tables",
           "btrapz_debug_axis_records", "btrapz_debug_resume_keys", "btrapz_debug_parse_double", "btrapz_debug_format_fixed",
           "btrapz_last_solve_form", "btrapz_build_has_experiments", "btrapz_workspace_bytes",
           "btrapz_multi_create", "btrapz_multi_destroy", "btrapz_multi_last_error", "btrapz_multi_transport",
           "btrapz_multi_transport_library", "btrapz_multi_device_count", "btrapz_multi_shard_bounds", "btrapz_multi_upload",
           "btrapz_multi_set_shards", "btrapz_multi_solve_argmin", "btrapz_multi_result", "btrapz_multi_wait",
           "btrapz_multi_shard_view", "btrapz_multi_download")


def build(verbose=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-j4", "-C", CSRC_DIR, "all"], stdout=out)
    return LIB_PATH


def kernel_source_hash():
    """Identifies the kernel build a counter profile belongs to: sha256 over the sources and flags that determine
    the solve kernel's code (profiles/*.json carry it; bench.py refuses profiles of other kernel sources)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("btrapz_kernels.hip", "btrapz_lean.hip", "btrapz_lean_warm.hip", "btrapz_lean_body.h", "btrapz_ipm.h", "btrapz_device.h", "Makefile",
              os.path.join("..", "..", "include", "btrapz_hip.h")):
        with open(os.path.join(CSRC_DIR, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_lib = None
ROCM_HIP_RUNTIME = "/opt/rocm/lib/libamdhip64.so"


def _bind_hip_runtime():
    """libbtrapz_hip.so is linked without a HIP runtime of its own (csrc/Makefile): a process
    must hold exactly one libamdhip64.  If PyTorch-ROCm is already imported its bundled runtime
    is promoted to the global symbol scope and reused; otherwise the system runtime is loaded."""
    import sys
    path = ROCM_HIP_RUNTIME
    if "torch" in sys.modules:
        bundled = os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib", "libamdhip64.so")
        if os.path.exists(bundled):
            path = bundled
    try:
        C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise BtrapzError("cannot load the HIP runtime %s: %s (there is no CPU path)" % (path, e))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BtrapzError("HIP library %s is missing: run spectral_amd.native.build() "
                              "(there is no CPU path)" % LIB_PATH)
        _bind_hip_runtime()
        l = C.CDLL(LIB_PATH)
        vp, dp, ip, llp = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
        l.btrapz_find_traj.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.POINTER(CParams)]
        l.btrapz_find_traj.restype = C.c_double
        l.btrapz_find_traj_mem.argtypes = [C.c_int, C.POINTER(CTrajInput), C.POINTER(CParams), C.c_int, C.c_void_p,
                                           C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int)]
        l.btrapz_find_traj_mem.restype = C.c_double
        l.btrapz_find_traj_mem_cap.argtypes = [C.c_int, C.POINTER(CTrajInput), C.POINTER(CParams), C.c_int, C.c_void_p,
                                               C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        l.btrapz_find_traj_mem_cap.restype = C.c_double
        l.btrapz_corridor_from_file.argtypes = [C.c_int, C.c_char_p, C.POINTER(CSegment), C.c_int]
        l.btrapz_corridor_from_file.restype = C.c_int
        l.btrapz_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        l.btrapz_destroy.argtypes = [C.c_void_p]
        l.btrapz_last_error.argtypes = [C.c_void_p]; l.btrapz_last_error.restype = C.c_char_p
        l.btrapz_device_count.restype = C.c_int
        l.btrapz_find_traj_last_iterations.restype = C.c_int
        l.btrapz_find_traj_last_status.argtypes = [C.c_void_p]; l.btrapz_find_traj_last_status.restype = C.c_int
        l.btrapz_options_init.argtypes = [C.POINTER(COptions)]; l.btrapz_options_init.restype = None
        l.btrapz_rescue_violations_device.argtypes = [vp, C.c_int, dp, vp]
        l.btrapz_last_solve_form.argtypes = [vp]; l.btrapz_last_solve_form.restype = C.c_int
        l.btrapz_workspace_bytes.argtypes = [vp]; l.btrapz_workspace_bytes.restype = C.c_longlong
        l.btrapz_debug_mqm_tables.argtypes = [vp, C.POINTER(CShared), dp, dp]
        l.btrapz_solve_batch_device.argtypes = [vp, C.POINTER(CShared), C.POINTER(COptions), C.c_int, C.c_int,
                                                dp, dp, dp, dp, dp, dp, ip, ip, vp]
        l.btrapz_argmin_device.argtypes = [vp, C.c_int, C.c_int, C.c_longlong, dp, llp, dp, vp]
        l.btrapz_argmin_pairs_device.argtypes = [vp, C.c_int, C.c_int, llp, dp, llp, vp]
        l.btrapz_sample_device.argtypes = [vp, C.c_int, C.c_int, C.c_double, dp, dp, dp, C.c_int, llp, C.c_int,
                                           dp, ip, vp]
        l.btrapz_solve_batch_host.argtypes = [vp, C.POINTER(CShared), C.POINTER(COptions), C.c_int, C.c_int,
                                              dp, dp, dp, dp, dp, dp, ip, ip]
        l.btrapz_solve_ragged_device.argtypes = [vp, C.POINTER(CShared), C.POINTER(COptions), C.c_int, C.c_int,
                                                 dp, ip, dp, dp, dp, dp, dp, ip, ip, vp]
        l.btrapz_corridor_batch_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                                   dp, dp, dp, dp, dp, dp, C.c_int, dp, ip, dp, dp, vp]
        l.btrapz_sample_ragged_device.argtypes = [vp, C.c_int, C.c_int, ip, C.c_double, dp, dp, dp, C.c_int, llp,
                                                  C.c_int, dp, ip, vp]
        l.btrapz_solve_warm_device.argtypes = [vp, C.POINTER(CShared), C.POINTER(COptions), C.POINTER(CWarm), C.c_int,
                                               C.c_int, dp, ip, dp, dp, dp, dp, dp, ip, ip, vp]
        l.btrapz_eval_states_device.argtypes = [vp, C.c_int, C.c_int, ip, dp, dp, C.c_int, dp, dp, vp]
        l.btrapz_prism_bounds_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(CRoad), dp, C.c_int, dp, dp, ip, vp]
        l.btrapz_prism_corridor_batch_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(CRoad), dp, C.c_int,
                                                         C.c_double, dp, dp, dp, dp, C.c_int, dp, ip, dp, dp, ip, vp]
        l.btrapz_multi_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int]
        l.btrapz_multi_destroy.argtypes = [vp]
        l.btrapz_multi_last_error.argtypes = [vp]; l.btrapz_multi_last_error.restype = C.c_char_p
        l.btrapz_multi_transport.argtypes = [vp]
        l.btrapz_multi_transport_library.argtypes = [vp]; l.btrapz_multi_transport_library.restype = C.c_char_p
        l.btrapz_multi_device_count.argtypes = [vp]
        l.btrapz_multi_shard_bounds.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.btrapz_multi_upload.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp]
        l.btrapz_multi_set_shards.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(CMultiShard)]
        l.btrapz_multi_solve_argmin.argtypes = [vp, C.POINTER(CShared), C.POINTER(COptions)]
        l.btrapz_multi_result.argtypes = [vp, C.c_int, llp, dp, dp]
        l.btrapz_multi_wait.argtypes = [vp]
        l.btrapz_multi_shard_view.argtypes = [vp, C.c_int, C.POINTER(CMultiView)]
        l.btrapz_multi_download.argtypes = [vp, dp, dp, ip, ip]
        _lib = l
    return _lib


def multi_shard_bounds(B, G, g, group=0):
    """btrapz_multi_shard_bounds: [lo, hi) of device slot g."""
    lo, hi = C.c_int(0), C.c_int(0)
    rc = lib().btrapz_multi_shard_bounds(int(B), int(G), int(g), int(group), C.byref(lo), C.byref(hi))
    if rc != 0:
        raise BtrapzError("btrapz_multi_shard_bounds(%d, %d, %d, %d) -> %d" % (B, G, g, group, rc))
    return lo.value, hi.value


class MultiContext:
    """btrapz_multi: one host process, one context + stream per entry of `devices` (an ordinal may repeat: logical
    devices), candidates sharded contiguously, one gather of the local winners (include/btrapz_hip.h)."""

    def __init__(self, devices, transport=MULTI_AUTO):
        self._h = C.c_void_p()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = lib().btrapz_multi_create(C.byref(self._h), devs, len(devices), int(transport))
        if rc != 0:
            raise BtrapzError("btrapz_multi_create(%s, transport=%d) failed with %d (no HIP device, or the transport "
                              "cannot be had: see stderr)" % (list(devices), transport, rc))
        self.G = len(devices); self.B = self.S = 0; self.group = 0

    def close(self):
        if self._h:
            lib().btrapz_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise BtrapzError("%s failed (%d): %s" % (what, rc, lib().btrapz_multi_last_error(self._h).decode()))

    def transport(self):
        return int(lib().btrapz_multi_transport(self._h))

    def transport_library(self):
        return lib().btrapz_multi_transport_library(self._h).decode()

    def fallback_reason(self):
        return lib().btrapz_multi_last_error(self._h).decode()

    def upload(self, batch, group=0):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        seg, init, ref_end, dlb = f(batch.seg), f(batch.init), f(batch.ref_end), f(batch.dl_bounds)
        assert seg.shape == (L.NUM_SEG_FIELDS, batch.B, batch.S)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(lib().btrapz_multi_upload(self._h, batch.B, batch.S, int(group), p(seg), p(init), p(ref_end), p(dlb)), "btrapz_multi_upload")
        self.B, self.S, self.group = batch.B, batch.S, int(group)

    def set_shards(self, B, S, shards, group=0):
        """shards: list of (B_g, index_base, seg, init, ref_end, dl_bounds) with torch tensors on the slot's device (or
        None for an empty shard); the tensors must stay alive while steps run."""
        arr = (CMultiShard * self.G)()
        ptr = lambda t: t.data_ptr() if t is not None else None
        for g, (Bg, base, seg, init, ref_end, dlb) in enumerate(shards):
            arr[g] = CMultiShard(int(Bg), int(base), ptr(seg), ptr(init), ptr(ref_end), ptr(dlb))
        self._check(lib().btrapz_multi_set_shards(self._h, int(B), int(S), int(group), arr), "btrapz_multi_set_shards")
        self.B, self.S, self.group = int(B), int(S), int(group)
        self._keep = shards

    def solve_argmin(self, shared, **options):
        sh = CShared.from_shared(shared)
        opt = _options(options.pop("max_iter", 0), options.pop("eps", 0.0), options.pop("elastic", 0), options.pop("elastic_tol", 0.0), **options)
        self._check(lib().btrapz_multi_solve_argmin(self._h, C.byref(sh), C.byref(opt)), "btrapz_multi_solve_argmin")

    def prepared_step(self, shared, **options):
        """solve_argmin with its argument structs built once: returns a function of no arguments (timing loops)."""
        sh = CShared.from_shared(shared)
        opt = _options(options.pop("max_iter", 0), options.pop("eps", 0.0), options.pop("elastic", 0), options.pop("elastic_tol", 0.0), **options)
        fn, check, h = lib().btrapz_multi_solve_argmin, self._check, self._h
        args = (h, C.byref(sh), C.byref(opt))

        def call(_keep=(sh, opt)):
            check(fn(*args), "btrapz_multi_solve_argmin")
        return call

    def result(self, device_slot=-1):
        """(best_idx, best_cost, best_ctrl): scalars + [12 S] for one arg-min group, arrays [n], [n], [n, 12 S] for n groups."""
        n = 1 if (self.group == 0 or self.group >= self.B) else self.B // self.group
        idx = np.zeros(n, dtype=np.int64); cost = np.zeros(n); ctrl = np.zeros((n, 12 * self.S))
        self._check(lib().btrapz_multi_result(self._h, int(device_slot), idx.ctypes.data, cost.ctypes.data, ctrl.ctypes.data), "btrapz_multi_result")
        return (int(idx[0]), float(cost[0]), ctrl[0]) if n == 1 else (idx, cost, ctrl)

    def wait(self):
        self._check(lib().btrapz_multi_wait(self._h), "btrapz_multi_wait")

    def view(self, device_slot):
        v = CMultiView()
        self._check(lib().btrapz_multi_shard_view(self._h, int(device_slot), C.byref(v)), "btrapz_multi_shard_view")
        return v

    def download(self):
        ctrl = np.zeros((self.B, 12 * self.S)); cost = np.zeros(self.B)
        status = np.zeros(self.B, dtype=np.int32); iters = np.zeros(self.B, dtype=np.int32)
        self._check(lib().btrapz_multi_download(self._h, ctrl.ctypes.data, cost.ctypes.data, status.ctypes.data, iters.ctypes.data), "btrapz_multi_download")
        return dict(ctrl=ctrl, cost=cost, status=status, iters=iters)


class Context:
    """Owns a btrapz_ctx on one HIP device."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().btrapz_create(C.byref(self._h), int(device))
        if rc != 0:
            raise BtrapzError("btrapz_create(device=%d) failed with %d: no HIP device "
                              "(this library has no CPU path)" % (device, rc))
        self.device = device

    def close(self):
        if self._h:
            lib().btrapz_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise BtrapzError("%s failed (%d): %s" % (what, rc, lib().btrapz_last_error(self._h).decode()))

    # ---- host-pointer path (numpy in, numpy out) ------------------------------------------
    def solve_host(self, batch, shared, max_iter=0, eps=0.0, elastic=0, elastic_tol=0.0, split=0):
        B, S = batch.B, batch.S
        seg = np.ascontiguousarray(batch.seg, dtype=np.float64)
        init = np.ascontiguousarray(batch.init, dtype=np.float64)
        ref_end = np.ascontiguousarray(batch.ref_end, dtype=np.float64)
        dlb = np.ascontiguousarray(batch.dl_bounds, dtype=np.float64)
        assert seg.shape == (L.NUM_SEG_FIELDS, B, S)
        ctrl = np.empty((B, 12 * S)); cost = np.empty(B)
        status = np.empty(B, dtype=np.int32); iters = np.empty(B, dtype=np.int32)
        sh = CShared.from_shared(shared); opt = _options(max_iter, eps, elastic, elastic_tol, split=split)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(lib().btrapz_solve_batch_host(self._h, C.byref(sh), C.byref(opt), B, S, p(seg), p(init),
                                                  p(ref_end), p(dlb), p(ctrl), p(cost), p(status), p(iters)),
                    "btrapz_solve_batch_host")
        return ctrl, cost, status, iters

    # ---- device-pointer path (torch tensors only carry the memory) --------------------------
    def solve_device(self, B, S, shared, seg, init, ref_end, dl_bounds, ctrl, cost, status, iters=None,
                     stream=None, max_iter=0, eps=0.0, elastic=0, elastic_tol=0.0, queue=0, split=0, start=0, cap_iter=0, lean=0, compact=0):
        sh = CShared.from_shared(shared); opt = _options(max_iter, eps, elastic, elastic_tol, queue=queue, split=split, start=start, cap_iter=cap_iter, lean=lean, compact=compact)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self._check(lib().btrapz_solve_batch_device(self._h, C.byref(sh), C.byref(opt), B, S, ptr(seg), ptr(init),
                                                    ptr(ref_end), ptr(dl_bounds), ptr(ctrl), ptr(cost),
                                                    ptr(status), ptr(iters), C.c_void_p(stream or 0)),
                    "btrapz_solve_batch_device")

    def prepared_solve(self, B, S, shared, seg, init, ref_end, dl_bounds, ctrl, cost, status, iters=None, stream=None,
                       **options):
        """btrapz_solve_batch_device with its argument structs built ONCE: returns a function of no arguments that
        makes the call (a replanning loop or a latency measurement pays the C call, not the marshalling).  The tensors
        and `stream` must stay alive and unchanged in place; options as in solve_device."""
        sh = CShared.from_shared(shared); opt = _options(options.pop("max_iter", 0), options.pop("eps", 0.0), options.pop("elastic", 0),
                                                         options.pop("elastic_tol", 0.0), **options)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        fn, check = lib().btrapz_solve_batch_device, self._check
        args = (self._h, C.byref(sh), C.byref(opt), B, S, ptr(seg), ptr(init), ptr(ref_end), ptr(dl_bounds), ptr(ctrl), ptr(cost),
                ptr(status), ptr(iters), C.c_void_p(stream or 0))
        keep = (sh, opt, seg, init, ref_end, dl_bounds, ctrl, cost, status, iters)

        def call(_keep=keep):
            check(fn(*args), "btrapz_solve_batch_device")
        return call

    def solve_ragged_device(self, B, seg_stride, shared, seg, seg_count, init, ref_end, dl_bounds, ctrl, cost,
                            status, iters=None, stream=None, max_iter=0, eps=0.0, elastic=0, elastic_tol=0.0, cap_iter=0, lean=0, compact=0):
        sh = CShared.from_shared(shared); opt = _options(max_iter, eps, elastic, elastic_tol, cap_iter=cap_iter, lean=lean, compact=compact)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self._check(lib().btrapz_solve_ragged_device(self._h, C.byref(sh), C.byref(opt), B, seg_stride, ptr(seg),
                                                     ptr(seg_count), ptr(init), ptr(ref_end), ptr(dl_bounds),
                                                     ptr(ctrl), ptr(cost), ptr(status), ptr(iters),
                                                     C.c_void_p(stream or 0)), "btrapz_solve_ragged_device")

    def solve_warm_device(self, B, seg_stride, shared, seg, seg_count, init, ref_end, dl_bounds, ctrl, cost, status,
                          iters=None, x0=None, lam0=None, lam_out=None, mu0=0.0, smin=0.0, stream=None, max_iter=0,
                          eps=0.0, hint=None, elastic=0, elastic_tol=0.0, lean=0):
        """btrapz_solve_warm_device: seg_count None = uniform batch; x0 / lam0 / lam_out optional."""
        sh = CShared.from_shared(shared); opt = _options(max_iter, eps, elastic, elastic_tol, lean=lean)
        raw = lambda t: t.data_ptr() if t is not None else None
        warm = CWarm(raw(x0), raw(lam0), raw(lam_out), float(mu0), float(smin), raw(hint))
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self._check(lib().btrapz_solve_warm_device(self._h, C.byref(sh), C.byref(opt), C.byref(warm), B, seg_stride,
                                                   ptr(seg), ptr(seg_count), ptr(init), ptr(ref_end), ptr(dl_bounds),
                                                   ptr(ctrl), ptr(cost), ptr(status), ptr(iters),
                                                   C.c_void_p(stream or 0)), "btrapz_solve_warm_device")

    def workspace_bytes(self):
        """btrapz_workspace_bytes: device memory the context holds for its launches right now."""
        return int(lib().btrapz_workspace_bytes(self._h))

    def last_solve_form(self):
        """btrapz_last_solve_form: 0 packed, 1 split, 2 long, 3 capped + resume, 4 queue; + 8: the two-wavefronts-per-SIMD form."""
        return int(lib().btrapz_last_solve_form(self._h))

    def rescue_violations_device(self, B, viol, stream=None):
        """btrapz_rescue_violations_device: viol [B][4] (position, velocity, acceleration, jerk rows) of the last solve
        with elastic != 0."""
        self._check(lib().btrapz_rescue_violations_device(self._h, int(B), C.c_void_p(viol.data_ptr()),
                                                          C.c_void_p(stream or 0)), "btrapz_rescue_violations_device")

    def debug_axis_records(self, B):
        """(iters [B, 2], status [B, 2]) of the axis problems of the last batched solve."""
        it = np.zeros((B, 2), dtype=np.int32); st = np.zeros((B, 2), dtype=np.int32)
        lib().btrapz_debug_axis_records.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self._check(lib().btrapz_debug_axis_records(self._h, B, it.ctypes.data, st.ctypes.data), "btrapz_debug_axis_records")
        return it, st

    def debug_resume_keys(self, B):
        """keys [2, B] of the last capped solve's resume launch (0: not handed over)."""
        k = np.zeros((2, B), dtype=np.int32)
        lib().btrapz_debug_resume_keys.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._check(lib().btrapz_debug_resume_keys(self._h, B, k.ctypes.data), "btrapz_debug_resume_keys")
        return k

    def debug_mqm_tables(self, shared):
        sh = CShared.from_shared(shared)
        h = np.zeros(168); d = np.zeros(168)
        self._check(lib().btrapz_debug_mqm_tables(self._h, C.byref(sh), h.ctypes.data, d.ctypes.data), "btrapz_debug_mqm_tables")
        return h, d

    def eval_states_device(self, B, seg_stride, seg_count, seg, ctrl, n_times, times, x, stream=None):
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self._check(lib().btrapz_eval_states_device(self._h, B, seg_stride, ptr(seg_count), ptr(seg), ptr(ctrl),
                                                    int(n_times), ptr(times), ptr(x), C.c_void_p(stream or 0)),
                    "btrapz_eval_states_device")

    def corridor_batch_device(self, variant, B, N, num_obs, delta, s_bounds, l_bounds, ds_bounds, dl_bounds_knots,
                              s_ref, l_ref, seg_stride, seg, seg_count, ref_end, dl_bounds, stream=None):
        ptr = lambda t: C.c_void_p(t.data_ptr())
        self._check(lib().btrapz_corridor_batch_device(self._h, int(variant), B, N, num_obs, float(delta),
                                                       ptr(s_bounds), ptr(l_bounds), ptr(ds_bounds),
                                                       ptr(dl_bounds_knots), ptr(s_ref), ptr(l_ref), seg_stride,
                                                       ptr(seg), ptr(seg_count), ptr(ref_end), ptr(dl_bounds),
                                                       C.c_void_p(stream or 0)), "btrapz_corridor_batch_device")

    def prism_bounds_device(self, B, P, N, road, prisms, O, s_bounds, l_bounds, n_strips, stream=None):
        ptr = lambda t: C.c_void_p(t.data_ptr())
        self._check(lib().btrapz_prism_bounds_device(self._h, B, P, N, C.byref(road), ptr(prisms), O, ptr(s_bounds),
                                                     ptr(l_bounds), ptr(n_strips), C.c_void_p(stream or 0)),
                    "btrapz_prism_bounds_device")

    def prism_corridor_batch_device(self, variant, B, P, N, road, prisms, O, delta, ds_bounds, dl_bounds_knots, s_ref, l_ref,
                                    seg_stride, seg, seg_count, ref_end, dl_bounds, n_strips=None, stream=None):
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self._check(lib().btrapz_prism_corridor_batch_device(self._h, int(variant), B, P, N, C.byref(road), ptr(prisms), O,
                                                             float(delta), ptr(ds_bounds), ptr(dl_bounds_knots), ptr(s_ref),
                                                             ptr(l_ref), seg_stride, ptr(seg), ptr(seg_count), ptr(ref_end),
                                                             ptr(dl_bounds), ptr(n_strips), C.c_void_p(stream or 0)),
                    "btrapz_prism_corridor_batch_device")

    def sample_ragged_device(self, B, seg_stride, seg_count, delta, seg, init, ctrl, sel, max_points, out, npoints,
                             stream=None):
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self._check(lib().btrapz_sample_ragged_device(self._h, B, seg_stride, ptr(seg_count), float(delta), ptr(seg),
                                                      ptr(init), ptr(ctrl), int(sel.numel()), ptr(sel),
                                                      int(max_points), ptr(out), ptr(npoints),
                                                      C.c_void_p(stream or 0)), "btrapz_sample_ragged_device")

    def argmin_device(self, B, group, index_base, cost, best_idx, best_cost, stream=None):
        ptr = lambda t: C.c_void_p(t.data_ptr())
        self._check(lib().btrapz_argmin_device(self._h, B, group, int(index_base), ptr(cost), ptr(best_idx),
                                               ptr(best_cost), C.c_void_p(stream or 0)), "btrapz_argmin_device")

    def argmin_pairs_device(self, world, n, pairs, best_cost, best_idx, stream=None):
        ptr = lambda t: C.c_void_p(t.data_ptr())
        self._check(lib().btrapz_argmin_pairs_device(self._h, int(world), int(n), ptr(pairs), ptr(best_cost), ptr(best_idx),
                                                     C.c_void_p(stream or 0)), "btrapz_argmin_pairs_device")

    def sample_device(self, B, S, delta, seg, init, ctrl, sel, max_points, out, npoints, stream=None):
        ptr = lambda t: C.c_void_p(t.data_ptr())
        self._check(lib().btrapz_sample_device(self._h, B, S, float(delta), ptr(seg), ptr(init), ptr(ctrl),
                                               int(sel.numel()), ptr(sel), int(max_points), ptr(out), ptr(npoints),
                                               C.c_void_p(stream or 0)), "btrapz_sample_device")


def find_traj_native(variant, params, input_path=None, output_path=None):
    """btrapz_find_traj(): the reference's find_traj with explicit paths."""
    cp = params if isinstance(params, CParams) else CParams(*params)
    enc = lambda s: os.fsencode(s) if s else None
    return lib().btrapz_find_traj(int(variant), enc(input_path), enc(output_path), C.byref(cp))


def find_traj_last_status():
    """(status, viol[4]) of the calling thread's last find_traj / find_traj_mem call (btrapz_find_traj_last_status)."""
    v = (C.c_double * 4)()
    st = lib().btrapz_find_traj_last_status(C.cast(v, C.c_void_p))
    return int(st), np.array(v[:])


class TrajCall:
    """btrapz_find_traj_mem() on candidate b of a spectral_amd.knots.KnotBatch, prepared once and callable many times
    (a replanning loop, a latency measurement): the input struct and the output buffers are built here, a call is the
    C function and two slices.  The arrays of `kb` are referenced, not copied, when they are contiguous float64."""

    def __init__(self, variant, params, kb, b=0, cap=None):
        self.variant = int(variant)
        self.cp = params if isinstance(params, CParams) else CParams(*params)
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self._arrs = [f(kb.s_bounds[b]), f(kb.l_bounds[b]), f(kb.ds_bounds[b]), f(kb.dl_bounds[b]), f(kb.s_ref[b]), f(kb.l_ref[b])]
        h = kb.header
        self.ti = CTrajInput(int(kb.N), int(kb.num_obs), float(kb.delta), (C.c_double * 3)(*kb.init[b, :3]),
                             (C.c_double * 3)(*kb.init[b, 3:]), float(h["ds_ref"]), float(h["dl_ref"]),
                             (C.c_double * 2)(*h["dds"]), (C.c_double * 2)(*h["ddds"]), (C.c_double * 2)(*h["ddl"]),
                             (C.c_double * 2)(*h["dddl"]), *[a.ctypes.data for a in self._arrs])
        self.cap = int(cap if cap is not None else 4 * kb.N + 16)
        self.traj = np.zeros((7, self.cap)); self.ctrl = np.zeros(12 * 256)
        self.n, self.S = C.c_int(0), C.c_int(0)
        self._fn = lib().btrapz_find_traj_mem_cap      # (control points of up to 256 segments: the buffer's size goes along)
        self._args = (self.variant, C.byref(self.ti), C.byref(self.cp), self.cap, self.traj.ctypes.data, C.byref(self.n),
                      self.ctrl.ctypes.data, self.ctrl.size, C.byref(self.S))

    def __call__(self, copy=True):
        """(cost, traj [7][n] rows t s l ds dl dds ddl, ctrl [12 S]); cost == 1e11 on failure (traj, ctrl None).
        copy=False returns views of the call's own buffers (overwritten by the next call)."""
        cost = self._fn(*self._args)
        if cost == 100000000000.0:
            return cost, None, None
        traj, ctrl = self.traj[:, :min(self.n.value, self.cap)], self.ctrl[:12 * self.S.value]
        return (cost, traj.copy(), ctrl.copy()) if copy else (cost, traj, ctrl)


def find_traj_mem(variant, params, kb, b=0, cap=None):
    """btrapz_find_traj_mem(): find_traj on candidate b of a spectral_amd.knots.KnotBatch (arrays in, arrays out).
    Returns (cost, traj [7][n] rows t s l ds dl dds ddl, ctrl [12 S]); cost == 1e11 on failure (traj, ctrl None)."""
    return TrajCall(variant, params, kb, b, cap)()


def corridor_from_file(variant, input_path, cap=256):
    """Host-side corridor stage (no GPU): list of CSegment, [] when nothing is selected."""
    buf = (CSegment * cap)()
    n = lib().btrapz_corridor_from_file(int(variant), os.fsencode(input_path), buf, cap)
    if n < 0:
        raise BtrapzError("btrapz_corridor_from_file(%s) -> %d" % (input_path, n))
    return [buf[i] for i in range(min(n, cap))]
