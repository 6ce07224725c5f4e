"""Python-side mirror of the reference's src/trp_wrapper.py (trapezoid-prism driver).

Same names and call shapes -- Params, Metrics, find_traj() -> bool, run_btrapz(trial) ->
float -- but bound to this repo's libtrp.so (HIP path).  The reference hard-codes its
library, weights and data paths (trp_wrapper.py:45,100); here they default to the same
literals and can be redirected:
    BTRAPZ_LIB_DIR        directory holding libtrp.so / libcub.so      (default: spectral_amd/lib)
    BTRAPZ_WEIGHTS        weights file, one tab-separated row of 10     (default: reference literal)
    BTRAPZ_INPUT          corridor text file                          (read by the library)
    BTRAPZ_OUTPUT_PREFIX  trajectory file prefix, "<prefix><iteration>.txt" (read by the library)
"""
import os
from ctypes import CDLL, POINTER, Structure, c_double, c_int

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libtrp.so"
DEFAULT_WEIGHTS = "/home/srujan_d/RISS/code/btrapz/src/weights.txt"  # trp_wrapper.py:100


class Metrics(Structure):  # trp_wrapper.py:7-17 (declared, unused by the live ABI)
    _fields_ = [("s_avg_acc", c_double), ("l_avg_acc", c_double), ("s_max_acc", c_double),
                ("l_max_acc", c_double), ("a_cost", c_double)]


class Params(Structure):  # trp_wrapper.py:19-32 == include/btrapz/py_cpp_.h:6-21
    _fields_ = [("s_acc_weight", c_double), ("s_jerk_weight", c_double), ("l_acc_weight", c_double),
                ("l_jerk_weight", c_double), ("weight_s_ref", c_double), ("weight_ds_ref", c_double),
                ("weight_l_ref", c_double), ("weight_dl_ref", c_double), ("weight_end_s", c_double),
                ("weight_end_l", c_double), ("iteration", c_int)]


_cdll = None


def _library():
    """CDLL on first use (the reference loads at import time, trp_wrapper.py:45)."""
    global _cdll
    if _cdll is None:
        path = os.path.join(os.environ.get("BTRAPZ_LIB_DIR", os.path.join(_HERE, "lib")), LIB_NAME)
        _cdll = CDLL(path)
        _cdll.find_traj.argtypes = (POINTER(Params),)
        _cdll.find_traj.restype = c_double
    return _cdll


def _run_btrapz(params):
    return _library().find_traj(params)


def read_weights(path=None):
    path = path or os.environ.get("BTRAPZ_WEIGHTS", DEFAULT_WEIGHTS)
    with open(path) as f:
        return [float(v) for v in f.readlines()[0].split("\t")[:10]]


def run_btrapz(trial):
    """Optuna objective of the reference (trp_wrapper.py:56-97): suggests the ten weights."""
    names = ("s_acc_weight", "s_jerk_weight", "l_acc_weight", "l_jerk_weight", "weight_s_ref", "weight_ds_ref",
             "weight_l_ref", "weight_dl_ref", "weight_end_s", "weight_end_l")
    w = [trial.suggest_float(n, 0, 50) for n in names]
    return _run_btrapz(Params(*w, 1))


def find_traj(weights=None, iteration=3):
    """trp_wrapper.py:99-121: True unless the optimizer failed (sentinel 1e11)."""
    w = weights if weights is not None else read_weights()
    a = _run_btrapz(Params(*[float(v) for v in w[:10]], int(iteration)))
    return False if a == 100000000000 else True
