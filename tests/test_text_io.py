"""The text side channel of the file-based find_traj (trp_wrapper.cpp:39-144 reads the corridor file with `ifs >> v`,
:288-301 writes the trajectory with fixed 3 decimals): the library scans and writes with fast paths around strtod and
printf("%.3f") -- same values, same text.  Host code only: no GPU needed."""
import ctypes as C
import math

import numpy as np

from spectral_amd import native


def _lib():
    l = native.lib()
    l.btrapz_debug_parse_double.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    l.btrapz_debug_parse_double.restype = C.c_double
    l.btrapz_debug_format_fixed.argtypes = [C.c_double, C.c_char_p]
    return l


def parse(l, text):
    n = C.c_int(-1)
    v = l.btrapz_debug_parse_double(text.encode(), C.byref(n))
    return v, n.value


def test_scanner_equals_strtod():
    l = _lib()
    libc = C.CDLL(None)
    libc.strtod.restype = C.c_double
    libc.strtod.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]

    def strtod(text):
        b = C.create_string_buffer(text.encode())
        end = C.c_void_p()
        v = libc.strtod(C.addressof(b), C.byref(end))
        return v, end.value - C.addressof(b)

    rng = np.random.default_rng(0)
    texts = ["0", "-0", "-0.000", "0.1", " 12.5 rest", "\n\t-3.25e2x", "1e22", "1e23", "9007199254740993", "123456789012345", "1234567890123456",
             "0.000000000000000000001", "1e-22", "1e-23", "5.", ".5", ".", "-", "+", "e5", "1e", "1e+", "1e+5", "1E-05", "inf", "-inf", "nan", "0x10", "0x1p3",
             "1.7976931348623157e308", "1e400", "4.9e-324", "00012.500", "-.5e-3", "1.0000000000000002", "3.14159265358979", "2.5 3.5", "12abc", "1..2", "--1", ""]
    for _ in range(20000):
        kind = rng.integers(0, 4)
        if kind == 0:
            texts.append("%.3f" % rng.uniform(-200, 200))
        elif kind == 1:
            texts.append(repr(float(rng.uniform(-1e3, 1e3))))
        elif kind == 2:
            texts.append("%.*f" % (int(rng.integers(0, 13)), rng.uniform(-1e4, 1e4)))
        else:
            texts.append("%.*e" % (int(rng.integers(0, 17)), rng.uniform(-1, 1) * 10.0 ** rng.integers(-30, 30)))
    for t in texts:
        got, want = parse(l, t), strtod(t)
        assert got[1] == want[1], (t, got, want)
        assert got[0] == want[0] or (math.isnan(got[0]) and math.isnan(want[0])), (t, got, want)
        assert math.copysign(1.0, got[0]) == math.copysign(1.0, want[0]) or math.isnan(want[0]), (t, got, want)


def test_writer_equals_printf():
    l = _lib()
    buf = C.create_string_buffer(336)
    rng = np.random.default_rng(1)
    vals = [0.0, -0.0, 0.0005, -0.0005, 0.0015, 0.0025, 0.0625, -0.0625, 0.1235, 1.0005, 2.5, 1e-9, -1e-9, -0.0004, 0.9995, 0.99949999999, 999.9995,
            1e9, -1e9, 1e15, 1e300, float("inf"), float("-inf"), float("nan"), 123456.7895, 0.5, 1.5, 1234.5675, 8.0005, 16.0005, 0.0004999999999999999]
    vals += [k / 2000.0 for k in range(-4001, 4002)]                      # every tie and near-tie around zero
    vals += [np.nextafter(k / 2000.0, s) for k in range(1, 2001, 2) for s in (-1e9, 1e9)]
    vals += list(rng.uniform(-500, 500, 50000)) + list(rng.uniform(-1e8, 1e8, 20000)) + list(rng.normal(0, 1e-3, 5000))
    vals += [round(float(v), 3) for v in rng.uniform(-100, 100, 20000)] + [round(float(v), 3) + 0.0005 for v in rng.uniform(-100, 100, 20000)]
    for v in vals:
        n = l.btrapz_debug_format_fixed(float(v), buf)
        assert buf.value.decode() == "%.3f" % float(v), (v, buf.value, "%.3f" % float(v))
        assert n == len(buf.value)
