"""The product's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer: the
corridor-file scanner and the %.3f trajectory writer (spectral_amd/csrc/corridor.cpp), TokenReader's
reference-compatible failure mode, the host instantiation of corridor_core.h, trajectory_cost (traj_cost.h) and the strip
geometry of prism_core.h, built by g++ without HIP (`make -C spectral_amd/csrc host_asan`) and driven by
spectral_amd/csrc/host_check/host_check.cpp.  The reference reads its text input unchecked
(/root/reference/src/trp_wrapper.cpp:39-144) and reads past x_ref (:221,257,269; src/solve_3d.cc:1161); here every bundled
input, damaged copies of them (cut anywhere, mid-token too; a knot count larger than the data; garbage) and seeded
scenes run clean, and what the sanitized build computes is what the shipped (hipcc-built) library computes."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spectral_amd", "csrc")
BIN = os.path.join(ROOT, "spectral_amd", "lib", "host_check_asan")
INPUTS = os.path.join(ROOT, "tests", "golden", "inputs")
CORRIDOR_FILES = sorted(f for f in os.listdir(INPUTS) if f.startswith("c") and f.endswith(".txt"))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


@pytest.fixture(scope="module")
def check():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["make", "-C", CSRC, "host_asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]

    def run(*args, ok_codes=(0,)):
        p = subprocess.run([BIN, *[str(a) for a in args]], capture_output=True, text=True, env=ENV, timeout=600)
        out = p.stdout + p.stderr
        assert "AddressSanitizer" not in out and "runtime error" not in out and "LeakSanitizer" not in out, out[-3000:]
        assert p.returncode in ok_codes, (p.returncode, out[-2000:])
        return p.stdout
    return run


def segments(text):
    lines = text.splitlines()
    n = int(lines[0].split()[1])
    return n, [l.split() for l in lines[1:1 + max(n, 0)]]


def test_bundled_inputs_run_clean_and_equal_the_shipped_library(check):
    """All 13 corridor files, both variants: clean, and the segments are the ones btrapz_corridor_from_file of the
    shipped library returns (same source, other compiler)."""
    from spectral_amd import native
    if not os.path.exists(native.LIB_PATH):
        native.build()
    assert len(CORRIDOR_FILES) == 13
    for name in CORRIDOR_FILES:
        for variant in (0, 1):
            n, segs = segments(check("corridor", variant, os.path.join(INPUTS, name)))
            ref = native.corridor_from_file(variant, os.path.join(INPUTS, name))
            assert n == len(ref), (name, variant)
            for row, c in zip(segs, ref):
                assert (int(row[0]), int(row[1])) == (c.beg_t, c.end_t)
                got = [float(v) for v in row[2:13]]
                want = [c.t, c.beg_l, c.end_l, c.upp_skew, c.upp_bias, c.down_skew, c.down_bias, c.l_upp_skew, c.l_upp_bias, c.l_down_skew, c.l_down_bias]
                assert got == want, (name, variant)


def test_damaged_corridor_files(check, tmp_path):
    """Cut anywhere (the reference's own c_road_s1_2.txt has a short last row and relies on the stream's failure mode:
    every later read leaves its target untouched), knot count beyond the data, header nonsense, no numbers at all:
    a decision (S -1 unreadable, S 0 nothing selected, S n), never a bad read."""
    rng = np.random.default_rng(7)
    n_runs = 0
    for name in ("c_road_s1_2.txt", "c1.txt", "c7.txt"):
        data = open(os.path.join(INPUTS, name), "rb").read()
        cuts = [0, 1, 2, len(data) // 3, len(data) - 1] + [int(c) for c in rng.integers(0, len(data), 12)]
        for cut in cuts:
            p = tmp_path / "cut.txt"
            p.write_bytes(data[:cut])
            n, _ = segments(check("corridor", int(rng.integers(0, 2)), p))
            assert n >= -1
            n_runs += 1
        toks = data.split()
        for edit in ("bigN", "hugeN", "negN", "zero_delta", "many_obs", "nan", "garbage", "binary"):
            t = list(toks)
            if edit == "bigN": t[0] = str(int(t[0]) * 3).encode()
            if edit == "hugeN": t[0] = b"99999999999"
            if edit == "negN": t[0] = b"-5"
            if edit == "zero_delta": t[1] = b"0"
            if edit == "many_obs": t[8] = b"900"
            if edit == "nan": t[20:40] = [b"nan"] * 10 + [b"-inf"] * 5 + [b"1e999"] * 5
            if edit == "garbage": t[30] = b"12.5.7e+"
            blob = b" ".join(t) if edit != "binary" else bytes(rng.integers(0, 256, 4000, dtype=np.uint8))
            p = tmp_path / "edit.txt"
            p.write_bytes(blob)
            for variant in (0, 1):
                n, _ = segments(check("corridor", variant, p))
                assert n >= -1
                n_runs += 1
    assert n_runs > 80
    assert segments(check("corridor", 0, tmp_path / "does_not_exist.txt"))[0] == -1


def test_trajectory_writer_and_cost_reads(check, tmp_path):
    """The writer on values a failed solve could hand over (1e300, nan, -inf, -0.0005): the text printf("%.3f") writes;
    the cost with fewer, as many and more samples than knots (the reference indexes x_ref by the sample, :221,257)."""
    out = tmp_path / "traj.txt"
    text = check("corridor", 0, os.path.join(INPUTS, "c1.txt"), out)
    assert "write 1" in text and text.count("cost np") == 10
    for line in text.splitlines():
        if line.startswith("cost np"):
            assert np.isfinite(float(line.split()[-1]))
    rows = open(out).read().splitlines()
    first = rows[0].split()
    assert first[0] == "0.000" and first[1] == "%.3f" % 1e300 and first[2] == "%.3f" % -0.0005
    assert rows[1].split()[1] == "nan" and rows[2].split()[1] == "-inf"
    assert all(len(r.split()) == 7 for r in rows)


def test_scanner_and_writer_against_libc(check):
    assert "200000 values, 0 differences" in check("text", 11, 200000)


def test_knot_test_of_a_segments_own_span_equals_the_general_one(check):
    """corridor_core.h::knot_inside_own_span -- what the device's selection evaluates for the knots of a segment's own span
    (round 6: two of the reference's four edge functions, solve_3d.cc:534-581, have their signs decided by the span there) --
    against knot_inside on 1.2 M seeded decisions that meet its preconditions: ordinary corridors, gaps of 1e-300 and
    1e+288, denormal slopes, one-knot spans, references on the edges and one ulp off them."""
    for seed in (3, 4, 5):
        out = check("knots", seed, 100000)
        import re
        n, inside, bad = (int(v) for v in re.match(r"knots (\d+) decisions \((\d+) inside\), (\d+) differences", out).groups())
        assert n > 300000 and 0.2 * n < inside < 0.6 * n and bad == 0, out


def test_prism_geometry_as_64_threads_equals_the_restatement(check):
    """prism_core.h (tables by ballot and barriers, strips per lane) run as 64 host threads per wavefront with a table
    block of exactly prism_tab_bytes(P): clean, and the strips are the CPU restatement's (oracle/prism_oracle.py) on
    every seeded scene -- up to 16 cars, inactive ones among them."""
    from oracle import prism_oracle as P
    text = check("prisms", 5, 40)
    cars = []; n_strips = 0; checked = 0; scenes = 0
    want = None
    for line in text.splitlines():
        w = line.split()
        if w[0] == "scene":
            n_strips = int(w[5]); scenes += 1
        elif w[0] == "cars":
            v = [float(x) for x in w[1:]]
            cars = [dict(centre=(v[i], v[i + 1], v[i + 2]), vel_s=v[i + 3], vel_l=v[i + 4], time=v[i + 5]) for i in range(0, len(v), 8) if v[i + 6] != 0.0]
            want = P.prism_bounds(cars) if cars else P.prism_bounds([])
            assert len(want) == n_strips, (len(want), n_strips)
        elif w[0] == "strip":
            j, i = int(w[1]), int(w[3])
            s_lo, s_hi, l_lo, l_hi = float(w[5]), float(w[6]), float(w[8]), float(w[9])
            sb, lb = want[j]
            assert [s_lo, s_hi] == list(sb[i]) and [l_lo, l_hi] == list(lb[i]), (j, i)
            checked += 1
    assert scenes == 40 and checked > 2000
