"""Corridor pipeline: oracle restatement vs committed corridors, and the PRODUCT's C++ host
implementation (spectral_amd/csrc/corridor.cpp, reached through the C-ABI without a GPU)
against the oracle, field by field (solve_3d.cc:323-486,488-714,729-772; cuboid_3d.cc)."""
import json
import os

import numpy as np
import pytest

from helpers import O
from spectral_amd import native

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["c1", "c2", "c3", "c4", "c6", "c7", "c7_7", "c_road_s1", "c_road_s1_2", "c_road_s1_3"]
CORR = json.load(open(os.path.join(GOLD, "corridors.json")))


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("variant", [0, 1])
def test_oracle_pipeline_matches_committed_corridors(name, variant):
    inp = O.ParsedInput(os.path.join(GOLD, "inputs", name + ".txt"))
    n, cubes = O.pipeline(variant, inp)
    want = CORR["corridors"]["%s/%d" % (name, variant)]
    assert n == want["S"]
    for c, row in zip(cubes, want["cubes"]):
        assert [getattr(c, f) for f in CORR["fields"]] == row


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("variant", [0, 1])
def test_product_host_pipeline_is_identical_to_oracle(name, variant):
    path = os.path.join(GOLD, "inputs", name + ".txt")
    seg = native.corridor_from_file(variant, path)
    inp = O.ParsedInput(path)
    n, cubes = O.pipeline(variant, inp)
    assert len(seg) == n
    for a, b in zip(seg, cubes):
        for f, _ in native.CSegment._fields_:
            assert getattr(a, f) == getattr(b, f), (name, variant, f)


def test_parser_tolerates_short_last_row():
    """c_road_s1_2.txt's last kappa row has 70 tokens; the reference's unchecked `ifs >>` keeps
    going (trp_wrapper.cpp:140-144).  Both parsers must accept the file."""
    path = os.path.join(GOLD, "inputs", "c_road_s1_2.txt")
    inp = O.ParsedInput(path)
    assert inp.N == 71 and inp.num_obs == 5
    assert len(native.corridor_from_file(0, path)) == 4


def test_missing_file_is_an_error_not_a_crash(tmp_path):
    with pytest.raises(native.BtrapzError):
        native.corridor_from_file(0, str(tmp_path / "nope.txt"))
    with pytest.raises(IOError):
        O.ParsedInput(str(tmp_path / "nope.txt"))


def test_split_peels_one_second_pieces():
    """CorridorSplit: a 7 s corridor becomes 1.0 s / 10-knot pieces (solve_3d.cc:735-770)."""
    N, delta = 71, 0.1
    xb = np.tile([0.0, 50.0], (N, 1)); yb = np.tile([1.0, 3.0], (N, 1))
    cubes = O.corridor_generation(0, N, delta, xb, yb)
    assert [(c.beg_t, c.end_t) for c in cubes[:7]] == [(10 * k, 10 * k + 10) for k in range(7)]
    assert all(abs(c.t - 1.0) < 1e-12 for c in cubes[:7])


def test_empty_selection_is_a_defined_failure():
    """No reference knot inside any cube: the reference underflows temp.size()-1; here -2 / []."""
    N, delta = 31, 0.1
    xb = np.tile([0.0, 50.0], (N, 1)); yb = np.tile([1.0, 3.0], (N, 1))
    cubes = O.corridor_generation(0, N, delta, xb, yb)
    n, out = O.collision_check(0, N, delta, [cubes], np.linspace(0, 10, N), np.full(N, 9.0))
    assert n == -2 and out == []
