"""The host corridor stage (corridor_core.h through btrapz_corridor_from_file) against the oracle on random inputs,
including the ones that are garbage in the reference's own terms (nan / inf bounds, collapsed or crossing bounds,
horizons of three knots): the counts and every field are the reference's arithmetic, operation by operation."""
import numpy as np
import pytest

from helpers import fuzz_knot_batch, oracle_corridor
from spectral_amd import knots, native

ATTRS = ("beg_t", "end_t", "t", "down_bias", "down_skew", "upp_bias", "upp_skew", "l_down_bias", "l_down_skew", "l_upp_bias",
         "l_upp_skew", "beg_l", "end_l")
# (seed, candidate): the three disagreements the device fuzz found first, then a spread of shapes
CASES = [(33, 12), (37, 8), (1068, 8), (191, 5), (2060, 12), (3110, 13)] + [(s, s % 16) for s in range(400, 460)]


def same(a, b):
    return a == b or (isinstance(a, float) and np.isnan(a) and np.isnan(b))


@pytest.mark.parametrize("variant", [0, 1])
def test_host_corridor_stage_equals_the_oracle_on_random_inputs(variant, tmp_path):
    checked = usable = 0
    for seed, b in CASES:
        kb = fuzz_knot_batch(seed)
        n, cubes = oracle_corridor(kb, b, variant)
        if n is None:
            continue
        path = str(tmp_path / "c.txt")
        knots.write_corridor_file(path, kb, b)
        back = knots.parse_corridor_file(path)                      # the writer loses nothing
        assert np.array_equal(back.s_bounds[0], kb.s_bounds[b], equal_nan=True) and np.array_equal(back.l_ref[0], kb.l_ref[b])
        try:
            segs = native.corridor_from_file(variant, path, cap=512)
        except native.BtrapzError:
            segs = None
        checked += 1
        if n <= 0:
            assert not segs, (seed, b, n, segs and len(segs))
            continue
        assert segs is not None and len(segs) == n, (seed, b, n, None if segs is None else len(segs))
        usable += 1
        for k, (c, s) in enumerate(zip(cubes, segs)):
            for attr in ATTRS:
                assert same(getattr(s, attr), getattr(c, attr)), (seed, b, k, attr, getattr(s, attr), getattr(c, attr))
    assert checked >= 50 and usable >= 25
