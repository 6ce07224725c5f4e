#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz|json from the CPU oracle (run in the build container).

What is committed and where it comes from:
  inputs/*.txt        DATA files copied from the reference repo (src/c*.txt, c_road_s1*.txt,
                      weights.txt, all_weights.txt): corridor inputs in find_traj's grammar.
  ref_outputs/*.txt   DATA files copied from the reference repo (src/s*_3d*.txt): trajectories the
                      reference itself wrote (7 columns, 3 decimals).  s4_slt_3d / s4_cub_3d /
                      s5_slt_3d (input c4/c5) and the s columns of s2_slt_3d_{4,5} (input c2) are
                      reproduced by the oracle to print precision with the s weights of weights.txt:
                      these are reference-generated golden vectors.
  corridors.json      new_corridor lists of the oracle's pipeline for every bundled input.
  scenario_xstar.npz  oracle x* (dense interior point) + OSQP-port solution per scenario.
  synthetic_xstar.npz oracle x* for the first candidates of the synthetic configs.
  scenario1_xstar.npz oracle x* for the first candidates of the scenario_1-shaped batches (configs 3 / 4).
The reference itself cannot be run here (needs Eigen + OSQP), so the last three are produced by
this repo's oracle; the parity tests compare the HIP path with them on the GPU box, where
/root/reference does not exist.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.environ.get("GOLDEN_OUT", HERE)   # (tests regenerate into a temporary directory and compare)
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from spectral_amd import synth  # noqa: E402

SCENARIOS = ["c1", "c2", "c3", "c4", "c6", "c7", "c7_7", "c_road_s1", "c_road_s1_2", "c_road_s1_3"]


def main():
    w = np.loadtxt(os.path.join(HERE, "inputs", "weights.txt"))
    p = O.params_from_weights(w)
    corr, arrays = {}, {}
    for name in SCENARIOS:
        for v in (0, 1):
            inp = O.ParsedInput(os.path.join(HERE, "inputs", name + ".txt"))
            n, cubes = O.pipeline(v, inp)
            key = "%s/%d" % (name, v)
            corr[key] = {"S": n, "cubes": [[getattr(c, f) for f, _ in O.Cube._fields_] for c in cubes]}
            if n < 1:
                continue
            qp = O.AssembledQp(v, cubes, p, inp)
            xs, ys, info = qp.solve_exact()
            xo, yo, io = qp.solve()
            arrays[key + "/xstar"] = xs
            arrays[key + "/xstar_status"] = np.array([info.status, info.iter])
            arrays[key + "/xstar_obj"] = np.array([info.obj_val])
            arrays[key + "/osqp"] = xo
            arrays[key + "/osqp_status"] = np.array([io.status, io.iter])
            if info.status == 1:
                rc, samp = O.sample(cubes, inp.delta, xs, inp.init_s, inp.init_l)
                arrays[key + "/traj"] = np.stack(samp) if rc == 0 else np.zeros((6, 0))
    json.dump({"fields": [f for f, _ in O.Cube._fields_], "corridors": corr},
              open(os.path.join(OUT, "corridors.json"), "w"), indent=0)
    np.savez_compressed(os.path.join(OUT, "scenario_xstar.npz"), **arrays)
    syn = {}
    for cfg, S, variant, nb in [(2, 10, 0, 32), (3, 20, 0, 32), (4, 20, 1, 32), (9, 7, 0, 16), (8, 13, 1, 16)]:
        batch, sh = synth.make_batch(256, S, config=cfg, variant=variant)
        ctrl, obj, st, it = O.batch_solve(batch, sh, 0, nb, exact=True, threads=8)
        syn["cfg%d/xstar" % cfg] = ctrl
        syn["cfg%d/obj" % cfg] = obj
        syn["cfg%d/status" % cfg] = st
        syn["cfg%d/meta" % cfg] = np.array([256, S, variant, nb])
    np.savez_compressed(os.path.join(OUT, "synthetic_xstar.npz"), **syn)
    print("wrote corridors.json, scenario_xstar.npz, synthetic_xstar.npz")
    scenario1()


def scenario1():
    """x* of the first candidates of the scenario_1-shaped batches (BASELINE configs 3 and 4, synth.make_scenario1_batch)."""
    s1 = {}
    for S, variant, nb in [(20, 0, 48), (20, 1, 48), (10, 0, 24)]:
        batch, sh = synth.make_scenario1_batch(256, S, variant)
        ctrl, obj, st, it = O.batch_solve(batch, sh, 0, nb, exact=True, threads=8)
        key = "S%d_v%d" % (S, variant)
        s1[key + "/xstar"] = ctrl; s1[key + "/obj"] = obj; s1[key + "/status"] = st
        s1[key + "/meta"] = np.array([256, S, variant, nb])
    np.savez_compressed(os.path.join(OUT, "scenario1_xstar.npz"), **s1)
    print("wrote scenario1_xstar.npz")


if __name__ == "__main__":
    scenario1() if "--scenario1" in sys.argv else main()
