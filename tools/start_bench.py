#!/usr/bin/env python3
"""btrapz_options.start = 1 (one Newton step of the unconstrained problem before the first iteration) against the
default start (constant-velocity propagation of the initial state): iterations, kernel time, accepted candidates,
on the bench batches.  One JSON object on stdout.

    python tools/start_bench.py [--batch 65536]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    import torch
    from spectral_amd import synth
    from spectral_amd.solver import BatchSolver
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    a = ap.parse_args(argv)
    solver = BatchSolver(0)
    dev = torch.device("cuda:0")
    out = {}
    cases = [("scenario1 x 20 trapezoid", lambda: synth.make_scenario1_batch(a.batch, 20, 0)),
             ("generic x 20 trapezoid", lambda: synth.make_batch(a.batch, 20, config=3)),
             ("scenario1 x 20 cuboid", lambda: synth.make_scenario1_batch(a.batch, 20, 1)),
             ("scenario1 x 10 trapezoid", lambda: synth.make_scenario1_batch(a.batch, 10, 0))]
    for label, make in cases:
        batch, sh = make()
        db = solver.upload(batch)
        rec = {}
        keep = {}
        for start in (0, 1):
            for _ in range(2):
                o = solver.solve(db, sh, start=start)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                o = solver.solve(db, sh, start=start)
            e1.record(); torch.cuda.synchronize(dev)
            st = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy() + 1
            ok = (st == 1) | (st == 2)
            keep[start] = (ok, o["ctrl"].cpu().numpy().copy())
            rec["start_%d" % start] = {"kernel_ms": e0.elapsed_time(e1) / 5, "mean_iterations": float(it.mean()),
                                       "mean_iterations_solved": float(it[ok].mean()), "solved": int(ok.sum())}
        both = keep[0][0] & keep[1][0]
        x0, x1 = keep[0][1][both], keep[1][1][both]
        rec["only_default_solves"] = int((keep[0][0] & ~keep[1][0]).sum()); rec["only_start_1_solves"] = int((~keep[0][0] & keep[1][0]).sum())
        rec["max_rel_ctrl_diff"] = float((np.abs(x0 - x1).max(axis=1) / np.abs(x0).max(axis=1)).max())
        rec["time_ratio"] = rec["start_1"]["kernel_ms"] / rec["start_0"]["kernel_ms"]
        out[label] = rec
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
