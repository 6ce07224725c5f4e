#!/usr/bin/env python3
"""btrapz_options.cap_iter (a capped first launch + a resume launch over the unfinished axis problems) against the
one-launch solve on the bench batches: time of the whole solve call by HIP events, and whether the results are the
one-launch solve's bit for bit.  One JSON object on stdout.

    python tools/cap_bench.py [--batch 65536] [--caps 8,9,10,11,12]
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
    ap.add_argument("--caps", default="4,5,6,7,8,10")
    ap.add_argument("--lean", type=int, default=0, help="btrapz_options.lean (0 automatic, 1 the two-wavefronts-per-SIMD form, -1 the packed form)")
    a = ap.parse_args(argv)
    caps = [int(c) for c in a.caps.split(",")]
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
        ref = None
        for cap in [-1] + caps:       # -1: the one-launch solve (0 would be the library's automatic choice)
            for _ in range(2):
                o = solver.solve(db, sh, split=-1, cap_iter=cap, lean=a.lean)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                o = solver.solve(db, sh, split=-1, cap_iter=cap, lean=a.lean)
            e1.record(); torch.cuda.synchronize(dev)
            res = {k: o[k].cpu().numpy().copy() for k in ("ctrl", "cost", "status", "iters")}
            r = {"solve_ms": e0.elapsed_time(e1) / 5, "form": solver.ctx.last_solve_form()}
            if cap > 0:
                keys = solver.ctx.debug_resume_keys(a.batch)
                r["handed_over_fraction"] = float((keys > 0).mean())
            if cap == -1:
                ref = res
                r["mean_iterations"] = float(res["iters"].mean() + 1)
            else:
                ok = ref["status"] > 0
                r["bit_identical"] = bool(np.array_equal(ref["status"], res["status"]) and np.array_equal(ref["iters"], res["iters"]) and
                                          np.array_equal(ref["ctrl"][ok], res["ctrl"][ok]) and np.array_equal(ref["cost"], res["cost"]))
                r["ratio"] = r["solve_ms"] / rec["one_launch"]["solve_ms"]
            rec["one_launch" if cap == -1 else "cap_%d" % cap] = r
        out[label] = rec
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
