#!/usr/bin/env python3
"""The multi-GPU step of the C-ABI (btrapz_multi_*) timed: one host process, G device slots.  On a one-GPU box the slots are
LOGICAL devices (ordinal 0 repeated): the shards then run on separate streams of the one GPU, which shows (a) what the
sharding, the per-slot launches, the gather and the select cost over the single-context step and (b) what a second
stream hides of a step's tail.  One JSON object on stdout.

    python tools/multi_bench.py [--batch 65536] [--segments 20] [--slots 1,2,3,4,8] [--devices 0,1,...]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    import torch
    from spectral_amd import native, synth
    from spectral_amd.solver import BatchSolver
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--segments", type=int, default=20)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--slots", default="1,2,3,4,8")
    ap.add_argument("--devices", default="", help="real device ordinals (default: logical slots on device 0)")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args(argv)
    batch, sh = synth.make_scenario1_batch(a.batch, a.segments, a.variant)
    out = {"workload": "scenario_1 x %d, %d candidates, variant %d" % (a.segments, a.batch, a.variant)}
    # the single-context step: solve + arg-min on one stream
    solver = BatchSolver(0)
    db = solver.upload(batch)

    def step():
        o = solver.solve(db, sh)
        return solver.argmin(o["cost"])
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        bi, bc = step()
    torch.cuda.synchronize()
    out["single_context_ms"] = 1e3 * (time.perf_counter() - t0) / a.reps
    want = (int(bi[0]), float(bc[0]))
    del solver, db
    devs = [int(x) for x in a.devices.split(",") if x]
    for G in [int(x) for x in a.slots.split(",")]:
        devices = devs[:G] if devs else [0] * G
        if len(devices) < G:
            continue
        m = native.MultiContext(devices, native.MULTI_AUTO)
        m.upload(batch)
        call = m.prepared_step(sh)
        for _ in range(3):
            call(); m.wait()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            call(); m.wait()
        ms = 1e3 * (time.perf_counter() - t0) / a.reps
        bi_, bc_, _ = m.result()
        out["slots_%d" % G] = {"ms_per_step": ms, "solves_per_s": a.batch / (ms * 1e-3), "transport": m.transport(),
                               "devices": devices, "winner_equals_single_context": (bi_, bc_) == want,
                               "shard_sizes": [m.view(g).B for g in range(G)]}
        m.close()
        print("slots", G, json.dumps(out["slots_%d" % G]), file=sys.stderr, flush=True)
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
