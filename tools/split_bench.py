#!/usr/bin/env python3
"""Split form of the solve kernel against the packed one (btrapz_options.split): latency of few
candidates, throughput of many.  One JSON object on stdout.

    python tools/split_bench.py [--reps 1000] [--batch 65536]
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
    ap.add_argument("--reps", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=65536)
    a = ap.parse_args(argv)
    solver = BatchSolver(0)
    dev = torch.device("cuda:0")
    out = {"latency_ms": {}, "throughput": {}, "find_traj_mem_ms": {}}

    def p50(fn, reps):
        lat = []
        for i in range(reps + 20):
            torch.cuda.synchronize(dev)
            t = time.perf_counter(); fn(); torch.cuda.synchronize(dev); lat.append(time.perf_counter() - t)
        lat = np.array(lat[20:]) * 1e3
        return float(np.percentile(lat, 50)), float(np.percentile(lat, 99))

    # few candidates, inputs resident: wall time of solve + synchronisation
    for gen, S in (("scenario1", 20), ("scenario1", 10), ("scenario1", 7), ("generic", 20)):
        for B in (1, 64, 512):
            batch, sh = (synth.make_scenario1_batch(max(B, 4), S, 0) if gen == "scenario1" else synth.make_batch(max(B, 4), S, config=3))
            db = solver.upload(batch.slice(0, B))
            rec = {}
            for name, split in (("packed", -1), ("split", 1)):
                o = solver.solve(db, sh, split=split)
                torch.cuda.synchronize(dev)
                rec[name + "_iters_mean"] = float(o["iters"].float().mean().item() + 1)
                rec[name + "_p50"], rec[name + "_p99"] = p50(lambda: solver.solve(db, sh, split=split), a.reps if B == 1 else max(100, a.reps // 5))
            rec["ratio"] = rec["split_p50"] / rec["packed_p50"]
            out["latency_ms"]["%s S=%d B=%d" % (gen, S, B)] = rec

    # many candidates: kernel time by HIP events on the launch stream
    for gen in ("scenario1", "generic"):
        batch, sh = (synth.make_scenario1_batch(a.batch, 20, 0) if gen == "scenario1" else synth.make_batch(a.batch, 20, config=3))
        db = solver.upload(batch)
        rec = {}
        for name, split in (("packed", -1), ("split", 1)):
            for _ in range(2):
                o = solver.solve(db, sh, split=split)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                o = solver.solve(db, sh, split=split)
            e1.record(); torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / 5
            st = o["status"].cpu().numpy()
            rec[name] = {"ms": ms, "solves_per_s": a.batch / (ms * 1e-3), "solved": int(((st == 1) | (st == 2)).sum()),
                         "mean_iterations": float(o["iters"].float().mean().item() + 1)}
        rec["split_over_packed_time"] = rec["split"]["ms"] / rec["packed"]["ms"]
        out["throughput"]["%s 20 segments x %d" % (gen, a.batch)] = rec

    # the reference's call pattern: one find_traj per replanning step, arrays in / arrays out
    prm = native.CParams(*[float(v) for v in synth.REFERENCE_WEIGHTS], 0)
    from spectral_amd import knots
    gold = os.path.join(ROOT, "tests", "golden", "inputs")
    scenes = [("scenario1 knots S=20", synth.scenario1_knots(1, 20))] + \
             [(n, knots.parse_corridor_file(os.path.join(gold, n + ".txt"))) for n in ("c1", "c2", "c_road_s1_3")]
    for label, kb in scenes:
        rec = {}
        for name, env in (("packed", "0"), ("split", None)):
            if env is None:
                os.environ.pop("BTRAPZ_SPLIT", None)
            else:
                os.environ["BTRAPZ_SPLIT"] = env
            lat = []
            for i in range(a.reps // 2 + 10):
                t = time.perf_counter(); c, _, ctl = native.find_traj_mem(0, prm, kb); lat.append(time.perf_counter() - t)
            rec[name + "_p50"] = float(np.percentile(np.array(lat[10:]) * 1e3, 50))
            rec[name + "_iters"] = int(native.lib().btrapz_find_traj_last_iterations())
        os.environ.pop("BTRAPZ_SPLIT", None)
        rec["segments"] = None if ctl is None else len(ctl) // 12
        out["find_traj_mem_ms"][label] = rec
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
