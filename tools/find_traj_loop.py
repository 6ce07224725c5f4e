#!/usr/bin/env python3
"""find_traj in a loop on one scene (for rocprofv3 --kernel-trace: duration of the single-candidate launch).

    python tools/find_traj_loop.py [--segments 20] [--calls 300] [--file tests/golden/inputs/c1.txt] [--warm]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    from spectral_amd import knots, native, synth
    ap = argparse.ArgumentParser()
    ap.add_argument("--segments", type=int, default=20)
    ap.add_argument("--calls", type=int, default=300)
    ap.add_argument("--file", default=None)
    ap.add_argument("--warm", action="store_true")
    a = ap.parse_args(argv)
    if a.warm:
        os.environ["BTRAPZ_WARM"] = "1"
    prm = native.CParams(*[float(v) for v in synth.REFERENCE_WEIGHTS], 0)
    kb = knots.parse_corridor_file(a.file) if a.file else synth.scenario1_knots(1, a.segments)
    lat = []
    call = native.TrajCall(0, prm, kb)      # struct and buffers built once: a call is the library's time
    for i in range(a.calls):
        t = time.perf_counter(); cost, traj, ctrl = call(); lat.append(time.perf_counter() - t)
    import json
    print(json.dumps({"p50_ms": float(np.percentile(np.array(lat[10:]) * 1e3, 50)), "calls": a.calls,
                      "iterations": int(native.lib().btrapz_find_traj_last_iterations()),
                      "segments": None if ctrl is None else len(ctrl) // 12, "warm": bool(a.warm),
                      "split_form": os.environ.get("BTRAPZ_SPLIT", "1") != "0"}))


if __name__ == "__main__":
    main()
