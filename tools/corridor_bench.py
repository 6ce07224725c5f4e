#!/usr/bin/env python3
"""The corridor stage alone (btrapz_corridor_batch_device), timed and fingerprinted: 65 536 jittered copies of a bundled
corridor file (default c_road_s1_3.txt: N = 71 knots, 3 obstacles -- the workload of profiles/r0N_corridor_pmc.json) and the
scenario_1 scene at knot level (N = 201, 2 obstacles).  One JSON line: ms (median of --reps launches, HIP events), the HBM
roofline fraction of the bytes the stage must move, and a hash of everything it wrote (two builds that differ in any bit
differ here: A/B runs of kernel variants, BTRAPZ_HIP_LIB=scratch/variants/X/libbtrapz_hip.so).

    python tools/corridor_bench.py [--input c_road_s1_3] [--batch 65536] [--reps 20] [--scenario1]
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", default="c_road_s1_3")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--seg-stride", type=int, default=16)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--scenario1", action="store_true")
    a = ap.parse_args(argv)
    import torch
    from spectral_amd import knots, synth, layout as L
    from spectral_amd.solver import BatchSolver
    solver = BatchSolver(0); d = solver.device
    B, st = a.batch, a.seg_stride
    if a.scenario1:
        kb = synth.scenario1_knots(B, 20); st = max(st, 32)
    else:
        kb = knots.jittered(knots.parse_corridor_file(os.path.join(ROOT, "tests", "golden", "inputs", a.input + ".txt")), B, seed=3)
    f = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(d)
    ins = [f(kb.s_bounds), f(kb.l_bounds), f(kb.ds_bounds), f(kb.dl_bounds), f(kb.s_ref), f(kb.l_ref)]
    seg = torch.zeros((L.NUM_SEG_FIELDS, B, st), dtype=torch.float64, device=d)
    cnt = torch.zeros(B, dtype=torch.int32, device=d); ref_end = torch.zeros((B, 2), dtype=torch.float64, device=d)
    dl10 = torch.zeros((B, 10), dtype=torch.float64, device=d)
    stream = torch.cuda.current_stream(d).cuda_stream

    def run():
        solver.ctx.corridor_batch_device(a.variant, B, kb.N, kb.num_obs, kb.delta, *ins, st, seg, cnt, ref_end, dl10, stream=stream)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1))
    ms = float(np.median(t))
    c = cnt.cpu().numpy()
    h = hashlib.sha256(seg.cpu().numpy().tobytes() + c.tobytes() + ref_end.cpu().numpy().tobytes() + dl10.cpu().numpy().tobytes()).hexdigest()[:16]
    in_bytes = (4 * kb.num_obs + 6) * kb.N * 8
    out_bytes = float(np.mean(np.maximum(c, 0))) * L.NUM_SEG_FIELDS * 8 + 4 + 16 + 80
    gbs = (in_bytes + out_bytes) * B / (ms * 1e-3) / 1e9
    out = {"workload": ("scenario_1 at knot level" if a.scenario1 else "jittered " + a.input + ".txt") + ", %d candidates, N = %d, %d obstacles, variant %d" % (B, kb.N, kb.num_obs, a.variant),
           "corridor_ms": ms, "min_ms": float(np.min(t)), "hash": h, "mean_segments": float(np.mean(np.maximum(c, 0))), "refused": int((c < 0).sum()),
           "bytes_per_candidate": in_bytes + out_bytes, "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0}}
    # a -DCABL_TIMING build: where a wavefront's lifetime goes, phase by phase (wall-clock cycles between the phase marks,
    # summed over the wavefronts of one more launch; slot 0 is the set-up in front of the first mark)
    import ctypes as C
    from spectral_amd import native
    try:
        fn = native.lib().btrapz_debug_corridor_timing
    except AttributeError:
        fn = None
    if fn is not None:
        buf = (C.c_ulonglong * 16)()
        fn(None, 1); run(); torch.cuda.synchronize(); fn(buf, 0)
        # slot i = the stretch that ENDS at mark i: set-up, then the phase each mark closes
        names = ["setup", "slopes", "refs issued", "segments: pieces (CorridorSplit)", "refs stored", "selection", "de-dup", "rank + place", "reorder", "overlap", "record", "break search",
                 "segments: lanes mapped to base segments", "segments: references stored", "segments: base segments built (bounds at their starts)"]
        tot = float(sum(buf[:len(names)])) or 1.0
        out["phase_share_of_wavefront_lifetime"] = {n: round(buf[i] / tot, 4) for i, n in enumerate(names)}
        out["cycles_per_wavefront"] = tot / B
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
