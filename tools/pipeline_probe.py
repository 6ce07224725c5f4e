#!/usr/bin/env python3
"""Iteration counts by status on the knot-level pipeline batch (65 536 jittered copies of c_road_s1_3.txt), and the ragged
solve's time for a few hand-over settings: what holds the ragged solve of tools/pipeline_bench.py where it is."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    import torch
    from spectral_amd import knots, synth, layout as L
    from spectral_amd.solver import BatchSolver
    gold = os.path.join(ROOT, "tests", "golden", "inputs")
    W = np.loadtxt(os.path.join(gold, "weights.txt"))
    solver = BatchSolver(0); d = solver.device
    B = int(os.environ.get("PP_BATCH", "65536")); st = 16
    kb = knots.jittered(knots.parse_corridor_file(os.path.join(gold, os.environ.get("PP_INPUT", "c_road_s1_3") + ".txt")), B, seed=3)
    sh = synth.shared_params(int(os.environ.get('PP_VARIANT', '0')), weights=W)
    sh.ds_ref, sh.dl_ref = kb.header["ds_ref"], kb.header["dl_ref"]
    sh.dds, sh.ddds, sh.ddl, sh.dddl = kb.header["dds"], kb.header["ddds"], kb.header["ddl"], kb.header["dddl"]
    rec = solver.corridor_batch(kb, int(os.environ.get('PP_VARIANT', '0')), seg_stride=st)
    out = {}
    base = None
    variant = int(os.environ.get("PP_VARIANT", "0"))
    cases = [("one_nocompact", dict(cap_iter=-1, compact=-1)), ("one_compact", dict(cap_iter=-1, compact=1)), ("default", dict())]
    for cap in (4, 5, 6, 7, 8):
        cases += [("cap%d_nocompact" % cap, dict(cap_iter=cap, compact=-1))]
    cases += [("cap6_compact", dict(cap_iter=6, compact=1)), ("packed", dict(cap_iter=-1, lean=-1, compact=-1))]
    for label, kw in cases:
        for _ in range(2): o = solver.solve_ragged(rec, sh, **kw)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); o = solver.solve_ragged(rec, sh, **kw); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        stt = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy()
        if base is None: base = stt
        out[label] = dict(ms=round(best, 4), form=solver.ctx.last_solve_form(), status_same=bool(np.array_equal(stt > 0, base > 0)))
        if label == "one_nocompact":
            itx, stx = solver.ctx.debug_axis_records(B)
            hist = {}
            for code in np.unique(stx):
                v = itx[stx == code]
                hist[int(code)] = dict(n=int(v.size), mean=float(v.mean()), p50=float(np.median(v)), p90=float(np.percentile(v, 90)), max=int(v.max()))
            out["axis_iterations_by_status"] = hist
            out["candidate_status_counts"] = {int(k): int(v) for k, v in zip(*np.unique(stt, return_counts=True))}
    print(json.dumps(out))

if __name__ == "__main__":
    main()
