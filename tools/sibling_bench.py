#!/usr/bin/env python3
"""Sibling warm start: candidate sets that share an ego state are alike -- solve K representatives
cold, start every other candidate from the nearest representative's joint states and multipliers
(btrapz_solve_warm_device: its in-kernel cold restart bounds the downside), and see what that does to iterations and to
the time of the whole batch.  The reference solves every call cold (src/solve_3d.cc:1246,1256).

"Nearest" needs no knowledge of the generator: a candidate's feature vector is its initial state and the s / l corridor
lines of its segments (bias and skew), scaled per feature; the representatives are every `stride`-th candidate.

    python tools/sibling_bench.py [--batch 65536] [--stride 256]

One JSON object on stdout: per bench batch, the cold solve (the library's automatic form), the warm solve from
siblings (packed warm-start kernel: the lean form has no warm-start instantiation), what the nearest-sibling search
and the gather cost, iterations, accept sets and the largest deviation of the control points from the cold solve's.
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
    from spectral_amd import layout as L
    from spectral_amd import synth
    from spectral_amd.solver import BatchSolver
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--stride", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args(argv)
    solver = BatchSolver(0)
    dev = torch.device("cuda:0")
    out = {"representatives": "every %d-th candidate" % a.stride}
    cases = [("scenario1 x 20 trapezoid", lambda: synth.make_scenario1_batch(a.batch, 20, 0)),
             ("generic x 20 trapezoid", lambda: synth.make_batch(a.batch, 20, config=3)),
             ("scenario1 x 20 cuboid", lambda: synth.make_scenario1_batch(a.batch, 20, 1)),
             ("generic x 10 trapezoid", lambda: synth.make_batch(a.batch, 10, config=2))]

    def timed(fn):
        for _ in range(2):
            r = fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            r = fn()
        e1.record(); torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / a.reps, r

    for label, make in cases:
        batch, sh = make()
        B, S = batch.B, batch.S
        db = solver.upload(batch)
        cold_ms, cold = timed(lambda: solver.solve(db, sh))
        cold_form = solver.ctx.last_solve_form()
        cold = {k: v.clone() for k, v in cold.items()}
        # representatives: every stride-th candidate, solved cold with their multipliers kept
        rep = np.arange(0, B, a.stride)
        rb = batch.slice(0, B)            # (a copy to index)
        sub = type(batch)(B=len(rep), S=S, seg=rb.seg[:, rep], init=rb.init[rep], ref_end=rb.ref_end[rep], dl_bounds=rb.dl_bounds[rep])
        dsub = solver.upload(sub)
        rep_ms, r = timed(lambda: solver.solve(dsub, sh, keep_multipliers=True))
        ends = torch.cumsum(dsub.seg[L.F_T], dim=1)                                     # [K][S] end time of every segment
        xrep = solver.eval_states(dsub, r["ctrl"], ends)                                 # [K][2][S][3]
        lam_rep = r["lam"]                                                               # [2][36][K][S]
        # nearest representative in feature space
        feats = [db.init] + [db.seg[f] for f in (L.F_UPP_BIAS, L.F_UPP_SKEW, L.F_DOWN_BIAS, L.F_DOWN_SKEW, L.F_L_UPP_BIAS, L.F_L_DOWN_BIAS, L.F_T)]
        F = torch.cat(feats, dim=1)
        F = (F - F.mean(0)) / (F.std(0) + 1e-9)

        def assign():
            d = torch.cdist(F, F[torch.from_numpy(rep).to(dev)])
            near = d.argmin(dim=1)
            x0 = xrep.index_select(0, near).contiguous()
            lam0 = lam_rep.index_select(2, near).contiguous()
            return near, x0, lam0
        assign_ms, (near, x0, lam0) = timed(assign)
        rec = {"cold": {"solve_ms": cold_ms, "form": cold_form, "mean_iterations": float(cold["iters"].double().mean().item() + 1),
                        "solved_fraction": float((cold["status"] > 0).double().mean().item())},
               "representatives": int(len(rep)), "representatives_solve_ms": rep_ms, "nearest_and_gather_ms": assign_ms}
        for name, warm in (("warm_x_and_multipliers", dict(x0=x0, lam=lam0)), ("warm_x_only", dict(x0=x0))):
            ms, w = timed(lambda: solver.solve(db, sh, warm=dict(warm)))
            both = (cold["status"] > 0) & (w["status"] > 0)
            scale = cold["ctrl"][both].abs().amax(dim=1, keepdim=True)
            rec[name] = {"solve_ms": ms, "mean_iterations": float(w["iters"].double().mean().item() + 1),
                         "accept_differs": int(((cold["status"] > 0) != (w["status"] > 0)).sum().item()),
                         "max_rel_ctrl_vs_cold": float(((w["ctrl"][both] - cold["ctrl"][both]).abs() / scale).max().item()) if both.any() else None,
                         "total_ms_with_representatives_and_gather": ms + rep_ms + assign_ms,
                         "ratio_to_cold": (ms + rep_ms + assign_ms) / cold_ms}
        out[label] = rec
        print(label, json.dumps(rec), file=sys.stderr, flush=True)
        del x0, lam0, lam_rep, xrep
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
