#!/usr/bin/env python3
"""Knot-level end-to-end timing: device corridor stage (btrapz_corridor_batch_device) + ragged QP
solve (btrapz_solve_ragged_device) on jittered copies of a bundled corridor file.

    python tools/pipeline_bench.py [--input c_road_s1_3] [--batch 65536] [--variant 0] [--reps 5]

The corridor stage is the one HBM-streaming kernel of the path: its roofline entry divides the input it must read
(per-knot bounds of every obstacle, reference, ds/dl bounds) plus the batch record it writes by the kernel's
HIP-event duration.  One JSON line on stdout."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    """Runs the tool; returns the dict it prints as ONE JSON line (bench.py calls it in-process)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", default="c_road_s1_3", help="bundled corridor file (tests/golden/inputs/<name>.txt)")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--seg-stride", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--scenario1", action="store_true",
                    help="BASELINE config 3 at knot level: synth.scenario1_knots (src/c1.txt's scene tiled over 20 s, N = 201, "
                         "18-24 segments per candidate) instead of jittered copies of a bundled file")
    ap.add_argument("--prisms", action="store_true",
                    help="start one stage earlier (SURVEY 8f rank 4): B scenes of two obstacle prisms (the harness's "
                         "constellation, cart_frenet.py:1536-1546, jittered) -> btrapz_prism_bounds_device -> corridors -> QP")
    ap.add_argument("--cap", type=int, default=0,
                    help="btrapz_options.cap_iter of the ragged solve (0: the library's choice = one launch; n > 0: two launches, "
                         "hand-over after n iterations; DESIGN 3.8)")
    a = ap.parse_args(argv)
    import torch
    from spectral_amd import knots, synth, layout as L
    from spectral_amd.solver import BatchSolver
    gold = os.path.join(ROOT, "tests", "golden", "inputs")
    W = np.loadtxt(os.path.join(gold, "weights.txt"))
    solver = BatchSolver(0)
    d = solver.device
    B, st = a.batch, a.seg_stride
    prism_ms = None
    if a.prisms:
        from spectral_amd.knots import KnotBatch
        rng = np.random.default_rng(11)
        N, O = 71, 5
        pr = np.zeros((B, 2, 8))
        pr[:, 0, :7] = np.stack([rng.uniform(18, 30, B), np.full(B, 1.2), np.zeros(B), rng.uniform(3, 5, B), np.zeros(B), np.full(B, 4.0), np.ones(B)], 1)
        pr[:, 1, :7] = np.stack([rng.uniform(5, 15, B), np.full(B, 4.2), np.zeros(B), rng.uniform(5, 7, B), np.zeros(B), np.full(B, 4.0), np.ones(B)], 1)
        d_pr = torch.from_numpy(pr).to(d)
        for _ in range(2):
            sb, lb, ns = solver.prism_bounds(d_pr, N, O)
        torch.cuda.synchronize()
        tp = []
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); sb, lb, ns = solver.prism_bounds(d_pr, N, O); e1.record()
            torch.cuda.synchronize(); tp.append(e0.elapsed_time(e1))
        prism_ms = float(np.median(tp))
        tt = np.arange(N) * 0.1
        kb = KnotBatch(B, N, O, 0.1, sb.cpu().numpy(), lb.cpu().numpy(), np.tile(np.array([0.0, 20.0]), (B, N, 1)),
                       np.tile(np.array([-3.0, 3.0]), (B, N, 1)), np.tile(40.0 / 7.0 * tt, (B, 1)),
                       np.tile(np.clip(1.2 + 0.0825 * (np.arange(N) - 15), 1.2, 4.5), (B, 1)),
                       np.tile(np.array([0.0, 6.0, 0.0, 1.2, 0.0, 0.0]), (B, 1)), dict(synth.C1_HEADER))
        st = max(st, 24)
        label = "%d scenes of two obstacle prisms -> strips (btrapz_prism_bounds_device)" % B
    elif a.scenario1:
        kb = synth.scenario1_knots(B, 20)
        st = max(st, 32)
        label = "%d scenario_1-shaped candidates at knot level (synth.scenario1_knots: c1.txt's scene tiled over 20 s)" % B
    else:
        kb = knots.jittered(knots.parse_corridor_file(os.path.join(gold, a.input + ".txt")), B, seed=3)
        label = "%d jittered copies of %s.txt" % (B, a.input)
    sh = synth.shared_params(a.variant, weights=W)
    sh.ds_ref, sh.dl_ref = kb.header["ds_ref"], kb.header["dl_ref"]
    sh.dds, sh.ddds, sh.ddl, sh.dddl = kb.header["dds"], kb.header["ddds"], kb.header["ddl"], kb.header["dddl"]
    f = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(d)
    ins = [f(kb.s_bounds), f(kb.l_bounds), f(kb.ds_bounds), f(kb.dl_bounds), f(kb.s_ref), f(kb.l_ref)]
    rec = dict(B=B, seg_stride=st, seg=torch.zeros((L.NUM_SEG_FIELDS, B, st), dtype=torch.float64, device=d),
               seg_count=torch.zeros(B, dtype=torch.int32, device=d), init=f(kb.init),
               ref_end=torch.zeros((B, 2), dtype=torch.float64, device=d),
               dl_bounds=torch.zeros((B, 10), dtype=torch.float64, device=d))
    stream = torch.cuda.current_stream(d).cuda_stream

    def corridors():
        solver.ctx.corridor_batch_device(a.variant, B, kb.N, kb.num_obs, kb.delta, *ins, st, rec["seg"], rec["seg_count"],
                                         rec["ref_end"], rec["dl_bounds"], stream=stream)

    for _ in range(2):
        corridors(); out = solver.solve_ragged(rec, sh, cap_iter=a.cap)
    torch.cuda.synchronize()
    tc, ts = [], []
    for _ in range(a.reps):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); corridors(); e1.record(); out = solver.solve_ragged(rec, sh, cap_iter=a.cap); e2.record()
        torch.cuda.synchronize()
        tc.append(e0.elapsed_time(e1)); ts.append(e1.elapsed_time(e2))
    fused_ms = None
    if a.prisms:   # the same stage in one launch: strips evaluated inside the corridor kernel (btrapz_prism_corridor_batch_device)
        from spectral_amd.native import CRoad
        rec2 = {k: (torch.zeros_like(v) if torch.is_tensor(v) else v) for k, v in rec.items()}
        ns2 = torch.empty(B, dtype=torch.int32, device=d)

        def fused():
            solver.ctx.prism_corridor_batch_device(a.variant, B, 2, kb.N, CRoad.reference(), d_pr, kb.num_obs, kb.delta, *ins[2:], st,
                                                   rec2["seg"], rec2["seg_count"], rec2["ref_end"], rec2["dl_bounds"], ns2, stream=stream)
        for _ in range(2):
            fused()
        torch.cuda.synchronize()
        tf = []
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fused(); e1.record(); torch.cuda.synchronize(); tf.append(e0.elapsed_time(e1))
        fused_ms = float(np.median(tf))
        same = bool(torch.equal(rec2["seg_count"], rec["seg_count"]) and rec2["seg"].cpu().numpy().tobytes() == rec["seg"].cpu().numpy().tobytes())
    cnt = rec["seg_count"].cpu().numpy(); status = out["status"].cpu().numpy()
    mean_cnt = float(cnt[cnt > 0].mean()) if (cnt > 0).any() else 0.0
    in_bytes = sum(t.numel() * 8 for t in ins)
    out_bytes = int(B * (L.NUM_SEG_FIELDS * mean_cnt * 8 + 4 + 16 + 80))
    c_ms, s_ms = float(np.median(tc)), float(np.median(ts))
    extra = {}
    if prism_ms is not None:
        pb = B * 5 * 71 * 32 + B * 2 * 64
        extra["prism_roofline"] = {"bound": "hbm", "kernel": "btrapz::prism_bounds_kernel", "kernel_ms": prism_ms,
                                   "algorithmic_bytes_per_scene": pb / B, "achieved": pb / (prism_ms * 1e-3) / 1e9, "peak": 8000.0,
                                   "unit": "GB/s", "frac": pb / (prism_ms * 1e-3) / 1e9 / 8000.0}
        extra["prism_ms"] = prism_ms
        extra["end_to_end_scenes_per_s"] = B / (prism_ms + float(np.median(tc)) + float(np.median(ts))) * 1e3
        extra["fused_prism_corridor_ms"] = fused_ms
        extra["fused_equals_two_launches"] = same
        extra["end_to_end_scenes_per_s_fused"] = B / (fused_ms + float(np.median(ts))) * 1e3
    result = {**extra,
        "workload": "%s (N = %d knots, %d obstacles), %s constraints" %
                    (label, kb.N, kb.num_obs, "trapezoid" if a.variant == 0 else "cuboid"),
        "corridor_ms": c_ms, "ragged_solve_ms": s_ms, "ragged_solve_cap_iter": a.cap, "ragged_solve_form": solver.ctx.last_solve_form(), "end_to_end_candidates_per_s": B / (c_ms + s_ms) * 1e3,
        "segments_per_candidate": {int(k): int(v) for k, v in zip(*np.unique(cnt, return_counts=True))},
        "solved_fraction": float(np.mean((status == 1) | (status == 2))),
        "corridor_roofline": {"bound": "hbm", "kernel": "btrapz::corridor_batch_kernel", "kernel_ms": c_ms,
                              "algorithmic_bytes_per_candidate": (in_bytes + out_bytes) / B,
                              "achieved": (in_bytes + out_bytes) / (c_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                              "frac": (in_bytes + out_bytes) / (c_ms * 1e-3) / 1e9 / 8000.0},
    }
    print(json.dumps(result))
    return result


if __name__ == "__main__":
    main()
