"""The lean form of the solve kernel (two wavefronts per SIMD) against the packed form on ragged batches made from every
bundled corridor file (jittered at knot level, both variants) and on uniform synthetic batches of every width: who accepts
what, how far apart the accepted control points are, how the iteration counts compare.

    python tests/fuzz/lean_vs_packed.py [B=16384] [SEEDS=2]

Round 4: see profiles/r04_fuzz_campaign.txt.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from spectral_amd import knots, synth
from spectral_amd.solver import BatchSolver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
solver = BatchSolver(0)
gold = os.path.join(ROOT, "tests", "golden", "inputs")
W = np.loadtxt(os.path.join(gold, "weights.txt"))
tot = dict(cands=0, accept_differs=0, status_differs=0, worst=0.0, lean_more_iters=0, packed_more_iters=0)


def compare(tag, p, l, l2):
    acc_p, acc_l = p["status"] > 0, l["status"] > 0
    both = acc_p & acc_l
    worst = 0.0
    if both.any():
        scale = np.abs(p["ctrl"][both]).max(axis=1, keepdims=True)
        worst = float((np.abs(l["ctrl"][both] - p["ctrl"][both]) / np.maximum(scale, 1e-300)).max())
    two = bool(np.array_equal(l["status"], l2["status"]) and np.array_equal(l["iters"], l2["iters"]) and np.array_equal(l["cost"], l2["cost"]) and
               np.array_equal(l["ctrl"][acc_l], l2["ctrl"][acc_l]))
    di = l["iters"][both].astype(int) - p["iters"][both]
    tot["cands"] += len(acc_p); tot["accept_differs"] += int((acc_p != acc_l).sum()); tot["status_differs"] += int((p["status"] != l["status"]).sum())
    tot["worst"] = max(tot["worst"], worst); tot["lean_more_iters"] += int((di > 0).sum()); tot["packed_more_iters"] += int((di < 0).sum())
    print(tag, "n", len(acc_p), "accepted", int(acc_p.sum()), int(acc_l.sum()), "accept differs", int((acc_p != acc_l).sum()), "status differs",
          int((p["status"] != l["status"]).sum()), "worst rel %.2e" % worst, "mean iters %.4f %.4f" % (p["iters"][both].mean() + 1 if both.any() else 0, l["iters"][both].mean() + 1 if both.any() else 0),
          "two launches bit-identical", two, flush=True)
    if (acc_p != acc_l).any():
        idx = np.nonzero(acc_p != acc_l)[0][:5]
        print("   differing candidates", idx.tolist(), "packed status", p["status"][idx].tolist(), "lean status", l["status"][idx].tolist(), "iters", p["iters"][idx].tolist(), l["iters"][idx].tolist())


def grab(o):
    torch.cuda.synchronize()
    return {k: v.cpu().numpy().copy() for k, v in o.items()}


for seed in range(seeds):
    for name in sorted(f[:-4] for f in os.listdir(gold) if f.startswith("c") and f.endswith(".txt")):
        for variant in (0, 1):
            kb = knots.jittered(knots.parse_corridor_file(os.path.join(gold, name + ".txt")), B, seed=40 + seed)
            sh = synth.shared_params(variant, weights=W)
            h = kb.header
            sh.ds_ref, sh.dl_ref, sh.dds, sh.ddds, sh.ddl, sh.dddl = h["ds_ref"], h["dl_ref"], h["dds"], h["ddds"], h["ddl"], h["dddl"]
            rec = solver.corridor_batch(kb, variant, seg_stride=32)
            p = grab(solver.solve_ragged(rec, sh, lean=-1, cap_iter=-1))
            l = grab(solver.solve_ragged(rec, sh, lean=1, cap_iter=-1))
            l2 = grab(solver.solve_ragged(rec, sh, lean=1, cap_iter=7))
            compare("%s v%d seed %d" % (name, variant, seed), p, l, l2)
    for S in (3, 5, 7, 10, 14, 20, 21, 32, 40, 64):
        for variant in (0, 1):
            batch, sh = synth.make_batch(min(B, 8192), S, config=2, variant=variant, seed=synth.SEED_BASE + 77 + seed * 100 + S)
            db = solver.upload(batch)
            p = grab(solver.solve(db, sh, lean=-1, cap_iter=-1, split=-1))
            l = grab(solver.solve(db, sh, lean=1, cap_iter=-1, split=-1))
            l2 = grab(solver.solve(db, sh, lean=1, cap_iter=6, split=-1))
            compare("uniform S %d v%d seed %d" % (S, variant, seed), p, l, l2)
print("TOTAL", tot)
