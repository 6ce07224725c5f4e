import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from spectral_amd import knots, synth
from spectral_amd.solver import BatchSolver
gold = os.path.join("tests", "golden")
W = np.loadtxt(os.path.join(gold, "inputs", "weights.txt"))
solver = BatchSolver(0)
B = 16384
sets = [("c_road_s1_3", 0), ("c1", 1), ("c3", 0), ("c1", 0), ("c2", 1), ("c6", 0), ("c7", 1)]
recs = []
for name, variant in sets:
    kb = knots.jittered(knots.parse_corridor_file(os.path.join(gold, "inputs", name + ".txt")), B, seed=3)
    sh = synth.shared_params(variant, weights=W)
    sh.ds_ref, sh.dl_ref = kb.header["ds_ref"], kb.header["dl_ref"]
    sh.dds, sh.ddds, sh.ddl, sh.dddl = kb.header["dds"], kb.header["ddds"], kb.header["ddl"], kb.header["dddl"]
    recs.append((name, variant, sh, solver.corridor_batch(kb, variant, seg_stride=24)))
for frac, thr in (("0", "0"), ("0.995", "1.0"), ("0.9999", "0.5"), ("0.999", "0.9"), ("0.9999", "0.95")):
    os.environ["BTRAPZ_STEP_FRACTION"] = frac; os.environ["BTRAPZ_STEP_THRESHOLD"] = thr
    out = []
    for name, variant, sh, rec in recs:
        o = solver.solve_ragged(rec, sh, lean=1)
        torch.cuda.synchronize()
        st = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy()
        out.append("%s/%d: %d %.3f" % (name, variant, int((st > 0).sum()), it[st > 0].mean() + 1))
    print("frac", frac, "thr", thr, " | ".join(out), flush=True)
