"""btrapz_prism_corridor_batch_device (prisms -> corridors in one launch) against btrapz_prism_bounds_device +
btrapz_corridor_batch_device on random scenes, shapes and reference lines: the batch record, the segment counts and the
strip counts must be the same bits.

    python tests/fuzz/fused_vs_two_launches.py SEED ROUNDS     # needs a GPU; ~1 s per round of 2 000 scenes

Round 3: 200 rounds x 2 000 scenes (P 1-16, N 11-301, O 1-33, both variants), 0 differences.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import time, numpy as np, torch
from spectral_amd.solver import BatchSolver
from test_gpu_prism_bounds import pack, random_scenes, _knot_inputs
solver = BatchSolver(0)
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(seed0)
bad = 0; scenes_total = 0; t0 = time.time()
for r in range(rounds):
    Pm = int(rng.choice([1, 2, 3, 4, 6, 9, 16])); N = int(rng.choice([11, 31, 71, 101, 121, 201, 301])); variant = int(rng.integers(0, 2))
    Omax = int(rng.choice([1, 2, 3, 5, 2 * Pm + 1, min(33, 2 * Pm + 3)])); st = int(rng.choice([8, 16, 24, 40]))
    scenes = random_scenes(1000, int(rng.integers(1 << 30)), max_cars=Pm) + random_scenes(1000, int(rng.integers(1 << 30)), max_cars=min(Pm, 4), nice=True)
    B = len(scenes); scenes_total += B
    pr = torch.from_numpy(pack(scenes, Pm))
    s_ref, l_ref, init, dsb, dlb = _knot_inputs(B, N, int(rng.integers(1 << 30)))
    if r % 5 == 4:   # garbage among the references
        s_ref[::7, N // 2] = np.nan; l_ref[::11, 0] = np.inf
    t = torch.from_numpy
    sb, lb, n = solver.prism_bounds(pr, N, Omax)
    two = solver.corridor_batch_tensors(variant, N, 0.1, sb, lb, t(dsb), t(dlb), t(s_ref), t(l_ref), t(init), seg_stride=st)
    one = solver.prism_corridor_batch(variant, pr, N, Omax, 0.1, t(dsb), t(dlb), t(s_ref), t(l_ref), t(init), seg_stride=st)
    torch.cuda.synchronize()
    same = (torch.equal(one["n_strips"], n) and torch.equal(one["seg_count"], two["seg_count"]) and
            one["seg"].cpu().numpy().tobytes() == two["seg"].cpu().numpy().tobytes() and
            one["ref_end"].cpu().numpy().tobytes() == two["ref_end"].cpu().numpy().tobytes() and torch.equal(one["dl_bounds"], two["dl_bounds"]))
    if not same:
        bad += 1
        print("DIFFERENCE: round", r, dict(P=Pm, N=N, O=Omax, variant=variant, seg_stride=st), flush=True)
    if r % 20 == 19:
        print("rounds", r + 1, "scenes", scenes_total, "differences", bad, "usable", float((two["seg_count"] > 0).float().mean()), "seconds %.1f" % (time.time() - t0), flush=True)
print("rounds", rounds, "scenes", scenes_total, "differences", bad, "seconds %.1f" % (time.time() - t0))
