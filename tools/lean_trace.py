import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from spectral_amd import synth
from spectral_amd.solver import BatchSolver
solver = BatchSolver(0); B = 65536
batch, sh = synth.make_scenario1_batch(B, 20, 0); db = solver.upload(batch)
o = solver.solve(db, sh, split=-1, lean=1, cap_iter=-1)
st = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy() + 1
tr = o["ctrl"].cpu().numpy().reshape(B, 2, 30, 4)
np.set_printoptions(linewidth=200, precision=3)
order = np.argsort(-it)
print("iters histogram", np.bincount(it))
shown = 0
for b in list(order[:6]) + list(np.where(it == 9)[0][:2]) + list(np.where(it == 13)[0][:2]) + list(np.where(it == 17)[0][:3]):
    print("candidate", b, "status", st[b], "iters", it[b])
    for ax in range(2):
        print("  axis", ax)
        for i in range(30):
            sc, mu, al, sr = tr[b, ax, i]
            if sc == 0 and mu == 0: break
            print(f"    it {i:2d} score {sc:9.2e} mu {mu:9.2e} alpha {abs(al):6.4f} {'dual' if al < 0 else 'prim'} sr {abs(sr):8.2e} {'(rp>rd)' if sr < 0 else ''}")
