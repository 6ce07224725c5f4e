"""Upper bound of 'sort candidates by predicted difficulty': the previous solve's own iteration counts as the hint."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spectral_amd import synth
from spectral_amd.solver import BatchSolver
solver = BatchSolver(0); dev = solver.device
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for B in (4096, 8192, 16384, 65536):
    batch, sh = synth.make_scenario1_batch(B, 20, 0)
    db = solver.upload(batch)
    o = solver.solve(db, sh, lean=1, cap_iter=-1, split=-1)
    torch.cuda.synchronize()
    iters = o["iters"].clone()
    os.makedirs("gpurun_out", exist_ok=True); np.save("gpurun_out/iters_B%d.npy" % B, np.stack([iters.cpu().numpy(), o["status"].cpu().numpy()]))
    t_plain = timed(lambda: solver.solve(db, sh, lean=1, cap_iter=-1, split=-1))
    t_auto = timed(lambda: solver.solve(db, sh, split=-1))
    hint = (iters + 1).to(torch.int32).contiguous()
    t_hint = timed(lambda: solver.solve(db, sh, lean=1, split=-1, warm=dict(hint=hint)))
    # coarse classes: <= 8, 9-10, 11-14, 15+
    h2 = torch.where(hint <= 8, 1, torch.where(hint <= 10, 2, torch.where(hint <= 14, 3, 4))).to(torch.int32).contiguous()
    t_h2 = timed(lambda: solver.solve(db, sh, lean=1, split=-1, warm=dict(hint=h2)))
    print("B %6d  lean one launch %.3f ms  automatic %.3f  perfect hint (iteration classes) %.3f  four classes %.3f  form %d" % (B, t_plain, t_auto, t_hint, t_h2, solver.ctx.last_solve_form()), flush=True)
