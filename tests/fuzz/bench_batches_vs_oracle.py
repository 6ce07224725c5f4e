"""Kernel vs the oracle's exact solve on the bench batches: who accepts what, and how far the accepted control points
are from x*.

    python tests/fuzz/bench_batches_vs_oracle.py [N=4096] [THREADS=16] [OFFSET=0]   # N = 65536: all of them, ~4 min

Round 2, all 4 x 65 536 candidates: 0 kernel-only, 0 oracle-only, worst relative deviation 1.5e-6.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sys, os, time, numpy as np
import torch
from oracle import oracle as O
from spectral_amd import synth
from spectral_amd.solver import BatchSolver
solver=BatchSolver(0)
n=int(sys.argv[1]) if len(sys.argv)>1 else 4096
thr=int(sys.argv[2]) if len(sys.argv)>2 else 16
off=int(sys.argv[3]) if len(sys.argv)>3 else 0
for tag,mk in (('scenario1 S20 trapezoid',lambda: synth.make_scenario1_batch(65536,20,0)),('scenario1 S20 cuboid',lambda: synth.make_scenario1_batch(65536,20,1)),
               ('generic S20',lambda: synth.make_batch(65536,20,config=3)),('scenario1 S10',lambda: synth.make_scenario1_batch(65536,10,0))):
    batch,sh=mk()
    o=solver.solve(solver.upload(batch),sh); torch.cuda.synchronize()
    st=o['status'].cpu().numpy()[off:off+n]; ctrl=o['ctrl'].cpu().numpy()[off:off+n]
    t0=time.time(); x,obj,ost,oit=O.batch_solve(batch,sh,off,off+n,exact=True,threads=thr); dt=time.time()-t0
    ka=st>0; oa=ost>0
    both=ka&oa
    err=(np.abs(ctrl[both]-x[both]).max(axis=1)/np.abs(x[both]).max(axis=1)).max() if both.any() else 0
    lost=np.nonzero(~ka&oa)[0]+off
    np.save(os.path.join(ROOT, 'gpurun_out', 'lost_%s.npy' % tag.replace(' ', '_')),lost)
    print(tag,'n',n,'kernel accepts',ka.sum(),'oracle accepts',oa.sum(),'kernel-only',(ka&~oa).sum(),'oracle-only',(~ka&oa).sum(),'worst rel err %.2e'%err,'oracle status of kernel -2:',dict(zip(*np.unique(ost[st==-2],return_counts=True))),'kernel status where oracle-only',dict(zip(*np.unique(st[~ka&oa],return_counts=True))),'oracle s %.0f'%dt, flush=True)
