"""Warm-started solves against the oracle: four batch families, replanning shifts of 0.1 / 0.3 / 1.0 s, warm start from
the previous solution (states + multipliers), with and without scheduling hints, and cold beside them.

    python tests/fuzz/warm_start_vs_oracle.py [B=4096] [LEAN=0]     # LEAN: btrapz_options.lean (1: the two-wavefronts-per-SIMD form)

Round 2: 147 456 solves, 0 lost, one borderline candidate accepted that the oracle scores just above its limit.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sys, numpy as np, time
import torch
from oracle import oracle as O
from spectral_amd import synth, layout as L
from spectral_amd.solver import BatchSolver
from test_gpu_warm_start import joint_times
solver=BatchSolver(0)
B=int(sys.argv[1]) if len(sys.argv)>1 else 4096
LEAN=int(sys.argv[2]) if len(sys.argv)>2 else 0
tot_lost=tot_extra=0
for tag,mk in (('scenario1 v0',lambda: synth.make_scenario1_batch(B,20,0)),('scenario1 v1',lambda: synth.make_scenario1_batch(B,20,1)),('generic',lambda: synth.make_batch(B,20,config=3)),('scenario1 S10',lambda: synth.make_scenario1_batch(B,10,0))):
    batch,sh=mk(); S=batch.S
    db=solver.upload(batch)
    prev=solver.solve(db,sh,keep_multipliers=True,lean=LEAN)
    p_ctrl=prev['ctrl'].clone(); lam=prev['lam'].clone(); p_it=prev['iters'].clone()
    for d in (0.1,0.3,1.0):
        x0=solver.eval_states(db,p_ctrl,joint_times(batch,d))
        new_init=solver.eval_states(db,p_ctrl,torch.full((B,1),d,dtype=torch.float64)).cpu().numpy()
        nb=batch.slice(0,B); seg=nb.seg.copy()
        for bias,skew in ((L.F_DOWN_BIAS,L.F_DOWN_SKEW),(L.F_UPP_BIAS,L.F_UPP_SKEW),(L.F_L_DOWN_BIAS,L.F_L_DOWN_SKEW),(L.F_L_UPP_BIAS,L.F_L_UPP_SKEW),(L.F_X_BIAS,L.F_X_SKEW),(L.F_Y_BIAS,L.F_Y_SKEW)):
            seg[bias]=seg[bias]+seg[skew]*d
        nb.seg=seg; nb.init=np.concatenate([new_init[:,0,0],new_init[:,1,0]],axis=1)
        nb.init=np.where(np.isfinite(nb.init),nb.init,batch.init)     # previous candidate unsolved: keep its old state
        ndb=solver.upload(nb)
        xs,obj,ost,_=O.batch_solve(nb,sh,0,B,exact=True,threads=16)
        for mode in ('warm','warm+hint','cold'):
            kw={} if mode=='cold' else dict(warm=dict(x0=x0,lam=lam,**({'hint':torch.clamp((p_it-4)//4+1,min=1).to(torch.int32)} if mode=='warm+hint' else {})))
            o=solver.solve(ndb,sh,lean=LEAN,**kw); torch.cuda.synchronize()
            st=o['status'].cpu().numpy(); c=o['ctrl'].cpu().numpy(); it=o['iters'].cpu().numpy()
            ka,oa=st>0,ost>0
            both=ka&oa
            err=(np.abs(c[both]-xs[both]).max(axis=1)/np.abs(xs[both]).max(axis=1)).max() if both.any() else 0
            lost=int((~ka&oa).sum()); extra=int((ka&~oa).sum()); tot_lost+=lost; tot_extra+=extra
            print('%-14s shift %.1f %-9s accepted %d oracle %d lost %d extra %d worst rel err %.1e mean iters %.2f'%(tag,d,mode,ka.sum(),oa.sum(),lost,extra,err,it[ka].mean()+1),flush=True)
print('TOTAL lost',tot_lost,'extra',tot_extra)
