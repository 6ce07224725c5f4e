"""The prism -> bounds kernel against oracle/prism_oracle.py, bit for bit, on random scenes (cars outside the road,
negative speeds, zero durations' neighbours, twins, nested lateral extents) at three horizons.

    python tests/fuzz/prisms_vs_restatement.py SEED SCENES

Round 2: 3 000 scenes x 3 horizons, 0 differences.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sys, numpy as np, time
import torch
from oracle import prism_oracle as P
from spectral_amd.solver import BatchSolver
solver=BatchSolver(0)
def pack(scenes,Pm):
    arr=np.zeros((len(scenes),Pm,8))
    for b,cars in enumerate(scenes):
        for p,c in enumerate(cars):
            arr[b,p,:7]=[c["centre"][0],c["centre"][1],c["centre"][2],c.get("vel_s",0.0),c.get("vel_l",0.0),c.get("time",3.0),1.0]
    return arr
seed=int(sys.argv[1]); n=int(sys.argv[2])
rng=np.random.default_rng(seed)
scenes=[]
for _ in range(n):
    cars=[]
    nice=rng.random()<0.5
    q=(lambda v,k: round(float(v),k)) if nice else (lambda v,k: float(v))
    for r in range(int(rng.integers(1,5))):
        ahead=rng.random()<0.5
        cars.append(dict(centre=(q(rng.uniform(-5,60),1), q(rng.uniform(-6.0,12.0),2), 0 if ahead else q(rng.uniform(0.1,6.5),1)),
                         vel_s=q(rng.choice([0.0,rng.uniform(0,12),rng.uniform(-3,0)]),1)+(0.005 if nice and rng.random()<0.5 else 0.0),
                         vel_l=float(rng.choice([0.0,0.25,-0.25,1.0,-1.0])), time=float(rng.choice([0.5,1.0,3.0,4.0,7.0]))))
    if rng.random()<0.1 and len(cars)>1: cars[1]=dict(cars[0])     # identical twins
    if rng.random()<0.1 and len(cars)>1: cars[1]=dict(cars[0],centre=(cars[0]["centre"][0]+3.0,cars[0]["centre"][1],cars[0]["centre"][2]))  # same lateral extent
    scenes.append(cars)
bad=0; t0=time.time()
for N in (71,201,11):
    Omax=9
    sb,lb,nn=solver.prism_bounds(torch.from_numpy(pack(scenes,4)),N,Omax); torch.cuda.synchronize()
    sb,lb,nn=sb.cpu().numpy(),lb.cpu().numpy(),nn.cpu().numpy()
    for b,cars in enumerate(scenes):
        try: want=P.prism_bounds(cars,N)
        except Exception as e:
            print('oracle raised',b,N,repr(e)[:80]); bad+=1; continue
        if len(want)>Omax:
            if nn[b]!=-1 and nn[b]!=len(want): bad+=1; print('count(overflow)',b,N,nn[b],len(want))
            continue
        if nn[b]!=len(want): bad+=1; print('count',b,N,nn[b],len(want)); continue
        for j,(ws,wl) in enumerate(want):
            if not (np.array_equal(lb[b,j],np.array(wl)) and np.array_equal(sb[b,j],np.array(ws))):
                bad+=1; d=np.abs(sb[b,j]-np.array(ws)); print('field',b,N,j,np.nanmax(d), cars); break
print('scenes',n,'x 3 horizons, mismatches',bad,'seconds %.1f'%(time.time()-t0))
