"""Knot-level end-to-end timing: device corridor stage + ragged QP solve on jittered copies of a bundled
corridor file (CB_IN, default c_road_s1_3) -- the SURVEY 8(f) rank-1 widening.  CB_B candidates."""
import sys, os, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spectral_amd import knots, synth, layout as L
from spectral_amd.solver import BatchSolver
GOLD=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests','golden')
W=np.loadtxt(GOLD+'/inputs/weights.txt')
solver=BatchSolver(0); d=solver.device
B=int(os.environ.get('CB_B','65536')); name=os.environ.get('CB_IN','c_road_s1_3')
kb=knots.jittered(knots.parse_corridor_file(GOLD+'/inputs/%s.txt'%name),B,seed=3)
sh=synth.shared_params(0,weights=W); sh.ds_ref,sh.dl_ref=kb.header['ds_ref'],kb.header['dl_ref']
sh.dds,sh.ddds,sh.ddl,sh.dddl=kb.header['dds'],kb.header['ddds'],kb.header['ddl'],kb.header['dddl']
f=lambda a: torch.from_numpy(np.ascontiguousarray(a,dtype=np.float64)).to(d)
ins=[f(kb.s_bounds),f(kb.l_bounds),f(kb.ds_bounds),f(kb.dl_bounds),f(kb.s_ref),f(kb.l_ref)]
st=16
rec=dict(B=B,seg_stride=st,seg=torch.zeros((L.NUM_SEG_FIELDS,B,st),dtype=torch.float64,device=d),seg_count=torch.zeros(B,dtype=torch.int32,device=d),init=f(kb.init),ref_end=torch.zeros((B,2),dtype=torch.float64,device=d),dl_bounds=torch.zeros((B,10),dtype=torch.float64,device=d))
stream=torch.cuda.current_stream(d).cuda_stream
def corr(): solver.ctx.corridor_batch_device(0,B,kb.N,kb.num_obs,kb.delta,*ins,st,rec['seg'],rec['seg_count'],rec['ref_end'],rec['dl_bounds'],stream=stream)
for _ in range(2): corr(); out=solver.solve_ragged(rec,sh)
torch.cuda.synchronize()
tc=[];ts=[]
for _ in range(5):
    a=torch.cuda.Event(enable_timing=True);b=torch.cuda.Event(enable_timing=True);c=torch.cuda.Event(enable_timing=True)
    a.record(); corr(); b.record(); out=solver.solve_ragged(rec,sh); c.record(); torch.cuda.synchronize()
    tc.append(a.elapsed_time(b)); ts.append(b.elapsed_time(c))
inbytes=sum(t.numel()*8 for t in ins); outbytes=L.NUM_SEG_FIELDS*B*10*8
cnt=rec['seg_count'].cpu().numpy(); stt=out['status'].cpu().numpy()
print(name,'B',B,'N',kb.N,'obs',kb.num_obs,'corridor ms min %.3f'%min(tc),'-> %.1f GB/s of %.0f MB input'%(inbytes/min(tc)/1e6,inbytes/1e6),'| ragged solve ms min %.3f'%min(ts),'| end-to-end %.3e cand/s'%(B/(min(tc)+min(ts))*1e3),'cnt',dict(zip(*np.unique(cnt,return_counts=True))),'solved %.3f'%np.mean(stt>0))
