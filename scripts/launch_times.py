import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from steered_mixture_of_experts_amd import blocks as blk
from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig
B=65536
blocks=blk.synthetic_blocks(B,(16,16),1,20260002); p0=blk.init_block_params(blocks,[2,2])
eng=BlockEngine(EngineConfig(block_shape=(16,16),channels=1,kernels=4,quantize_pis=True))
T=torch.from_numpy(blk.to_planar(blocks)).cuda(); dp={k:torch.from_numpy(v).cuda() for k,v in p0.items()}
st=eng.new_adam_state(dp); act=torch.full((B,),15,dtype=torch.int32,device='cuda')
ev=[]
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); eng.fit(T,dp,st,act,100); e1.record(); ev.append((e0,e1))
torch.cuda.synchronize()
ts=[round(a.elapsed_time(b),3) for a,b in ev]; print(ts if len(ts) <= 100 else ts[:12] + ["... every 100th:"] + ts[99::100])
