import sys,time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from steered_mixture_of_experts_amd import blocks as blk
from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig
B=65536; shape=(16,16); b=blk.synthetic_blocks(B,shape,1,3)
for kw in ({}, {"quantize_pis":True}, {"quantization_mode":2}, {"quantization_mode":3}):
    p={k:torch.from_numpy(v).cuda() for k,v in blk.init_block_params(b,[2,2]).items()}
    eng=BlockEngine(EngineConfig(block_shape=shape,channels=1,kernels=4,lr_steer=0.01,**kw))
    T=torch.from_numpy(blk.to_planar(b)).cuda(); st=eng.new_adam_state(p); act=torch.full((B,),15,dtype=torch.int32,device="cuda")
    eng.fit(T,p,st,act,20); torch.cuda.synchronize(); t=time.perf_counter(); eng.fit(T,p,st,act,100); torch.cuda.synchronize()
    print(kw, "ms/100it %.2f"%(1e3*(time.perf_counter()-t))); eng.close()
