import sys; sys.path.insert(0,'/root/repo')
import ctypes as C, torch
from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig
torch.zeros(1).cuda()
for ch,B in ((3,32400),(1,65536)):
    for q in (False,True):
        e=BlockEngine(EngineConfig(block_shape=(16,16),channels=ch,kernels=4,use_yuv=ch==3,quantize_pis=q))
        print("C",ch,"qpis",q,e.fit_variant(B),"waves/CU",e.lib.smoe_fit_occupancy(e._h,B)); e.close()
p=torch.cuda.get_device_properties(0); print(p.name, getattr(p,'shared_memory_per_multiprocessor',None), getattr(p,'shared_memory_per_block_optin',None))
