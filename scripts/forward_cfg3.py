"""Evaluation pass on 2 040 blocks of 32x32 / K = 8 / RGB (cfg3): time per output set."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from steered_mixture_of_experts_amd import blocks as blk
from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig
B, shape, C, kpd = 2040, (32, 32), 3, [2, 4]
K, N = 8, 1024
blocks = blk.synthetic_blocks(B, shape, C, 7)
p0 = blk.init_block_params(blocks, kpd)
eng = BlockEngine(EngineConfig(block_shape=shape, channels=C, kernels=K, use_yuv=True, quantize_pis=True))
T = torch.from_numpy(blk.to_planar(blocks)).cuda()
dp = {k: torch.from_numpy(v).cuda() for k, v in p0.items()}
act = torch.full((B,), 255, dtype=torch.int32, device="cuda")
for name, kw in (("loss only", dict(want_recon=False)), ("recon", dict(want_recon=True)),
                 ("recon+argmax+gate", dict(want_recon=True, want_argmax=True, want_gate=True))):
    for _ in range(30):
        eng.forward(T, dp, act, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        eng.forward(T, dp, act, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    byt = B * N * C * 4 + B * K * 20 * 4 + (B * N * C * 4 if kw.get("want_recon") else 0) + (B * N * K * 4 + B * N if kw.get("want_gate") else 0)
    print(json.dumps({"outputs": name, "ms": round(ms, 4), "MB": round(byt / 1e6, 1), "GBps": round(byt / ms / 1e6, 1)}))
