"""Throughput of the loss-weight variant of the fit kernels (ragged images: every block carries per-pixel loss weights)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from steered_mixture_of_experts_amd import blocks as blk
from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
for B, shape, C, kpd in ((65536, (16, 16), 1, [2, 2]), (2040, (32, 32), 3, [2, 4]), (32400, (16, 16), 3, [2, 2])):
    b = blk.synthetic_blocks(B, shape, C, 3)
    K = int(np.prod(kpd)); N = int(np.prod(shape))
    T = torch.from_numpy(blk.to_planar(b)).cuda()
    for with_lw in (False, True):
        p = {k: torch.from_numpy(v).cuda() for k, v in blk.init_block_params(b, kpd).items()}
        eng = BlockEngine(EngineConfig(block_shape=shape, channels=C, kernels=K, use_yuv=(C == 3), quantize_pis=True))
        st = eng.new_adam_state(p); act = torch.full((B,), (1 << K) - 1, dtype=torch.int32, device="cuda")
        lw = torch.ones((B, N), device="cuda") if with_lw else None
        if with_lw: lw[:, -N // 8:] = 0.0
        for _ in range(4): eng.fit(T, p, st, act, 100, loss_w=lw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.fit(T, p, st, act, 100, loss_w=lw); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"{shape} C{C} K{K} B={B} loss_w={with_lw}: {ms:.3f} ms per 100 iterations = {B * N * 100 / ms / 1e6:.1f} Gpx-it/s ({eng.fit_variant(B)})")
        eng.close()
