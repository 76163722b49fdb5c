"""Where the cycles of a fit iteration go (diagnostic build: make -C steered_mixture_of_experts_amd/csrc EXTRA=-DSMOE_PHASE_CLOCKS=1
after touching smoe_block.hip.h; lane 0 of every wavefront of workgroup 0 sums the shader-clock cycles per phase).
usage: python scripts/phase_clocks.py   ->  table per (blocks, tiling)"""
import numpy as np
import torch

from steered_mixture_of_experts_amd import blocks as blk
from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig

PH = ["load+derive+hoist", "pixel loop+complete", "finish_partials", "reduce", "pair hand-off", "owner post", "adam", "write+refresh"]
PH_DUO = ["load+derive+hoist", "pixel loop+complete", "partial writes", "barrier A", "row sums+publish", "barrier B", "gradient+adam+write", "barrier C"]
shape, C, kpd, K = (16, 16), 1, [2, 2], 4
for B, tiling in [(1024, 264), (1024, 128), (1024, 64), (2048, 64), (4096, 32), (4096, 16), (65536, 16)]:
    b = blk.synthetic_blocks(B, shape, C, 20260002)
    T = torch.from_numpy(blk.to_planar(b)).cuda()
    eng = BlockEngine(EngineConfig(block_shape=shape, channels=C, kernels=K, quantize_pis=True))
    eng.set_tiling(tiling)
    p = {k: torch.from_numpy(v).cuda() for k, v in blk.init_block_params(b, kpd).items()}
    st = eng.new_adam_state(p)
    act = torch.full((B,), 15, dtype=torch.int32, device="cuda")
    loss = torch.zeros(B, device="cuda")
    n = 100
    for _ in range(3):
        eng.fit(T, p, st, act, n, loss_out=loss)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.fit(T, p, st, act, n, loss_out=loss); e1.record(); torch.cuda.synchronize()
    c = loss.cpu().numpy()[B // 2:B // 2 + 32].reshape(4, 8) / n
    print(f"B={B} tiling={tiling} {eng.fit_variant(B)}: {e0.elapsed_time(e1) * 1e3 / n:.2f} us per iteration (whole launch)")
    for w in range(4):
        if c[w].sum() > 0:
            names = PH_DUO if tiling == 264 else PH
            print(f"  wave {w}: total {c[w].sum():8.0f} clk/iter  " + "  ".join(f"{names[i]} {c[w][i]:.0f}" for i in range(8)))
    eng.close()
