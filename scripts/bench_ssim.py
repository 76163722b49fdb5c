"""Throughput of the SSIM-loss fit (ssim_opt) next to the margin-loss fit on the headline workload shape
(blocks of 16x16, K = 4, C = 1), same tiling (one block per wavefront) for both plus the default tiling."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from steered_mixture_of_experts_amd import blocks as blk
from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig
from _warm import warm_block_engine


def run(B, shape, C, kpd, iters, ssim, tiling):
    K = int(np.prod(kpd))
    b = blk.synthetic_blocks(B, shape, C, 1)
    p = {k: torch.from_numpy(v).cuda() for k, v in blk.init_block_params(b, kpd).items()}
    eng = BlockEngine(EngineConfig(block_shape=shape, channels=C, kernels=K, use_yuv=(C == 3), lr_steer=0.01, ssim_opt=ssim))
    if tiling:
        eng.set_tiling(tiling)
    T = torch.from_numpy(blk.to_planar(b)).cuda()
    st = eng.new_adam_state(p)
    act = torch.full((B,), (1 << K) - 1, dtype=torch.int32, device="cuda")
    eng.fit(T, p, st, act, 5)
    torch.cuda.synchronize()
    warm_block_engine(eng, T, p, act, iters=(200 if ssim else 600), other_tiling=False, restore_tiling=tiling)
    t0 = time.perf_counter()
    eng.fit(T, p, st, act, iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    name = eng.fit_variant(B)
    eng.close()
    return {"ssim": ssim, "variant": name, "blocks": B, "shape": list(shape), "C": C, "K": K,
            "ms_per_iter": 1e3 * dt / iters, "Mpixel_iters_per_s": B * int(np.prod(shape)) * iters / dt / 1e6}


if __name__ == "__main__":
    out = []
    for shape, C, kpd, B in (((16, 16), 1, [2, 2], 65536), ((16, 16), 3, [2, 2], 32768), ((32, 32), 3, [2, 4], 2040),
                             ((8, 8), 3, [2, 2], 65536)):
        out.append(run(B, shape, C, kpd, 50, False, 0))
        out.append(run(B, shape, C, kpd, 50, False, 64))
        out.append(run(B, shape, C, kpd, 50, True, 0))
        if tuple(shape) == (16, 16):
            out.append(run(B, shape, C, kpd, 50, True, 64))
    for r in out:
        print(json.dumps(r))
