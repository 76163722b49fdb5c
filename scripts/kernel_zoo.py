#!/usr/bin/env python3
"""Launches every kernel of the library a few times at a realistic size so that ONE
`rocprofv3 --kernel-trace --stats` run gives the average duration of each (profiles/r01/kernel_zoo_stats.csv),
and prints the algorithmic HBM bytes / flops of each launch next to it."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from steered_mixture_of_experts_amd import blocks as blk                                          # noqa: E402
from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig, SharedConfig, SharedEngine   # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _warm import warm_block_engine, warm_shared_engine                                                                       # noqa: E402


def block_mode(B, shape, C, kpd, reps, **kw):
    K, N = int(np.prod(kpd)), int(np.prod(shape))
    b = blk.synthetic_blocks(B, shape, C, 3)
    p = {k: torch.from_numpy(v).cuda() for k, v in blk.init_block_params(b, kpd, True, kw.get("train_inverse_cov", False)).items()}
    eng = BlockEngine(EngineConfig(block_shape=shape, channels=C, kernels=K, use_yuv=(C == 3), quantize_pis=True, lr_steer=0.01, **kw))
    T = torch.from_numpy(blk.to_planar(b)).cuda()
    st = eng.new_adam_state(p)
    act = torch.full((B,), (1 << K) - 1, dtype=torch.int32, device="cuda")
    best = {k: v.clone() for k, v in p.items()}
    bl = torch.full((B,), 1e9, device="cuda")
    warm_block_engine(eng, T, p, act, iters=(300 if kw.get("ssim_opt") else 600))      # other tiling: its own row in the trace
    for _ in range(reps):
        out = eng.forward(T, p, act, want_recon=True, want_argmax=True, want_gate=True)
        eng.fit(T, p, st, act, 100)
        eng.update_kernel_list(p, act)
        eng.checkpoint_best(out["loss"], bl, p, best)
        eng.reduce_scalars(out["loss"], out["sse"], act)
    torch.cuda.synchronize()
    P = 1 + len(shape) + 2 * len(shape) ** 2 + len(shape) * C + C
    info = {"blocks": B, "N": N, "K": K, "C": C,
            "forward_bytes": B * (4 * C * N * 2 + N + 4 * K * N + 4 * K * P),          # targets + recon + argmax + gate + params
            "fit_alg_bytes_per_launch": B * N * 100 * (4 * (len(shape) + C) + 24 * K * P / N)}
    eng.close()
    return info


def shared_mode(reps, **kw):
    shape, bs, C, kpd = (512, 512), (32, 32), 1, [12, 12]
    b = blk.synthetic_blocks(1024, (16, 16), C, 9)
    img = blk.blocks_to_image(b, shape, (16, 16))
    p0 = {k: v[0] for k, v in blk.init_block_params(img[None], kpd).items()}
    K = p0["pis"].shape[0]
    eng = SharedEngine(SharedConfig(image_shape=shape, batch_shape=bs, channels=C, kernels=K, quantize_pis=True, lr_steer=0.01, **kw))
    tb, _ = blk.image_to_blocks(img, bs)
    T = torch.from_numpy(blk.to_planar(tb)).cuda()
    dp = {k: torch.from_numpy(v).cuda() for k, v in p0.items()}
    st = eng.new_adam_state(dp)
    lists = eng.new_lists()
    eng.forward(T, dp, lists, want_recon=False)
    for _ in range(reps):
        eng.fit(T, dp, st, lists, 100)
        eng.update_kernel_list(dp, lists)
        eng.forward(T, dp, lists, want_recon=True)
    torch.cuda.synchronize()
    eng.close()


if __name__ == "__main__":
    out = {"headline": block_mode(65536, (16, 16), 1, [2, 2], 3),
           "ssim16": block_mode(65536, (16, 16), 1, [2, 2], 2, ssim_opt=True),
           "quant3": block_mode(65536, (16, 16), 1, [2, 2], 2, quantization_mode=3),
           "invcov": block_mode(65536, (16, 16), 1, [2, 2], 2, train_inverse_cov=True)}
    shared_mode(2)
    shared_mode(1, ssim_opt=True)
    print(json.dumps(out))
