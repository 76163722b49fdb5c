"""Loss trajectory of ONE 16x16 block (BASELINE configs[0]) at the CLI defaults: GPU (one iteration per launch) against the fp32 and
fp64 restatements, iteration by iteration.  Prints where the trajectories part.  usage: diag_cfg1.py [seed ...]"""
import sys
import numpy as np
import torch
from oracle import smoe_oracle as o
from steered_mixture_of_experts_amd.blocks import synthetic_blocks
from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig

shape, C, kpd, K, N = (16, 16), 1, [2, 2], 4, 256
coords = o.block_coords(shape)
seeds = [int(x) for x in sys.argv[1:]] or [3, 5]
for seed in seeds:
    b = synthetic_blocks(1, shape, C, 20260500 + seed)
    tgt = b.reshape(1, -1, C)
    p = o.init_params(b, kpd)
    cfg = o.OracleConfig(block_shape=shape, channels=C, kernels=K, quantize_pis=True)
    _, _, i32 = o.fit(p, tgt, coords, cfg, 200, val_iter=100, dtype=np.float32, record_every=1)
    _, _, i64 = o.fit(p, tgt, coords, cfg, 200, val_iter=100, dtype=np.float64, record_every=1)
    l32 = np.array([t[1][0] for t in i32["trace"]]); l64 = np.array([t[1][0] for t in i64["trace"]])
    eng = BlockEngine(EngineConfig(block_shape=shape, channels=C, kernels=K, quantize_pis=True))
    T = torch.from_numpy(np.ascontiguousarray(np.transpose(tgt, (0, 2, 1)))).cuda()
    dp = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda() for k, v in p.items()}
    act = torch.full((1,), 15, dtype=torch.int32, device="cuda")
    st = eng.new_adam_state(dp)
    f0 = eng.forward(T, dp, act, want_recon=False)
    lo = torch.zeros(1, device="cuda")
    lg = []
    for i in range(1, 201):
        eng.fit(T, dp, st, act, 1, loss0=f0["loss"], loss_out=lo)
        lg.append(float(lo.item()))
        if i % 100 == 0:
            eng.update_kernel_list(dp, act)
    lg = np.array(lg)
    r32 = np.abs(lg - l32) / np.abs(l32); r64 = np.abs(l64 - l32) / np.abs(l32)
    first = lambda r, th: int(np.argmax(r > th)) + 1 if (r > th).any() else -1
    print(f"seed {seed} ({eng.fit_variant(1)}): first iteration with |dloss|/loss > 1e-4 / 1e-2: gpu-vs-fp32 {first(r32, 1e-4)} / {first(r32, 1e-2)}   fp64-vs-fp32 {first(r64, 1e-4)} / {first(r64, 1e-2)}")
    for i in list(range(0, 200, 10)) + [199]:
        print(f"   it {i + 1:3d}  fp32 {l32[i]:.6e}  fp64 {l64[i]:.6e}  gpu {lg[i]:.6e}")
    eng.close()
