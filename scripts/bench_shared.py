#!/usr/bin/env python3
"""Throughput of the shared-kernel image mode on one MI355X: the reference's
whole-image fit (global kernel grid, per-batch kernel lists).  Secondary measurement -- the
headline bench line is bench.py.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from steered_mixture_of_experts_amd import blocks as blk                          # noqa: E402
from steered_mixture_of_experts_amd.engine import SharedConfig, SharedEngine       # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _warm import warm_shared_engine                                                                       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--image", type=int, nargs="+", default=[512, 512])
    ap.add_argument("--batch", type=int, nargs="+", default=[32, 32])
    ap.add_argument("--kernels-per-dim", type=int, nargs="+", default=[12, 12])
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--quantization-mode", type=int, default=0, help="0, 2 or 3 (fake-quantised graph)")
    args = ap.parse_args()
    shape, bs, C = tuple(args.image), tuple(args.batch), args.channels
    d = len(shape)
    g = [-(-s // 16) for s in shape[:2]]
    b = blk.synthetic_blocks(int(np.prod(g)) * (1 if d == 2 else shape[2] // 4), (16, 16) if d == 2 else (16, 16, 4), C, 20260009)
    img = blk.blocks_to_image(b, shape, (16, 16) if d == 2 else (16, 16, 4))
    kpd = list(args.kernels_per_dim)
    p0 = {k: v[0] for k, v in blk.init_block_params(img[None], kpd).items()}
    K = p0["pis"].shape[0]
    eng = SharedEngine(SharedConfig(image_shape=shape, batch_shape=bs, channels=C, kernels=K, use_yuv=(C == 3),
                                    quantization_mode=args.quantization_mode, quantize_pis=args.quantization_mode >= 2))
    tb, _ = blk.image_to_blocks(img, bs)
    T = torch.from_numpy(blk.to_planar(tb)).cuda()
    dp = {k: torch.from_numpy(v).cuda() for k, v in p0.items()}
    st = eng.new_adam_state(dp)
    lists = eng.new_lists()
    f0 = eng.forward(T, dp, lists, want_recon=False)
    npx = int(np.prod(shape))
    psnr0 = -10 * np.log10(float(f0["sse"].sum()) / (npx * C))
    eng.fit(T, dp, st, lists, 5)
    torch.cuda.synchronize()
    warm_shared_engine(eng, T, dp, lists, iters=max(200, int(30e-3 / 30e-6 * 512 * 512 / npx)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    done = 0
    while done < args.steps:
        n = min(100, args.steps - done)
        eng.fit(T, dp, st, lists, n)
        eng.update_kernel_list(dp, lists)
        eng.forward(T, dp, lists, want_recon=False)
        done += n
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    f1 = eng.forward(T, dp, lists, want_recon=False, update_lists=False)
    psnr1 = -10 * np.log10(float(f1["sse"].sum()) / (npx * C))
    bits = lists.cpu().numpy().view(np.uint32)
    kact = float(np.mean([bin(int(x)).count("1") for row in bits for x in row]) * bits.shape[1])
    out = {"mode": "shared-kernel image fit (SURVEY 8(f-1))", "quantization_mode": args.quantization_mode, "image": list(shape), "batch": list(bs), "channels": C,
           "kernels": K, "steps": args.steps, "ms_per_step": round(ms / args.steps, 4),
           "value": round(npx * args.steps / (ms * 1e-3) / 1e6, 1), "unit": "Mpixel-iters/s",
           "mean_listed_kernels_per_batch_at_end": round(kact, 2),
           "pixel_kernel_evals_per_s_G": round(npx * kact * args.steps / (ms * 1e-3) / 1e9, 2),
           "psnr_db": [round(float(psnr0), 3), round(float(psnr1), 3)]}
    if args.cpu_iters:
        from oracle import smoe_oracle as o
        cfg = o.OracleConfig(block_shape=bs, channels=C, kernels=K, use_yuv=(C == 3))
        coords = o.global_batch_coords(shape, bs)
        tgt = tb.reshape(tb.shape[0], -1, C)
        mask = np.stack([(bits[:, k >> 5] >> np.uint32(k & 31)) & 1 for k in range(K)], axis=1).astype(bool)
        pn = {k: v.cpu().numpy()[None] for k, v in dp.items()}
        t0 = time.perf_counter()
        for _ in range(args.cpu_iters):
            o.shared_pass(pn, tgt, coords, mask, cfg, np.float32, want_grads=True)
        dt = (time.perf_counter() - t0) / args.cpu_iters
        out["cpu_numpy_restatement_Mpixel_iters_s"] = round(npx / dt / 1e6, 3)
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
