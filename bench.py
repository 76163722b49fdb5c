#!/usr/bin/env python3
"""bench.py -- Mpixel-iters/s of the per-block SMoE fit hot path on MI355X.

A "step" is ONE training iteration (forward + analytic backward + prune + TF1 Adam) over the
whole batch of independent 16x16 blocks, K=4 kernels, grayscale (BASELINE.json configs[1]
shape).  The batch is ``--blocks`` synthetic blocks per GPU (default 65536 = 64 images of
512x512), resident in HBM before the timed region.  Steps are issued as launches of
``--iters-per-launch`` iterations (the fit kernel keeps parameters/Adam state on chip across
the iterations of one launch).  One process per GPU; blocks are sharded across ranks with
no data-path collective; the only collective is the 3-scalar RCCL all-reduce for the global
loss / PSNR after the timed region.

Prints ONE JSON line on rank 0 (contract: see the task statement / DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from steered_mixture_of_experts_amd import blocks as blk            # noqa: E402
from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_px_iter(N, K, d, C):
    """the reference's boundary layout -- (d+C) fp32 per pixel fed per pass
    (smoe.py:545) + {param, m, v} read+written once per block per iteration."""
    P = 1 + d + 2 * d * d + d * C + C
    return 4.0 * (d + C) + 24.0 * K * P / N


def cpu_baseline(block_shape, C, kpd, blocks_np, n_iters, budget_s):
    """Times the plain-C restatement (oracle/smoe_oracle.c, kind "port") on this host's cores
    on a bounded sample of the same workload.  The only place bench.py touches oracle/."""
    from oracle import c_oracle as co
    from oracle import smoe_oracle as o
    K = int(np.prod(kpd))
    cfg = o.OracleConfig(block_shape=tuple(block_shape), channels=C, kernels=K, quantize_pis=True)   # CLI default -qp
    threads = min(os.cpu_count() or 1, 64)
    N = int(np.prod(block_shape))
    coords = np.ascontiguousarray(o.block_coords(block_shape).T)

    def run(nb, iters):
        sub = blocks_np[:nb]
        p = {k: np.ascontiguousarray(v) for k, v in blk.init_block_params(sub, kpd).items()}
        m = {k: np.zeros_like(v) for k, v in p.items()}
        v = {k: np.zeros_like(v) for k, v in p.items()}
        act = np.full(nb, (1 << K) - 1, np.uint32)
        T = blk.to_planar(sub)
        co.forward(cfg, coords, T, p, act, want_recon=False, threads=threads)
        bp = np.array([cfg.beta1, cfg.beta2], np.float32)
        t0 = time.perf_counter()
        co.fit(cfg, coords, T, p, m, v, act, iters, bp, threads=threads)
        return time.perf_counter() - t0

    cal_nb = min(len(blocks_np), 64 * threads)
    t = run(cal_nb, 5)
    rate = cal_nb * N * 5 / max(t, 1e-6)                     # px-iters / s
    nb = int(min(len(blocks_np), max(threads, rate * budget_s / (N * n_iters))))
    nb = max(threads, nb - nb % threads)
    t = run(nb, n_iters)
    return {"value": round(nb * N * n_iters / t / 1e6, 3), "unit": "Mpixel-iters/s", "cores": threads,
            "kind": "port", "sample": f"{nb} of the bench blocks x {n_iters} iterations, "
            f"oracle/smoe_oracle.c (fp32 scalar C, OpenMP over blocks), {t:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100,
                    help="untimed steps; the default is one full launch so that every fit launch of a default run has the same size")
    ap.add_argument("--blocks", type=int, default=65536, help="blocks per GPU")
    ap.add_argument("--iters-per-launch", type=int, default=100)
    ap.add_argument("--tiling", type=int, default=0, help="lanes per block: 0 auto, 16, 64")
    ap.add_argument("--block-shape", type=int, nargs="+", default=[16, 16])
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--kernels-per-dim", type=int, nargs="+", default=[2, 2])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-quantize-pis", action="store_true",
                    help="diagnostic: run the constructor default (pis not fake-quantised) instead of the CLI default")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary single-image measurement")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0)
    ap.add_argument("--clock-warm-iters", type=int, default=600,
                    help="untimed iterations on SCRATCH copies of the parameters before the W warm-up steps: the engine "
                         "clocks of an idle MI355X take ~25 ms of load to settle (launch times 7.1, 5.3, 5.1, 4.9, 4.7, "
                         "4.6, 4.6 ... ms, scripts/launch_times.py), a real fit runs thousands of iterations.  Runs the "
                         "OTHER tiling of the kernel so that a kernel trace of this command averages steady launches "
                         "only.  0 = off")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("SMOE_BENCH_FORCE_DIST") == "1":     # the env switch exercises the RCCL path at N=1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    n_gpus = world

    shape = tuple(args.block_shape)
    d, C = len(shape), args.channels
    kpd = list(args.kernels_per_dim)
    K = int(np.prod(kpd))
    N = int(np.prod(shape))
    B = args.blocks
    use_yuv = (C == 3)

    # ---- synthetic inputs, resident in HBM before timing ------------------------------
    blocks_np = blk.synthetic_blocks(B, shape, C, 20260002 + rank)
    params_np = blk.init_block_params(blocks_np, kpd)
    # CLI defaults (smoe_test.py:262-352): -qp/--quantize_pis defaults to True, so the graph fake-quantises the pis
    eng = BlockEngine(EngineConfig(block_shape=shape, channels=C, kernels=K, use_yuv=use_yuv,
                                   quantize_pis=not args.no_quantize_pis))
    if args.tiling:
        eng.set_tiling(args.tiling)
    dev = eng.device
    target = torch.from_numpy(blk.to_planar(blocks_np)).to(dev)
    params = {k: torch.from_numpy(v).to(dev) for k, v in params_np.items()}
    state = eng.new_adam_state(params)
    active = torch.full((B,), (1 << K) - 1, dtype=torch.int32, device=dev)
    diverged = torch.zeros((B,), dtype=torch.int32, device=dev)
    f0 = eng.forward(target, params, active, want_recon=False)          # iteration-0 eval pass
    loss0 = f0["loss"].clone()
    psnr0 = None

    def global_scalars(loss, sse):
        s = eng.reduce_scalars(loss, sse, active)
        if dist is not None:
            dist.all_reduce(s)                                            # RCCL, 3 doubles
        return s.cpu().numpy()

    s0 = global_scalars(f0["loss"], f0["sse"])
    psnr0 = -10.0 * np.log10(s0[1] / (n_gpus * B * N * C))

    ipl = max(1, min(args.iters_per_launch, args.steps))

    def run_steps(k, events=None):
        done = 0
        while done < k:
            n = min(ipl, k - done)
            if events is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
            eng.fit(target, params, state, active, n, diverged=diverged, loss0=loss0)
            if events is not None:
                e1.record()
                events.append((e0, e1, n))
            done += n

    variant = eng.fit_variant(B)
    if args.clock_warm_iters > 0:                  # scratch state: the measured trajectory starts from the same point
        p2 = {k: v.clone() for k, v in params.items()}
        st2 = eng.new_adam_state(p2)
        a2, d2 = active.clone(), diverged.clone()
        eng.set_tiling(64 if "_g16" in variant else 16)
        if not eng.fit_variant(B):                 # no second tiling for this shape: warm with the measured kernel
            eng.set_tiling(args.tiling)
        left = args.clock_warm_iters
        while left > 0:
            eng.fit(target, p2, st2, a2, min(100, left), diverged=d2, loss0=loss0)
            left -= 100
        torch.cuda.synchronize()
        eng.set_tiling(args.tiling)
        assert eng.fit_variant(B) == variant
        del p2, st2, a2, d2
    run_steps(args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    events = []
    t0 = time.perf_counter()
    run_steps(args.steps, events)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([t_local], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_wall = float(tt.item())
    else:
        t_wall = t_local

    # per-launch device time from HIP events on the launch stream (full-size launches only)
    full = [e0.elapsed_time(e1) for (e0, e1, n) in events if n == ipl]
    launch_ms = float(np.mean(full)) if full else float("nan")

    # ---- final quality (outside the timed region) --------------------------------------
    f1 = eng.forward(target, params, active, want_recon=False)
    s1 = global_scalars(f1["loss"], f1["sse"])
    psnr1 = -10.0 * np.log10(s1[1] / (n_gpus * B * N * C))
    sse_blocks = f1["sse"].cpu().numpy()
    psnr_med = float(np.median(-10.0 * np.log10(np.maximum(sse_blocks, 1e-12) / (N * C))))
    n_div = int(diverged.sum().item())

    # ---- secondary: BASELINE configs[1] literally (ONE 512x512 image = 1024 blocks), rank 0 only ----
    single = None
    if rank == 0 and not args.no_extras and shape == (16, 16) and C == 1 and B >= 1024:
        Bs = 1024
        eng1 = BlockEngine(EngineConfig(block_shape=shape, channels=C, kernels=K, use_yuv=use_yuv, quantize_pis=True))
        p1 = {k: torch.from_numpy(v[:Bs].copy()).to(dev) for k, v in params_np.items()}
        st1 = eng1.new_adam_state(p1)
        a1 = torch.full((Bs,), (1 << K) - 1, dtype=torch.int32, device=dev)
        t1 = target[:Bs].contiguous()
        eng1.forward(t1, p1, a1, want_recon=False)
        eng1.fit(t1, p1, st1, a1, min(20, args.steps))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        done = 0
        while done < args.steps:
            n = min(ipl, args.steps - done)
            eng1.fit(t1, p1, st1, a1, n)
            done += n
        e1.record()
        torch.cuda.synchronize()
        ms1 = e0.elapsed_time(e1)
        single = {"workload": "one 512x512 grayscale image = 1024 blocks of 16x16, K=4 (BASELINE configs[1] literally)",
                  "value": round(Bs * N * args.steps / (ms1 * 1e-3) / 1e6, 1), "unit": "Mpixel-iters/s",
                  "ms_total": round(ms1, 4), "kernel_variant": eng1.fit_variant(Bs),
                  "note": "1024 blocks cannot fill 256 CUs (1 wavefront per SIMD); the headline batch is 64 such images"}
        eng1.close()

    if rank == 0:
        total_px_iters = float(n_gpus) * B * N * args.steps
        value = total_px_iters / t_wall / 1e6
        bpi = algorithmic_bytes_per_px_iter(N, K, d, C)
        achieved = (B * N * ipl * bpi) / (launch_ms * 1e-3) / 1e9 if launch_ms == launch_ms else None
        out = {
            "metric": "Mpixel-iters/s (SMoE fit: forward + analytic backward + TF1 Adam), 16x16 blocks / 4 kernels",
            "value": round(value, 1), "unit": "Mpixel-iters/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(t_wall * 1e3 / args.steps, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{B} independent {'x'.join(map(str, shape))} blocks per GPU "
                                   f"(= {B * N // (512 * 512)} images of 512x512), C={C}, K={K} kernels/block, "
                                   f"{args.steps} Adam iterations, CLI-default hyper-parameters",
                       "blocks_per_gpu": B, "block_shape": list(shape), "channels": C, "kernels": K,
                       "iters_per_launch": ipl, "clock_warm_iters": args.clock_warm_iters, "kernel_variant": eng.fit_variant(B),
                       "parallelism": f"blocks sharded over {n_gpus} rank(s), no data-path collective"},
            "final_psnr_db": round(float(psnr1), 3), "initial_psnr_db": round(float(psnr0), 3),
            "final_median_block_psnr_db": round(psnr_med, 3), "diverged_blocks": n_div,
            "roofline": {"bound": "hbm", "achieved": None if achieved is None else round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": None,
                         "algorithmic_bytes_per_px_iter": bpi,
                         "kernel_ms_per_launch": None if launch_ms != launch_ms else round(launch_ms, 4)},
        }
        if single is not None:
            out["single_image"] = single
        if shape == (16, 16) and C == 1 and K == 4:
            # secondary view: the kernel's real bound.  PMC of fit_kernel<2,1,4,16,4,1>: 2 021 VALU instructions per
            # wavefront-iteration of 16 x 64 pixels = 126 per pixel-iteration and lane; the pixel loop is 112 instructions
            # per step with 30 FMAs (2 flop), the per-iteration phases about a third FMAs -> ~175 fp32 flop per
            # pixel-iteration.
            flop_per_px_iter = 175.0
            tf = value * 1e6 * flop_per_px_iter / 1e12 / n_gpus
            out["valu"] = {"bound": "fp32 VALU issue", "flop_per_pixel_iter_est": flop_per_px_iter,
                           "achieved_tflops_per_gpu_est": round(tf, 1), "peak_tflops": 157.3,
                           "frac_est": round(tf / 157.3, 3)}
        out["roofline"]["note"] = ("contract figure (SURVEY 8(d)): algorithmic bytes of the reference boundary layout / "
                                   "kernel time; the persistent kernel keeps the working set on chip, measured HBM "
                                   "traffic is in 'traffic'; the real bound is fp32 VALU issue (DESIGN.md section 4)")
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                ent = tj.get(out["config"]["kernel_variant"])
                if ent and ent.get("blocks") == B and ent.get("iters_per_launch") == ipl:
                    out["roofline"]["traffic"] = ent["hbm_bytes_per_launch"]
            except Exception:
                pass
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(shape, C, kpd, blocks_np, args.steps, args.cpu_budget_s)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
