"""Randomised GPU-vs-oracle sweep of the shared-kernel mode: image / batch shapes, halo, option combinations; one
evaluation pass and one accumulate + apply per case.  Prints every failing case; exit code = number of failures."""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import smoe_oracle as o                                                  # noqa: E402
from steered_mixture_of_experts_amd import blocks as blk                             # noqa: E402
from test_gpu_shared import QKW, _dev, _engine, _setup                               # noqa: E402

STATS = []


def one_case(rng, idx):
    d = 3 if rng.random() < 0.15 else 2
    C = int(rng.choice([1, 3]))
    if d == 3:
        bshape = (int(rng.choice([8, 16])), int(rng.choice([8, 16])), int(rng.choice([2, 4])))
        grid = (int(rng.integers(1, 3)), int(rng.integers(1, 4)), int(rng.integers(1, 3)))
        kpd = [int(rng.integers(2, 4)), int(rng.integers(2, 4)), 2]
    else:
        bshape = (int(rng.choice([8, 16, 24, 32])), int(rng.choice([8, 16, 32])))
        grid = (int(rng.integers(1, 5)), int(rng.integers(1, 5)))
        kpd = [int(rng.integers(2, 9)), int(rng.integers(2, 9))]
    shape = tuple(g * b for g, b in zip(grid, bshape))
    yuv = bool(C == 3 and rng.random() < 0.7)
    kw = {}
    mode = int(rng.choice([0, 0, 2, 3]))
    if mode:
        kw.update(quantization_mode=mode, quantize_pis=True, **QKW)
    elif rng.random() < 0.5:
        kw["quantize_pis"] = True
    if rng.random() < 0.4:
        kw["train_inverse_cov"] = True
    if d == 2 and min(bshape) >= 8 and rng.random() < 0.3:
        kw["ssim_opt"] = True
    if rng.random() < 0.4:
        kw.update(pis_l1=0.05, u_l1=0.002)
    if rng.random() < 0.2:
        kw["kernel_count_as_norm_l1"] = True
        kw.setdefault("pis_l1", 0.05)
    if rng.random() < 0.15:
        kw["radial_as"] = True
    centred = mode >= 2 and rng.random() < 0.3             # use_diff_center: the quantised variable is musX - grid
    if yuv and rng.random() < 0.3:
        kw["only_y_gamma"] = True
    ov = int(rng.choice([0, 0, 0, 2, 3])) if d == 2 else 0
    desc = dict(idx=idx, shape=shape, bshape=bshape, C=C, kpd=kpd, yuv=yuv, overlap=ov, **kw)
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, yuv, seed=100 + idx, **kw)
    if mode == 3 or kw.get("radial_as"):
        p["A_corr"] = p["A_corr"] * np.tril(np.ones((d, d), np.float32), -1)
    if kw.get("radial_as"):
        a0 = np.abs(p["A_diagonal"][0, :, 0, 0])
        p["A_diagonal"] = (a0[None, :, None, None] * np.eye(d)).astype(np.float32)
        p["A_corr"] = np.zeros_like(p["A_corr"])
    if kw.get("train_inverse_cov"):
        p["A_diagonal"] = (p["A_diagonal"] ** 2).astype(np.float32)
        p["A_corr"] = (p["A_corr"] * 0.3).astype(np.float32)
    lists = np.ones((NB, K), bool)
    grid = None
    if centred:
        grid = o.shared_init_params(img, kpd)["musX"][0].astype(np.float32)
        p["musX"] = (grid + rng.uniform(-0.02, 0.02, size=grid.shape))[None].astype(np.float32)
        cfg = o.OracleConfig(**{**cfg.__dict__, "mus_grid": grid[None]})
        desc["centred"] = True
    try:
        eng = _engine(shape, bshape, C, K, yuv, overlap=ov, **kw)
    except Exception as e:
        return desc, "refused: " + str(e)[:90]
    try:
        gdev = None
        if grid is not None:
            gdev = torch.from_numpy(grid).cuda()
            eng.set_center_grid(gdev)
        dp = _dev(p)
        dl = eng.new_lists()
        T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
        fw = eng.forward(T, dp, dl, want_recon=True, update_lists=False)
        torch.cuda.synchronize()
        recon = fw["recon"].cpu().numpy().transpose(0, 2, 1)
        P = o._bcast(p, NB)
        if mode:
            q32, back, _ = o.quantize_graph_params(P, cfg, np.float32)
            cfg0 = o.OracleConfig(**{**cfg.__dict__, "quantization_mode": 0, "quantize_pis": False})
            ref = o.forward(q32, tgt, coords, lists, cfg0, None, np.float64, want_grads=True, q_override=recon)
            ref["grads"] = o.route_quant_grads(ref["grads"], back, np.float64)
        else:
            ref = o.forward(P, tgt, coords, lists, cfg, None, np.float64, want_grads=True, q_override=recon)
        lerr = np.abs(fw["loss"].cpu().numpy() - ref["loss"]).max()
        g = {k: v.sum(axis=0) for k, v in ref["grads"].items()}
        st = eng.new_adam_state(dp)
        eng.accumulate(T, dp, dl)
        eng.apply(dp, st)
        torch.cuda.synchronize()
        frac = (np.clip(ref["y"], 0, 1) * 255 + 0.5) % 1.0
        shaky = ((frac < 3e-4) | (frac > 1 - 3e-4)).any() or (np.abs(ref["w"] - 0.5 / 256) < 1e-6).any() or \
            ((np.abs(ref["y"]) < 1e-6) | (np.abs(ref["y"] - 1) < 1e-6)).any()
        worst = ("", 0.0)
        for name in o.PARAM_NAMES:
            scale = np.abs(g[name]).max() + 1e-30
            err = np.abs(st.m[name].cpu().numpy() / 0.1 - g[name]).max() / scale
            if err > worst[1]:
                worst = (name, float(err))
        STATS.append((float(lerr), worst[1], bool(shaky)))
        ok = (lerr < 5e-5) and (worst[1] < (5e-3 if shaky else 3e-4))
        return desc, None if ok else f"loss err {lerr:.2e}, worst gradient {worst}, ties {bool(shaky)}"
    finally:
        eng.close()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    fails = refused = 0
    for i in range(n):
        try:
            desc, msg = one_case(rng, i)
        except Exception:
            desc, msg = {"idx": i}, "EXCEPTION " + traceback.format_exc()[-500:]
        if msg and msg.startswith("refused"):
            refused += 1
            print("refused", desc, msg)
        elif msg:
            fails += 1
            print("FAIL", desc, msg)
    if STATS:
        a = np.array(STATS, dtype=float)
        clean = a[a[:, 2] == 0]
        print(f"checked {len(a)} cases ({len(clean)} free of ties): max loss err {a[:, 0].max():.2e}, "
              f"max gradient err on tie-free cases {clean[:, 1].max() if len(clean) else float('nan'):.2e}, overall {a[:, 1].max():.2e}")
    print(f"cases {n}, refused {refused}, failed {fails}")
    return fails


if __name__ == "__main__":
    sys.exit(min(main(), 100))
