"""Randomised GPU-vs-oracle sweep over block shapes and option combinations: one evaluation pass and one fit step per
case, the tolerances of tests/test_gpu_parity.py.  Prints every failing case; exit code = number of failures."""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import smoe_oracle as o                                                             # noqa: E402
from test_gpu_parity import _engine, _mask_to_bits, _planar, _setup, _to_dev, _to_host          # noqa: E402

STATS = []
QKW = dict(bit_depths=(14, 12, 8, 10, 10), lower_bounds=(-60, -.3, -1, 0, -4), upper_bounds=(60, 1.3, 2, 2, 4))


def one_case(rng, idx):
    d = 3 if rng.random() < float(os.environ.get('FUZZ_P3D', '0.2')) else 2
    C = 3 if ((d == 3 and rng.random() < 0.7) or (d == 2 and rng.random() < 0.5)) else 1
    if d == 3:
        shape = (int(rng.integers(5, 17)), int(rng.integers(5, 17)), int(rng.integers(2, 6)))
        kpd = [2, 2, 1]
    else:
        shape = (int(rng.integers(5, 41)), int(rng.integers(5, 41)))
        kpd = [2, 4] if (C == 3 and rng.random() < 0.3) or (C == 1 and rng.random() < 0.3) else [2, 2]
    generic = rng.random() < float(os.environ.get('FUZZ_PGENERIC', '0.35'))
    if generic or (d == 3 and C == 1):
        # any kernel grid (csrc/smoe_variants.def, basic instantiations: margin loss, quantize_pis, train_inverse_cov)
        grids = {(2, 1): [[1, 1], [1, 2], [3, 1], [2, 3], [3, 3], [3, 4], [4, 4]],
                 (2, 3): [[1, 1], [2, 1], [1, 3], [3, 2], [3, 3]],
                 (3, 1): [[1, 1, 1], [2, 1, 1], [1, 3, 1], [2, 2, 1], [3, 2, 1], [2, 2, 2]],
                 (3, 3): [[1, 1, 1], [1, 2, 1], [3, 1, 1], [1, 3, 2], [2, 2, 2]]}[(d, C)]
        kpd = grids[int(rng.integers(len(grids)))]
        generic = True
    yuv = bool(C == 3 and rng.random() < 0.7)
    kw = {}
    if rng.random() < 0.4:
        kw["train_inverse_cov"] = True
    mode = 0 if generic else int(rng.choice([0, 0, 2, 3]))
    if mode:
        kw.update(quantization_mode=mode, quantize_pis=True, **QKW)
    elif rng.random() < 0.5:
        kw["quantize_pis"] = True
    ssim = not generic and rng.random() < 0.3 and (d == 2 or (C == 3 and min(shape) >= 5))     # 3-d: the FULL triple (3, 3, 4)
    if ssim:
        kw["ssim_opt"] = True
    if C == 3 and yuv and rng.random() < 0.3:
        kw["only_y_gamma"] = True
    if rng.random() < 0.3:
        kw.update(pis_l1=0.05, u_l1=0.002)
    if rng.random() < 0.2 and not mode:
        kw["kernel_count_as_norm_l1"] = True
        kw.setdefault("pis_l1", 0.05)
    if rng.random() < 0.15:
        kw["radial_as"] = True
    centred = mode >= 2 and rng.random() < 0.3              # use_diff_center: the quantised variable is musX - grid
    B = int(rng.integers(3, 40))
    tiling = int(rng.choice([0, 16, 32, 64, 128]))
    if os.environ.get('FUZZ_TILING'):
        tiling = int(os.environ['FUZZ_TILING'])
    desc = dict(idx=idx, shape=shape, C=C, kpd=kpd, yuv=yuv, B=B, tiling=tiling, **kw)
    cfg, p, coords, tgt, K = _setup(shape, C, kpd, yuv, B, 1000 + idx, **kw)
    dd = len(shape)
    if mode == 3 or kw.get("radial_as"):
        p["A_corr"] = p["A_corr"] * np.tril(np.ones((dd, dd), np.float32), -1)
    if kw.get("radial_as"):
        a = np.abs(p["A_diagonal"][:, :, 0, 0])
        p["A_diagonal"] = (a[..., None, None] * np.eye(dd)).astype(np.float32)
        p["A_corr"] = np.zeros_like(p["A_corr"])
    if kw.get("train_inverse_cov"):
        p["A_diagonal"] = (p["A_diagonal"] ** 2).astype(np.float32)
        p["A_corr"] = (p["A_corr"] * 0.3).astype(np.float32)
    grid = None
    if centred:
        grid = o.init_params(tgt.reshape((B,) + tuple(shape) + (C,)), kpd)["musX"].astype(np.float32)
        p["musX"] = (grid + rng.uniform(-0.05, 0.05, size=grid.shape)).astype(np.float32)
        cfg = o.OracleConfig(**{**cfg.__dict__, "mus_grid": grid})
        desc["centred"] = True
    active = np.ones((B, K), bool)
    try:
        eng = _engine(shape, C, K, use_yuv=yuv, **kw)
    except Exception as e:                                   # unsupported shape / combination: must say so
        return desc, "refused: " + str(e)[:80]
    try:
        if tiling:
            try:
                eng.set_tiling(tiling)
                if not eng.fit_variant(B):
                    eng.set_tiling(0)
            except Exception:
                eng.set_tiling(0)
        dp = _to_dev(p)
        gdev = None
        if grid is not None:
            gdev = torch.from_numpy(grid).cuda()
            eng.set_center_grid(gdev)
        act = torch.from_numpy(_mask_to_bits(active).view(np.int32)).cuda()
        T = _planar(tgt)
        fw = eng.forward(T, dp, act, want_recon=True, update_active=False)
        recon = np.transpose(fw["recon"].cpu().numpy(), (0, 2, 1))
        if mode:
            q32, back, _ = o.quantize_graph_params(p, cfg, np.float32)
            cfg0 = o.OracleConfig(**{**cfg.__dict__, "quantization_mode": 0, "quantize_pis": False})
            ref = o.forward(q32, tgt, coords, active, cfg0, None, np.float64, want_grads=True, q_override=recon)
            ref["grads"] = o.route_quant_grads(ref["grads"], back, np.float64)
        else:
            ref = o.forward(p, tgt, coords, active, cfg, None, np.float64, want_grads=True, q_override=recon)
        frac = (np.clip(ref["y"], 0, 1) * 255 + 0.5) % 1.0
        tie_px = (frac < 3e-4) | (frac > 1 - 3e-4)
        lerr = np.abs(fw["loss"].cpu().numpy() - ref["loss"]).max()
        st = eng.new_adam_state(dp)
        eng.fit(T, dp, st, act, 1)
        torch.cuda.synchronize()
        tie = (np.abs(ref["w"] - 0.5 / 256) < 1e-6).any(axis=(1, 2))
        edge = ((np.abs(ref["y"]) < 1e-6) | (np.abs(ref["y"] - 1) < 1e-6)).any(axis=(1, 2))
        # the fit kernel hoists lane-constant terms, the evaluation kernel does not: their y can differ in the last bits, so a
        # pixel whose y*255 sits on a rounding tie may quantise one level apart in the two kernels -- such blocks are skipped
        clean = ~(tie | edge | tie_px.any(axis=(1, 2)))
        m = _to_host(st.m)
        worst = ("", 0.0)
        if clean.any():
            for name in o.PARAM_NAMES:
                g = ref["grads"][name][clean]
                # (one kernel per block: the gradients of pis / musX / A vanish identically; rounding residue vs the expert scale)
                floor = np.abs(ref["grads"]["nu_e"][clean]).max() if K == 1 else 0.0
                err = np.abs(m[name][clean] / 0.1 - g).max() / (max(np.abs(g).max(), floor) + 1e-30)
                if err > worst[1]:
                    worst = (name, float(err))
        ok = (lerr < 5e-5) and (worst[1] < 2e-4)
        STATS.append((float(lerr), worst[1], int(clean.sum()), B))
        note = ""
        if not ok and worst[0] and not mode:
            # how far the SAME arithmetic in fp32 on the CPU is from fp64 on this case: a conditioning problem shows here too
            r32 = o.forward(p, tgt, coords, active, cfg, None, np.float32, want_grads=True, q_override=recon)
            g64, g32 = ref["grads"][worst[0]][clean], r32["grads"][worst[0]][clean]
            dev = np.abs(m[worst[0]][clean] / 0.1 - g64)
            where = np.unravel_index(np.argmax(dev), dev.shape)
            vs32 = np.abs(m[worst[0]][clean] / 0.1 - g32).max() / (np.abs(g64).max() + 1e-30)
            note = (f"; fp32 restatement vs fp64: {np.abs(g32 - g64).max() / (np.abs(g64).max() + 1e-30):.2e}; GPU vs fp32 restatement: "
                    f"{vs32:.2e}; worst element {where}: gpu "
                    f"{(m[worst[0]][clean] / 0.1)[where]:.6g} fp64 {g64[where]:.6g} fp32 {g32[where]:.6g}")
            if lerr < 5e-5 and vs32 < 2e-5:
                # ill-conditioned in fp32 (the fp32 restatement is as far from fp64 as the GPU, and the GPU sits on it): the
                # reference computes in fp32 too -- reported, not a failure
                print("fp32-conditioned", desc, note)
                return desc, None
        return desc, None if ok else f"loss err {lerr:.2e}, worst gradient {worst}{note}"
    finally:
        eng.close()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    fails = refused = 0
    for i in range(n):
        try:
            desc, msg = one_case(rng, i)
        except Exception:
            desc, msg = {"idx": i}, "EXCEPTION " + traceback.format_exc()[-400:]
        if msg and msg.startswith("refused"):
            refused += 1
            print("refused", desc, msg)
        elif msg:
            fails += 1
            print("FAIL", desc, msg)
    if STATS:
        a = np.array(STATS)
        print(f"checked {len(a)} cases: max loss err {a[:, 0].max():.2e}, max gradient err {a[:, 1].max():.2e}, "
              f"clean blocks {int(a[:, 2].sum())} of {int(a[:, 3].sum())}")
    print(f"cases {n}, refused {refused}, failed {fails}")
    return fails


if __name__ == "__main__":
    sys.exit(min(main(), 100))
