"""GPU tests beyond the small parity cases: committed golden fixtures, the ``Smoe`` facade on
the real engine, and size-independent properties at BASELINE.json's full sizes (block
independence / partition invariance, run-to-run determinism, idempotent evaluation,
statistical PSNR parity of a full 200-iteration fit)."""
import os

import numpy as np
import pytest
import torch

from fake_engine import OracleEngine
from oracle import c_oracle as co
from oracle import smoe_oracle as o
from steered_mixture_of_experts_amd import blocks as blk

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _engine(shape, C, K, **kw):
    from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig
    return BlockEngine(EngineConfig(block_shape=shape, channels=C, kernels=K, **kw))


def _dev(p):
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda() for k, v in p.items()}


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(GOLD) if f.endswith(".npz") and not f.startswith("ref_")))
def test_golden_fixtures(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    shape = tuple(int(x) for x in g["block_shape"])
    C, K, yuv, n = int(g["channels"]), int(g["kernels"]), bool(g["use_yuv"]), int(g["n_iters"])
    B = g["target"].shape[0]
    p = {k: g["p_" + k] for k in o.PARAM_NAMES}
    T = torch.from_numpy(np.ascontiguousarray(g["target"].transpose(0, 2, 1))).cuda()
    for tiling in (16, 64):
        eng = _engine(shape, C, K, use_yuv=yuv)
        eng.set_tiling(tiling)
        dp = _dev(p)
        act = torch.full((B,), (1 << K) - 1, dtype=torch.int32, device="cuda")
        out = eng.forward(T, dp, act, update_active=False)
        recon = out["recon"].cpu().numpy().transpose(0, 2, 1)
        frac = (np.clip(g["y"], 0, 1) * 255 + 0.5) % 1.0
        tie = (frac < 2e-4) | (frac > 1 - 2e-4)
        assert (np.abs(recon - g["recon"])[~tie] < 1e-7).all()
        if not tie.any():
            assert np.allclose(out["loss"].cpu().numpy(), g["loss"], rtol=2e-5)
            assert np.allclose(out["sse"].cpu().numpy(), g["sse"], rtol=2e-5)
            # gradients: m after one Adam step is 0.1 * g
            st = eng.new_adam_state(dp)
            eng.fit(T, dp, st, act, 1)
            for k in o.PARAM_NAMES:
                scale = np.abs(g["g_" + k]).max() + 1e-30
                assert np.abs(st.m[k].cpu().numpy() / 0.1 - g["g_" + k]).max() / scale < 2e-5, k
        # n iterations: the well-conditioned parameters follow the fp64 master closely
        dp = _dev(p)
        st = eng.new_adam_state(dp)
        eng.forward(T, dp, act, want_recon=False)
        eng.fit(T, dp, st, act, n)
        for k in ("nu_e", "musX", "pis"):
            assert np.abs(dp[k].cpu().numpy() - g["fit_" + k]).max() < 1e-3, k
        assert np.median(np.abs(dp["A_diagonal"].cpu().numpy() - g["fit_A_diagonal"])) < 1e-2
        eng.close()


def test_facade_on_gpu_matches_facade_on_the_c_oracle():
    from steered_mixture_of_experts_amd.smoe import Adam, Smoe
    b = blk.synthetic_blocks(48, (16, 16), 1, 4321)
    img = blk.blocks_to_image(b, (96, 128), (16, 16))

    def run(factory):
        s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, engine_factory=factory)
        s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1.0))
        s.train(10, val_iter=5)
        return s

    g, c = run(None), run(OracleEngine)
    assert [i for i, _ in g.get_losses()] == [0, 5, 10]
    assert abs(g.get_losses()[0][1] - c.get_losses()[0][1]) < 1e-7
    assert np.allclose([v for _, v in g.get_losses()], [v for _, v in c.get_losses()], rtol=0.05)
    pg, pc = g.get_params(), c.get_params()
    for k in ("nu_e", "musX", "pis", "gamma_e"):
        assert np.median(np.abs(pg[k] - pc[k])) < 1e-5, k
    assert g.get_reconstruction().shape == img.shape
    d = np.abs(g.get_reconstruction() - c.get_reconstruction())
    assert (d < 1.5 / 255).mean() > 0.98
    assert abs(g.get_psnr() - c.get_psnr()) < 0.3
    assert g.get_num_pis() == c.get_num_pis()


def _cfg2_oracle_run(p0, T, cfg, coords, B, K, n_rounds=2):
    """CPU restatement (plain C, fp32) of the cfg2 schedule: eval, then n_rounds x (100 iterations, readmit, eval)."""
    pc = {k: v.copy() for k, v in p0.items()}
    m = {k: np.zeros_like(v) for k, v in pc.items()}
    v = {k: np.zeros_like(v) for k, v in pc.items()}
    bits = np.full(B, 15, np.uint32)
    f0 = co.forward(cfg, coords, T, pc, bits, want_recon=False, threads=8)
    bp = np.array([cfg.beta1, cfg.beta2], np.float32)
    div = np.zeros(B, np.uint32)
    fc = f0
    for _ in range(n_rounds):
        co.fit(cfg, coords, T, pc, m, v, bits, 100, bp, diverged=div, loss0=f0["loss"], threads=8)
        mask = ((bits[:, None] >> np.arange(K, dtype=np.uint32)) & 1).astype(bool)
        mask = o.readmit(pc, mask, cfg, np.float32)
        bits[:] = (mask.astype(np.uint32) << np.arange(K, dtype=np.uint32)).sum(axis=1)
        fc = co.forward(cfg, coords, T, pc, bits, want_recon=False, threads=8)
    fc["params"] = pc
    return f0, fc, div


def _cfg2_gpu_run(p0, T, shape, C, K, B, n_rounds=2, **kw):
    eng = _engine(shape, C, K, **kw)
    dp = _dev(p0)
    st = eng.new_adam_state(dp)
    Td = torch.from_numpy(T).cuda()
    act = torch.full((B,), 15, dtype=torch.int32, device="cuda")
    dv = torch.zeros((B,), dtype=torch.int32, device="cuda")
    g0 = eng.forward(Td, dp, act, want_recon=False)
    fg = g0
    for _ in range(n_rounds):
        eng.fit(Td, dp, st, act, 100, diverged=dv, loss0=g0["loss"])
        eng.update_kernel_list(dp, act)
        fg = eng.forward(Td, dp, act, want_recon=False)
    out = (g0["loss"].cpu().numpy(), g0["sse"].cpu().numpy(), fg["sse"].cpu().numpy(), int(dv.sum()),
           {k: v.cpu().numpy() for k, v in dp.items()})
    eng.close()
    return out


def test_cfg2_full_fit_psnr_parity():
    """BASELINE configs[1]: 512x512 grayscale, 16x16 blocks, K=4, 200 Adam iterations.

    (a) Well-conditioned run (steering lr = base_lr * 10 instead of the CLI's * 1000): the fit is a deterministic
        descent and the contract bound holds -- median block PSNR and aggregate PSNR within 0.05 dB of the CPU
        restatement.
    (b) CLI-default hyper-parameters: individual trajectories are chaotic.  The statistic's own
        noise floor is measured here -- the restatement rerun with its parameters perturbed by 1e-7 relative moves
        the median block PSNR by sigma ~ 0.09 dB (up to 0.15 dB) -- and the GPU has to sit inside that band."""
    B, shape, C, kpd, K = 1024, (16, 16), 1, [2, 2], 4
    b = blk.synthetic_blocks(B, shape, C, 20260002)
    T = blk.to_planar(b)
    p0 = blk.init_block_params(b, kpd)
    coords = np.ascontiguousarray(o.block_coords(shape).T)
    ps = lambda s: -10 * np.log10(np.maximum(s, 1e-12) / 256)
    agg = lambda s: -10 * np.log10(s.sum() / (B * 256))

    # (a) gentle steering step: tight parity
    cfg_a = o.OracleConfig(block_shape=shape, channels=C, kernels=K, lr_steer=1e-2)
    f0, fc, div = _cfg2_oracle_run(p0, T, cfg_a, coords, B, K)
    l0, sse0, sse_g, ndiv, pg = _cfg2_gpu_run(p0, T, shape, C, K, B, lr_steer=1e-2)
    rel0 = np.abs(l0 - f0["loss"]) / f0["loss"]
    assert np.quantile(rel0, 0.9) < 2e-5 and rel0.max() < 2e-2          # blocks with a quantiser tie differ by one LSB
    assert abs(np.median(ps(sse_g)) - np.median(ps(fc["sse"]))) < 0.05, (np.median(ps(sse_g)), np.median(ps(fc["sse"])))
    assert abs(agg(sse_g) - agg(fc["sse"])) < 0.05, (agg(sse_g), agg(fc["sse"]))
    assert np.median(ps(sse_g)) > np.median(ps(sse0)) + 5.0              # the fit actually fits
    assert ndiv == int(div.sum()) == 0
    # parameters after the 200 Adam steps (<= 1e-3 relative on nu, Gamma, mu, pi; <= 1e-2 on A), judged per
    # tensor on its own scale; the bulk of the 1024 blocks is far tighter, a few blocks carry a quantiser-tie history
    for name, tol in (("nu_e", 1e-3), ("gamma_e", 1e-3), ("musX", 1e-3), ("pis", 1e-3), ("A_diagonal", 1e-2), ("A_corr", 1e-2)):
        ref = fc["params"][name]
        dev = np.abs(pg[name] - ref).reshape(B, -1).max(axis=1) / (np.abs(ref).max() + 1e-30)
        assert np.quantile(dev, 0.95) <= tol, (name, np.quantile(dev, 0.95))
        assert np.median(dev) <= tol / 5, (name, np.median(dev))

    # (b) CLI defaults: inside the restatement's own perturbation band
    cfg_b = o.OracleConfig(block_shape=shape, channels=C, kernels=K)
    _, fc, div = _cfg2_oracle_run(p0, T, cfg_b, coords, B, K)
    rng = np.random.default_rng(0)
    meds = []
    for _ in range(4):
        pp = {k: (v * (1 + rng.normal(size=v.shape).astype(np.float32) * 1e-7)).astype(np.float32) for k, v in p0.items()}
        meds.append(np.median(ps(_cfg2_oracle_run(pp, T, cfg_b, coords, B, K)[1]["sse"])))
    base = np.median(ps(fc["sse"]))
    band = max(3 * float(np.std(meds + [base])), float(np.max(np.abs(np.array(meds) - base))), 0.05)
    _, sse0, sse_g, ndiv, _ = _cfg2_gpu_run(p0, T, shape, C, K, B)
    assert abs(np.median(ps(sse_g)) - base) <= band, (np.median(ps(sse_g)), base, band)
    assert band < 0.6                                                    # the band itself is a fraction of a dB
    assert np.median(ps(sse_g)) > np.median(ps(sse0)) + 5.0
    assert abs(agg(sse_g) - agg(fc["sse"])) < 1.5            # a few blown-up blocks dominate: fp32-vs-fp64 spread is 0.3-2.5 dB
    assert ndiv == int(div.sum()) == 0


def test_block_independence_determinism_idempotence_at_full_size():
    """B = 65536 (the bench batch): a block's result does not depend on its neighbours or on
    where it sits in the batch; two runs are bit-identical; evaluation is idempotent."""
    B, shape, C, kpd, K = 65536, (16, 16), 1, [2, 2], 4
    b = blk.synthetic_blocks(B, shape, C, 20260002)
    T = torch.from_numpy(blk.to_planar(b)).cuda()
    p0 = blk.init_block_params(b, kpd)
    eng = _engine(shape, C, K)
    eng.set_tiling(16)          # one reduction order for every batch size (the default picks it from B)

    def run(Tsub, psub, n):
        dp = _dev(psub)
        st = eng.new_adam_state(dp)
        act = torch.full((Tsub.shape[0],), 15, dtype=torch.int32, device="cuda")
        eng.forward(Tsub, dp, act, want_recon=False)
        eng.fit(Tsub, dp, st, act, n)
        out = eng.forward(Tsub, dp, act, want_recon=True, update_active=False)
        out2 = eng.forward(Tsub, dp, act, want_recon=True, update_active=False)
        assert torch.equal(out["recon"], out2["recon"]) and torch.equal(out["loss"], out2["loss"])
        return {k: v.cpu().numpy() for k, v in dp.items()}, out["recon"].cpu().numpy(), act.cpu().numpy()

    full, rec_full, act_full = run(T, p0, 12)
    again, rec_again, _ = run(T, p0, 12)
    for k in full:
        assert np.array_equal(full[k], again[k]), k
    assert np.array_equal(rec_full, rec_again)
    lo, hi = 40000, 40150              # a ragged slice: different workgroup alignment, tail workgroup
    sub, rec_sub, act_sub = run(T[lo:hi].contiguous(), {k: v[lo:hi] for k, v in p0.items()}, 12)
    for k in full:
        assert np.array_equal(full[k][lo:hi], sub[k]), k
    assert np.array_equal(rec_full[lo:hi], rec_sub) and np.array_equal(act_full[lo:hi], act_sub)
    assert np.isfinite(full["nu_e"]).all()
    eng.close()


@pytest.mark.parametrize("kw", [{"ssim_opt": True}, {"quantization_mode": 3, "quantize_pis": True, "bit_depths": (14, 12, 8, 10, 10)},
                                {"quantization_mode": 2, "quantize_pis": True, "bit_depths": (16, 14, 8, 10, 10),
                                 "lower_bounds": (-200, -.3, -1, 0, -4), "upper_bounds": (200, 1.3, 2, 2, 4)},
                                {"train_inverse_cov": True}, {"radial_as": True}],
                         ids=["ssim", "quant3", "quant2", "invcov", "radial"])
@pytest.mark.parametrize("tiling", [16, 64])
def test_variant_kernels_keep_blocks_independent_at_full_size(kw, tiling):
    """The SSIM planes, the quantised LDS image with its block-wide all-reduces, the inverse-covariance and radial
    variants: a block's fit inside the 65 536-block bench batch is bit-identical to the same block fitted in a ragged
    150-block slice, and two runs agree bit for bit."""
    B, shape, C, kpd, K = 65536, (16, 16), 1, [2, 2], 4
    b = blk.synthetic_blocks(B, shape, C, 20260002)
    T = torch.from_numpy(blk.to_planar(b)).cuda()
    p0 = blk.init_block_params(b, kpd, True, kw.get("train_inverse_cov", False))
    eng = _engine(shape, C, K, lr_steer=0.05, **kw)
    eng.set_tiling(tiling)

    def run(Tsub, psub, n):
        dp = _dev(psub)
        st = eng.new_adam_state(dp)
        act = torch.full((Tsub.shape[0],), 15, dtype=torch.int32, device="cuda")
        eng.forward(Tsub, dp, act, want_recon=False)
        eng.fit(Tsub, dp, st, act, n)
        eng.update_kernel_list(dp, act)
        out = eng.forward(Tsub, dp, act, want_recon=True, update_active=False)
        return {k: v.cpu().numpy() for k, v in dp.items()}, out["recon"].cpu().numpy(), out["loss"].cpu().numpy()

    full, rec_full, loss_full = run(T, p0, 8)
    again, rec_again, _ = run(T, p0, 8)
    for k in full:
        assert np.array_equal(full[k], again[k]), k
    lo, hi = 40000, 40150
    sub, rec_sub, loss_sub = run(T[lo:hi].contiguous(), {k: v[lo:hi] for k, v in p0.items()}, 8)
    for k in full:
        assert np.array_equal(full[k][lo:hi], sub[k]), k
    assert np.array_equal(rec_full[lo:hi], rec_sub) and np.array_equal(loss_full[lo:hi], loss_sub)
    assert np.isfinite(full["nu_e"]).all() and np.isfinite(loss_full).all()
    eng.close()


@pytest.mark.parametrize("name,img_shape,bs,C,kpd,n_iters", [
    ("cfg3", (1080, 1920), (32, 32), 3, [2, 4], 3),
    ("cfg4", (2160, 3840), (16, 16), 3, [2, 2], 3),
    ("cfg5", (1080, 1920, 30), (16, 16, 4), 3, [2, 2, 1], 2),
])
def test_large_configs_against_the_c_oracle_on_a_sample(name, img_shape, bs, C, kpd, n_iters):
    """BASELINE configs[2..4] at full size on one GPU (synthetic blocks of the padded grid):
    a sample of blocks is re-computed by the C restatement and must agree."""
    K = int(np.prod(kpd))
    grid = blk.grid_shape(img_shape, bs)
    B = int(np.prod(grid))
    assert B == {"cfg3": 2040, "cfg4": 32400, "cfg5": 65280}[name]
    N = int(np.prod(bs))
    b = blk.synthetic_blocks(B, bs, C, 20260000 + len(name) + C + K)
    T = blk.to_planar(b)
    p0 = blk.init_block_params(b, kpd)
    eng = _engine(bs, C, K, use_yuv=True)
    dp = _dev(p0)
    st = eng.new_adam_state(dp)
    Td = torch.from_numpy(T).cuda()
    act = torch.full((B,), (1 << K) - 1, dtype=torch.int32, device="cuda")
    g0 = eng.forward(Td, dp, act, want_recon=False)
    eng.fit(Td, dp, st, act, n_iters)
    g1 = eng.forward(Td, dp, act, want_recon=False, update_active=False)
    torch.cuda.synchronize()
    idx = np.linspace(0, B - 1, 64).astype(int)
    cfg = o.OracleConfig(block_shape=bs, channels=C, kernels=K, use_yuv=True)
    coords = np.ascontiguousarray(o.block_coords(bs).T)
    pc = {k: np.ascontiguousarray(v[idx]) for k, v in p0.items()}
    m = {k: np.zeros_like(v) for k, v in pc.items()}
    v = {k: np.zeros_like(v) for k, v in pc.items()}
    bits = np.full(len(idx), (1 << K) - 1, np.uint32)
    Ts = np.ascontiguousarray(T[idx])
    c0 = co.forward(cfg, coords, Ts, pc, bits, want_recon=False, threads=8)
    rel = np.abs(g0["loss"].cpu().numpy()[idx] - c0["loss"]) / c0["loss"]
    assert np.median(rel) < 3e-5 and rel.max() < 2e-2                   # quantiser ties flip single LSBs
    rel = np.abs(g0["sse"].cpu().numpy()[idx] - c0["sse"]) / c0["sse"]
    assert np.median(rel) < 3e-5 and rel.max() < 2e-2
    bp = np.array([cfg.beta1, cfg.beta2], np.float32)
    co.fit(cfg, coords, Ts, pc, m, v, bits, n_iters, bp, threads=8)
    c1 = co.forward(cfg, coords, Ts, pc, bits, want_recon=False, update_active=False, threads=8)
    for k in ("nu_e", "musX", "pis"):
        assert np.median(np.abs(dp[k].cpu().numpy()[idx] - pc[k])) < 1e-5, k
    # a few quantisation-tie flips per block are expected after n steps; the loss stays close
    rel = np.abs(g1["loss"].cpu().numpy()[idx] - c1["loss"]) / c1["loss"]
    assert np.median(rel) < 2e-2
    assert g1["loss"].mean().item() < g0["loss"].mean().item()
    eng.close()


def test_smoke_entry_point():
    import __graft_entry__ as g
    g.smoke()


def test_end_to_end_fit_quality_on_a_smooth_image():
    """Functional check through the facades: with a gentle steering learning rate (lr_mult = 10 instead
    of the reference CLI's 1000, which random-walks the steering diagonals through zero on small
    domains -- DESIGN.md section 5) both modes converge and gain several dB."""
    from steered_mixture_of_experts_amd.smoe import Adam, SharedSmoe, Smoe
    rng = np.random.default_rng(7)
    H = W = 128
    yy, xx = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
    img = 0.35 + 0.3 * xx - 0.15 * yy
    for _ in range(8):
        cy, cx, s, a = rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0.05, 0.25), rng.uniform(-0.3, 0.3)
        img += a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
    img += 0.2 * (xx + 0.5 * yy > 0.9) + rng.normal(scale=1.5 / 255, size=img.shape)
    img = (np.round(np.clip(img, 0, 1) * 255).astype(np.uint8).astype(np.float32) / np.float32(255.))[..., None]
    b = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True)
    b.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1e-2))
    p0 = b.get_psnr()
    b.train(300, val_iter=100)
    assert b.get_psnr() > p0 + 4.0, (p0, b.get_psnr())
    assert [i for i, _ in b.get_losses()] == [0, 100, 200, 300]
    assert b.get_losses()[-1][1] < b.get_losses()[0][1]
    s = SharedSmoe(img, train_inverse_cov=False, kernels_per_dim=[8, 8], batch_size=[32, 32], use_determinant=True)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1e-2))
    q0 = s.get_psnr()
    s.train(300, val_iter=100)
    assert s.get_psnr() > q0 + 4.0, (q0, s.get_psnr())
    assert np.diagonal(s.get_params()["A_diagonal"], axis1=-2, axis2=-1).min() > 0


def test_bench_line_keeps_the_driver_contract():
    """python bench.py prints ONE JSON line with the keys the driver reads, the roofline and cpu_baseline objects, and
    times exactly the requested steps."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "30", "--warmup", "10",
                          "--blocks", "4096", "--cpu-budget-s", "1.5", "--clock-warm-iters", "100"],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 30 and d["warmup"] == 10 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert abs(d["value"] - 4096 * 256 * 30 / (d["ms_per_step"] * 30 / 1e3) / 1e6) / d["value"] < 0.02


def test_facade_pads_kernel_counts_without_their_own_kernel():
    """kernels_per_dim = [1, 5]: no (2, 1, 5) instantiation -> the facade runs K = 6 with one prior-zero kernel
    (smoe.py:480,738 drops it); results follow the restatement at K = 5."""
    from fake_engine import OracleEngine
    from steered_mixture_of_experts_amd.smoe import Adam, Smoe
    b = blk.synthetic_blocks(12, (16, 16), 1, 77)
    img = blk.blocks_to_image(b, (48, 64), (16, 16))

    def run(factory):
        s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[1, 5], batch_size=[16, 16], use_determinant=True,
                 quantize_pis=True, engine_factory=factory)
        s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1e-2))
        s.train(8, val_iter=4)
        return s
    g, c = run(None), run(OracleEngine)
    assert g._kp == 6 and g.kernels == 5 and c._kp == 5
    pg, pc = g.get_params(), c.get_params()
    for k in pg:
        assert pg[k].shape == pc[k].shape
        assert np.abs(pg[k] - pc[k]).max() < 2e-3 * (np.abs(pc[k]).max() + 1e-6), k
    assert np.allclose(g.get_losses(), c.get_losses(), rtol=1e-3)
    assert (np.abs(g.get_reconstruction() - c.get_reconstruction()) < 1.5 / 255).mean() > 0.999
    assert g.get_weight_matrix().shape == c.get_weight_matrix().shape


def test_pixel_sub_sampling_on_the_gpu():
    from steered_mixture_of_experts_amd.smoe import Adam, Smoe
    b = blk.synthetic_blocks(16, (16, 16), 1, 78)
    img = blk.blocks_to_image(b, (64, 64), (16, 16))
    s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1e-2))
    l0 = s.run_batched(train=False, update_reconstruction=True)[0]
    w = s._sample_pixels(25)
    assert ((w > 0).sum(dim=1) == 64).all() and torch.allclose(w[w > 0], torch.tensor(4.0, device=w.device))
    s.train(20, val_iter=10, sampling_percentage=50)
    assert s.get_iter() == 20 and s.get_losses()[-1][1] < l0
