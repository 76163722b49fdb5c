"""GPU parity of the SSIM loss (ssim_opt): loss_pixel = 1 - SSIM with custom_ssim on
SYMMETRIC-padded blocks (smoe.py:929,980-1011; ops/image_ops_impl.py:77-233) against the CPU restatement
(oracle.ssim_and_grad, itself checked against torch.autograd in tests/test_oracle.py).  The GPU evaluates
the window sums as Tr * plane * Tc with per-axis tap matrices; the oracle pads and correlates literally."""
import numpy as np
import pytest
import torch

from oracle import smoe_oracle as o
from test_gpu_parity import _close, _engine, _mask_to_bits, _planar, _setup, _to_dev, _to_host

pytestmark = pytest.mark.gpu

SSIM_SHAPES = [
    # block_shape, C, kernels_per_dim, use_yuv
    ((16, 16), 1, [2, 2], False),
    ((16, 16), 3, [2, 2], True),
    ((16, 12), 3, [2, 2], False),      # 64 lanes are not a multiple of the last axis: un-hoisted kernel, ragged sweep
    ((8, 8), 1, [2, 2], False),        # block smaller than the 11-tap window: every window wraps both borders
    ((32, 32), 3, [2, 4], True),
    ((16, 16), 1, [2, 4], False),
    # 3-d blocks (smoe.py:999-1003): the 11x11x11 window, three padded axes
    ((8, 8, 6), 3, [2, 2, 1], True),
    ((6, 7, 5), 3, [2, 2, 1], False),  # every axis shorter than the window, nothing a multiple of anything
    ((8, 8, 8), 3, [2, 1, 2], True),
]


def _tilings(shape):
    # 16x16 blocks: register / DPP stage on 16 lanes per block (default) and the LDS stage on a whole wavefront
    return (16, 64) if tuple(shape) == (16, 16) else (0,)


@pytest.mark.parametrize("shape,C,kpd,yuv", SSIM_SHAPES)
def test_ssim_forward_loss(shape, C, kpd, yuv):
    for tiling in _tilings(shape):
        _forward_loss(shape, C, kpd, yuv, tiling)


def _forward_loss(shape, C, kpd, yuv, tiling):
    B = 19
    cfg, p, coords, tgt, K = _setup(shape, C, kpd, yuv, B, 40 + C, pis_l1=0.2, u_l1=0.003, ssim_opt=True)
    active = np.ones((B, K), bool)
    eng = _engine(shape, C, K, use_yuv=yuv, pis_l1=0.2, u_l1=0.003, ssim_opt=True)
    plain = _engine(shape, C, K, use_yuv=yuv, pis_l1=0.2, u_l1=0.003)
    assert ("g16" if tuple(shape) == (16, 16) else "g64") in eng.fit_variant(B)
    if tiling:
        eng.set_tiling(tiling)
        plain.set_tiling(tiling)
        assert f"g{tiling}" in eng.fit_variant(B)
    dp = _to_dev(p)
    act = torch.from_numpy(_mask_to_bits(active).view(np.int32)).cuda()
    T = _planar(tgt)
    out = eng.forward(T, dp, act, want_recon=True, update_active=False)
    ref_plain = plain.forward(T, dp, act, want_recon=True, update_active=False)
    torch.cuda.synchronize()
    assert torch.equal(out["recon"], ref_plain["recon"]) and torch.equal(out["sse"], ref_plain["sse"])
    recon = np.transpose(out["recon"].cpu().numpy(), (0, 2, 1))
    ref = o.forward(p, tgt, coords, active, cfg, None, np.float64, q_override=recon)
    loss = out["loss"].cpu().numpy()
    # fp32 window statistics (E[x^2] - mu^2 cancels 2-3 digits) against the fp64 restatement
    assert np.abs(loss - ref["loss"]).max() < 2e-5, np.abs(loss - ref["loss"]).max()
    ref32 = o.forward(p, tgt, coords, active, cfg, None, np.float32, q_override=recon)
    assert np.abs(loss - ref["loss"]).max() < 4 * np.abs(ref32["loss"] - ref["loss"]).max() + 2e-6
    assert (loss > 0.01).all()                      # really the SSIM loss, not the (tiny) margin loss
    # a perfect reconstruction has loss_pixel = 0: feed parameters whose blend equals a constant target
    const = np.full_like(tgt, 128 / 255)
    p2 = {k: v.copy() for k, v in p.items()}
    p2["nu_e"][:] = 128 / 255
    p2["gamma_e"][:] = 0
    o2 = eng.forward(_planar(const), _to_dev(p2), act, want_recon=False, update_active=False)
    reg = o.forward(p2, const, coords, active, cfg, None, np.float64)["loss"]
    assert np.abs(o2["loss"].cpu().numpy() - reg).max() < 2e-5     # fp32 E[x^2] - mu^2 against c2 = 9e-4
    eng.close()
    plain.close()


@pytest.mark.parametrize("shape,C,kpd,yuv", SSIM_SHAPES)
def test_ssim_one_step_gradients(shape, C, kpd, yuv):
    for tiling in _tilings(shape):
        _one_step_gradients(shape, C, kpd, yuv, tiling)


def _one_step_gradients(shape, C, kpd, yuv, tiling):
    B = 21
    cfg, p, coords, tgt, K = _setup(shape, C, kpd, yuv, B, 300 + C, ssim_opt=True)
    active = np.ones((B, K), dtype=bool)
    eng = _engine(shape, C, K, use_yuv=yuv, ssim_opt=True)
    if tiling:
        eng.set_tiling(tiling)
    dp = _to_dev(p)
    act = torch.from_numpy(_mask_to_bits(active).view(np.int32)).cuda()
    T = _planar(tgt)
    fw = eng.forward(T, dp, act, want_recon=True, update_active=False)
    recon = np.transpose(fw["recon"].cpu().numpy(), (0, 2, 1))
    ref64 = o.forward(p, tgt, coords, active, cfg, None, np.float64, want_grads=True, q_override=recon)
    state = eng.new_adam_state(dp)
    loss = torch.zeros(B, device="cuda")
    sse = torch.zeros(B, device="cuda")
    eng.fit(T, dp, state, act, 1, loss_out=loss, sse_out=sse)
    torch.cuda.synchronize()
    tie = (np.abs(ref64["w"] - 0.5 / 256) < 1e-6).any(axis=(1, 2))
    edge = ((np.abs(ref64["y"]) < 1e-6) | (np.abs(ref64["y"] - 1) < 1e-6)).any(axis=(1, 2))
    clean = ~(tie | edge)
    assert clean.sum() >= B // 2
    assert np.abs(loss.cpu().numpy() - ref64["loss"])[clean].max() < 2e-5
    assert _close(sse.cpu().numpy()[clean], ref64["sse"][clean], rtol=2e-5).all()
    m = _to_host(state.m)
    for name in o.PARAM_NAMES:
        g_ref = ref64["grads"][name][clean]
        scale = np.abs(g_ref).max() + 1e-30
        err = np.abs(m[name][clean] / 0.1 - g_ref).max() / scale
        assert err < 5e-5, (name, err)          # fp32 window sums: E[x^2] - mu^2 cancels ~3 digits
    eng.close()


def test_ssim_gentle_fit_follows_the_restatement_and_improves_ssim():
    shape, C, kpd = (16, 16), 1, [2, 2]
    B = 48
    cfg, p, coords, tgt, K = _setup(shape, C, kpd, False, B, 991, perturb=False, lr_steer=1e-2, ssim_opt=True)
    n = 30
    p32, _, i32 = o.fit(p, tgt, coords, cfg, n, val_iter=10 ** 9, dtype=np.float32)
    p64, _, _ = o.fit(p, tgt, coords, cfg, n, val_iter=10 ** 9, dtype=np.float64)
    eng = _engine(shape, C, K, lr_steer=1e-2, ssim_opt=True)
    dp = _to_dev(p)
    state = eng.new_adam_state(dp)
    act = torch.full((B,), (1 << K) - 1, dtype=torch.int32, device="cuda")
    T = _planar(tgt)
    l0 = eng.forward(T, dp, act, want_recon=False)["loss"].cpu().numpy()
    eng.fit(T, dp, state, act, n)
    l1 = eng.forward(T, dp, act, want_recon=False, update_active=False)["loss"].cpu().numpy()
    torch.cuda.synchronize()
    assert np.median(l1) < np.median(l0) - 0.02          # 1 - SSIM went down
    got = _to_host(dp)
    for name in o.PARAM_NAMES:
        dev = np.abs(got[name] - p32[name])
        floor = np.abs(p32[name] - p64[name])
        assert np.median(dev) <= 3 * np.median(floor) + 1e-5, (name, np.median(dev), np.median(floor))
    eng.close()


QKW = dict(bit_depths=(14, 12, 8, 10, 10), lower_bounds=(-60, -.3, -1, 0, -4), upper_bounds=(60, 1.3, 2, 2, 4))


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("shape,C,kpd,yuv", [SSIM_SHAPES[0], SSIM_SHAPES[1], SSIM_SHAPES[4], SSIM_SHAPES[6]])
def test_ssim_on_fake_quantised_variables(shape, C, kpd, yuv, mode):
    """ssim_opt together with quantization_mode 2 / 3: the QUANT instantiation of the SSIM kernels (quantised LDS image,
    masked / routed backward) -- loss and one-step gradients against the restatement, both tilings of 16x16 blocks."""
    for tiling in _tilings(shape):
        B = 21
        kw = dict(ssim_opt=True, quantization_mode=mode, quantize_pis=True, **QKW)
        cfg, p, coords, tgt, K = _setup(shape, C, kpd, yuv, B, 500 + C, **kw)
        d = len(shape)
        p["A_corr"] = p["A_corr"] * np.tril(np.ones((d, d), np.float32), -1)
        active = np.ones((B, K), dtype=bool)
        eng = _engine(shape, C, K, use_yuv=yuv, **kw)
        if tiling:
            eng.set_tiling(tiling)
        dp = _to_dev(p)
        act = torch.from_numpy(_mask_to_bits(active).view(np.int32)).cuda()
        T = _planar(tgt)
        fw = eng.forward(T, dp, act, want_recon=True, update_active=False)
        recon = np.transpose(fw["recon"].cpu().numpy(), (0, 2, 1))
        # the parameter lattice in fp32 (what TF and the kernels compute), the SSIM statistic in fp64: graph on the
        # quantised variables, then the fake-quant backward on its gradients
        q32, back, _ = o.quantize_graph_params(p, cfg, np.float32)
        cfg0 = o.OracleConfig(**{**cfg.__dict__, "quantization_mode": 0, "quantize_pis": False})
        ref64 = o.forward(q32, tgt, coords, active, cfg0, None, np.float64, want_grads=True, q_override=recon)
        ref64["grads"] = o.route_quant_grads(ref64["grads"], back, np.float64)
        assert np.abs(fw["loss"].cpu().numpy() - ref64["loss"]).max() < 3e-5
        state = eng.new_adam_state(dp)
        loss = torch.zeros(B, device="cuda")
        eng.fit(T, dp, state, act, 1, loss_out=loss)
        torch.cuda.synchronize()
        tie = (np.abs(ref64["w"] - 0.5 / 256) < 1e-6).any(axis=(1, 2))
        edge = ((np.abs(ref64["y"]) < 1e-6) | (np.abs(ref64["y"] - 1) < 1e-6)).any(axis=(1, 2))
        clean = ~(tie | edge)
        assert clean.sum() >= B // 2
        assert np.abs(loss.cpu().numpy() - ref64["loss"])[clean].max() < 3e-5
        m = _to_host(state.m)
        for name in o.PARAM_NAMES:
            g_ref = ref64["grads"][name][clean]
            scale = np.abs(g_ref).max() + 1e-30
            err = np.abs(m[name][clean] / 0.1 - g_ref).max() / scale
            assert err < 1e-4, (name, tiling, err)
        eng.close()


def test_ssim_unsupported_configurations():
    from steered_mixture_of_experts_amd import _lib
    from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig
    with pytest.raises(_lib.SmoeError) as e:       # a triple with the basic kernel set only (csrc/smoe_variants.def)
        BlockEngine(EngineConfig(block_shape=(8, 8, 6), channels=1, kernels=4, ssim_opt=True))
    assert e.value.code == _lib.SMOE_ERR_UNSUPPORTED
    with pytest.raises(_lib.SmoeError) as e:       # fewer than 5 frames: TF refuses the SYMMETRIC pad
        BlockEngine(EngineConfig(block_shape=(16, 16, 4), channels=3, kernels=4, ssim_opt=True))
    assert e.value.code == _lib.SMOE_ERR_INVALID
    with pytest.raises(_lib.SmoeError) as e:
        BlockEngine(EngineConfig(block_shape=(4, 16), channels=1, kernels=4, ssim_opt=True))
    assert e.value.code == _lib.SMOE_ERR_INVALID


def test_ssim_facade_on_gpu_follows_the_oracle_backed_facade():
    from fake_engine import OracleEngine
    from steered_mixture_of_experts_amd import blocks as blk
    from steered_mixture_of_experts_amd.smoe import Adam, Smoe
    b = blk.synthetic_blocks(16, (16, 16), 3, 21)
    img = blk.blocks_to_image(b, (64, 64), (16, 16))
    runs = []
    for factory in (None, OracleEngine):
        s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, use_yuv=True, ssim_opt=True,
                 **({} if factory is None else {"engine_factory": factory}))
        s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
        s.train(20, val_iter=10)
        runs.append(([v for _, v in s.get_losses()], [v for _, v in s.get_mses()], s.get_params()))
    (lg, mg, pg), (lo, mo, po) = runs
    assert abs(lg[0] - lo[0]) < 2e-5 and abs(mg[0] - mo[0]) < 1e-3 * mo[0]
    assert np.allclose(lg, lo, atol=3e-3) and lg[-1] < lg[0] - 0.01
    assert np.median(np.abs(pg["nu_e"] - po["nu_e"])) < 2e-4


def test_ssim_video_facade_on_gpu_follows_the_oracle_backed_facade():
    from fake_engine import OracleEngine
    from steered_mixture_of_experts_amd.smoe import Adam, Smoe
    rng = np.random.default_rng(5)
    g = np.stack(np.meshgrid(*[np.linspace(0, 1, n) for n in (32, 32, 6)], indexing="ij"), -1)
    vid = np.clip(0.5 + 0.3 * np.sin(5 * g[..., :1] + 3 * g[..., 1:2] + 2 * g[..., 2:]) * np.ones(3)
                  + 0.02 * rng.standard_normal((32, 32, 6, 3)), 0, 1).astype(np.float32)
    runs = []
    for factory in (None, OracleEngine):
        s = Smoe(vid, train_inverse_cov=False, kernels_per_dim=[2, 2, 1], batch_size=[8, 8, 6], use_determinant=True,
                 use_yuv=True, ssim_opt=True, **({} if factory is None else {"engine_factory": factory}))
        s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
        s.train(20, val_iter=10)
        runs.append(([v for _, v in s.get_losses()], s.get_params()))
    (lg, pg), (lo, po) = runs
    assert abs(lg[0] - lo[0]) < 2e-5
    assert np.allclose(lg, lo, atol=3e-3) and lg[-1] < lg[0] - 0.005
    assert np.median(np.abs(pg["nu_e"] - po["nu_e"])) < 2e-4
