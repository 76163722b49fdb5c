"""CPU tests of the ``Smoe`` facade's host logic, driven through a test double of the engine
(tests/fake_engine.py, plain-C oracle): entry-point names and layouts of the reference,
iteration / validation / kernel-list cadence (smoe.py:1485-1603), best snapshot, histories,
padding, checkpoint round trip, and the loud refusal of options outside the hot path."""
import numpy as np
import pytest

from fake_engine import OracleEngine
from oracle import smoe_oracle as o
from steered_mixture_of_experts_amd import blocks as blk
from steered_mixture_of_experts_amd.smoe import Adam, Smoe
from steered_mixture_of_experts_amd import utils


def _image(h, w, C=1, seed=0):
    gh, gw = -(-h // 16), -(-w // 16)
    b = blk.synthetic_blocks(gh * gw, (16, 16), C, seed)
    return blk.blocks_to_image(b, (gh * 16, gw * 16), (16, 16))[:h, :w]


def _make(img, **kw):
    s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, engine_factory=OracleEngine, **kw)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1.0))
    return s


def test_train_matches_the_numpy_restatement():
    img = _image(64, 48)
    s = _make(img)
    assert s.num_blocks == 12 and s.start_batches == 12 and s.kernels == 4 and s.dim_domain == 2
    s.train(20, val_iter=10)
    blocks, _ = blk.image_to_blocks(img, (16, 16))
    cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=4)
    tgt = blocks.reshape(12, -1, 1)
    p0 = o.init_params(blocks, [2, 2])
    pn, st, info = o.fit(p0, tgt, o.block_coords((16, 16)), cfg, 20, val_iter=10, dtype=np.float32)
    got = s.get_params()
    assert set(got) == {"pis", "musX", "A_diagonal", "A_corr", "gamma_e", "nu_e"}
    assert got["A_diagonal"].shape == (12, 4, 2, 2) and got["gamma_e"].shape == (12, 4, 2, 1)
    for k in ("nu_e", "musX", "pis", "gamma_e"):
        assert np.abs(got[k] - pn[k]).max() < 5e-4, k
    assert [it for it, _ in s.get_losses()] == [0, 10, 20] and s.get_iter() == 20
    want_loss = [float(np.mean(l)) for l in info["hist"]["loss"]]
    assert np.allclose([v for _, v in s.get_losses()], want_loss, rtol=2e-2)
    want_mse = [float(np.sum(e) / (12 * 256) * 65536) for e in info["hist"]["sse"]]
    assert np.allclose([v for _, v in s.get_mses()], want_mse, rtol=2e-2)
    assert s.get_num_pis()[-1][1] == 48
    # first evaluation is exact (same initial parameters, same maths)
    assert abs(s.get_losses()[0][1] - want_loss[0]) < 1e-7
    rec = s.get_reconstruction()
    assert rec.shape == img.shape and rec.dtype == np.float32
    am = s.get_weight_matrix_argmax()
    assert am.shape == (64, 48) and am.max() < 48 and (am[:16, :16] < 4).all() and (am[16:32, :16] // 4 == 3).all()
    assert s.get_weight_matrix().shape == (12, 4, 16, 16)
    best = s.get_best_params()
    assert np.abs(best["nu_e"] - info["best"]["nu_e"]).max() < 5e-4
    assert abs(s.get_psnr() - (-10 * np.log10(np.mean((rec.astype(np.float64) - img) ** 2)))) < 1e-9


def test_single_block_is_configs0():
    img = blk.synthetic_blocks(1, (16, 16), 1, 7)[0]
    s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[None], use_determinant=True, engine_factory=OracleEngine)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1.0))
    assert s.num_blocks == 1 and s.batch_size_valued == (16, 16)
    loss, mse, num_pi, num_sv = s.run_batched(train=False, update_reconstruction=True)
    assert num_pi == 4 and num_sv == 0
    l2, _, _, _ = s.run_batched(train=True)
    assert abs(l2 - loss) < 1e-7        # a train pass reports the loss at the parameters it started from
    assert len(s.kernel_list_per_batch) == 1 and s.kernel_list_per_batch[0].dtype == bool


def test_padded_image_and_loss_mask():
    img = _image(40, 37, C=3, seed=3)
    s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, use_yuv=True,
             engine_factory=OracleEngine)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1.0))
    assert s.padded and s.num_blocks == 9 and s.use_yuv
    s.train(4, val_iter=2)
    assert s.get_reconstruction().shape == (40, 37, 3)
    g = Smoe(img[..., :1], train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], engine_factory=OracleEngine)
    assert not g.use_yuv                      # forced off unless C == 3 (smoe_test.py:41-44)


def test_checkpoint_restore_and_model_pickle(tmp_path):
    img = _image(32, 32)
    s = _make(img)
    s.train(6, val_iter=3)
    path = str(tmp_path / "ckpt.pkl")
    s.checkpoint(path)
    ref = s.get_params()
    s.train(3, val_iter=3)
    after = s.get_params()
    t = _make(img)
    t.restore(path)
    assert all(np.array_equal(t.get_params()[k], ref[k]) for k in ref) and t.get_iter() == 6
    t.train(3, val_iter=3)
    again = t.get_params()
    assert all(np.array_equal(again[k], after[k]) for k in ref)      # resume is exact (Adam slots + beta powers)
    mp = str(tmp_path / "params_10.pkl")
    utils.save_model(s, mp)
    cp = utils.load_checkpoint(mp)
    assert set(cp) >= {"params", "mses", "losses", "num_pis", "use_yuv", "use_determinant", "batch_size"}
    assert np.array_equal(utils.load_params(mp)["nu_e"], after["nu_e"])
    r = Smoe(img, train_inverse_cov=False, init_params=cp["params"], batch_size=list(cp["batch_size"]), use_determinant=True,
             engine_factory=OracleEngine)
    assert np.array_equal(r.get_reconstruction(), s.get_reconstruction())


def test_options_outside_the_hot_path_are_refused():
    img = _image(16, 16)
    for kw in ({"overlap_of_batches": 2}, {"add_kernel_slots": 4}, {"train_svs": True}):
        with pytest.raises(NotImplementedError):
            Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], engine_factory=OracleEngine, **kw)
    with pytest.raises(AssertionError):
        Smoe(img, train_inverse_cov=False, batch_size=[16, 16], engine_factory=OracleEngine)
    with pytest.raises(ValueError):
        Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16, 4], engine_factory=OracleEngine)
    s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], engine_factory=OracleEngine)
    with pytest.raises(AssertionError):
        s.train(1)                               # "no optimizer found" (smoe.py:1492)
    assert utils.psnr(65536.0 * 1e-3, 8) == pytest.approx(30.0)


def test_quantizer_matches_per_block_restatement_and_mode1_training():
    from oracle.quantizer_oracle import quantize_block
    from steered_mixture_of_experts_amd.quantizer import quantize_params, rescaler
    img = _image(48, 32)
    s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, quantization_mode=1,
             bit_depths=[20, 18, 6, 10, 10], engine_factory=OracleEngine)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1.0))
    s.train(6, val_iter=3)
    assert [i for i, _ in s.get_qlosses()] == [0, 3, 6] and len(s.get_qmses()) == 3
    p = s.get_params()
    p["pis"][2, 1] = 0.0                      # a dropped kernel
    q = quantize_params(s, p)
    r = rescaler(s, q)
    for b in range(s.num_blocks):
        idx, qb, rb = quantize_block({k: v[b] for k, v in p.items()}, s.bit_depths)
        assert np.array_equal(q["used_kernels"][b], idx)
        for k in ("A_diagonal", "A_corr", "musX", "nu_e", "pis", "gamma_e"):
            assert np.array_equal(q[k][b][idx], qb[k]), k
            assert np.allclose(r[k][b][idx], rb[k], rtol=0, atol=1e-15), k
        assert np.allclose(r["A"][b][idx], rb["A"])
        assert (r["pis"][b][~idx] == 0).all()
    # 6-bit nu_e: the rescaled expert means sit on a 63-step lattice between the block's min and max
    lo, hi = q["lower_bounds"]["nu_e"], q["upper_bounds"]["nu_e"]
    t = (r["nu_e"] - lo) / np.maximum(hi - lo, 1e-30) * 63
    assert np.allclose(t, np.round(t), atol=1e-6)
    # quantised reconstruction exists, differs little from the unquantised one
    qr, rr = s.get_qreconstruction(), s.get_reconstruction()
    assert qr.shape == rr.shape and np.abs(qr - rr).mean() < 0.02
    # fixed bounds (quantize_pis) path
    s.quantize_pis, s.lower_bounds, s.upper_bounds = True, [-2500, -.3, -5, 0, -32], [2500, 1.3, 5, 2, 32]
    q2 = quantize_params(s, p)
    assert q2["lower_bounds"]["pis"].max() == 0 and q2["upper_bounds"]["pis"].min() == 2


def test_smoe_reconstruction_entry_point(tmp_path):
    import steered_mixture_of_experts_amd.smoe_reconstruction as rec
    import steered_mixture_of_experts_amd.smoe as smod
    img = _image(32, 48)
    s = _make(img)
    s.train(4, val_iter=2)
    np.save(tmp_path / "img.npy", np.uint8(np.round(img * 255)))
    mp = str(tmp_path / "params_4.pkl")
    utils.save_model(s, mp)
    orig_factory = smod._default_engine_factory
    smod._default_engine_factory = lambda cfg, device: OracleEngine(cfg, device)
    try:
        out = str(tmp_path / "out")
        recon, loss, mse = rec.main(str(tmp_path / "img.npy"), out, mp)
        assert np.array_equal(recon, s.get_reconstruction())
        assert np.array_equal(np.load(out + "/4_reconstruction.npy"), np.uint8(np.round(recon * 255)))
        recon_q, _, _ = rec.main(str(tmp_path / "img.npy"), out, mp, quant_params=True)
        assert recon_q.shape == recon.shape and np.abs(recon_q - recon).mean() < 0.02
    finally:
        smod._default_engine_factory = orig_factory


def test_shared_kernel_facade_matches_the_numpy_restatement():
    from fake_engine import OracleSharedEngine
    from steered_mixture_of_experts_amd.smoe import SharedSmoe
    img = _image(64, 48)
    s = SharedSmoe(img, train_inverse_cov=False, kernels_per_dim=[4, 3], batch_size=[16, 16], use_determinant=True,
                   engine_factory=OracleSharedEngine)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1.0))
    assert s.num_batches == 12 and s.kernels == 12
    s.train(8, val_iter=4)
    p0 = o.shared_init_params(img, [4, 3])
    coords = o.global_batch_coords((64, 48), (16, 16))
    tb, _ = blk.image_to_blocks(img, (16, 16))
    cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=12)
    pn, st, info = o.shared_fit(p0, tb.reshape(12, -1, 1), coords, cfg, 8, val_iter=4, dtype=np.float32)
    got = s.get_params()
    assert got["musX"].shape == (12, 2) and got["A_diagonal"].shape == (12, 2, 2)       # the reference's layout
    for k in got:
        assert np.allclose(got[k], pn[k][0], rtol=1e-5, atol=1e-6), k
    assert [i for i, _ in s.get_losses()] == [0, 4, 8]
    assert np.allclose([v for _, v in s.get_losses()], info["hist"]["loss"], rtol=1e-6)
    assert np.allclose([v for _, v in s.get_mses()], info["hist"]["mse"], rtol=1e-6)
    assert s.get_reconstruction().shape == img.shape and s.get_weight_matrix_argmax().max() < 12
    assert np.array_equal(np.array(s.kernel_list_per_batch), info["lists"])
    with pytest.raises(ValueError):
        SharedSmoe(img, train_inverse_cov=False, kernels_per_dim=[4, 3], batch_size=[10, 16], engine_factory=OracleSharedEngine)


def test_shared_facade_with_overlapping_batches():
    """overlap_of_batches: the halo only feeds the kernel-list influence test and the readmission probes
    (smoe.py:909-923,2322-2331); the facade follows oracle.shared_fit with the extended windows."""
    from fake_engine import OracleSharedEngine
    from steered_mixture_of_experts_amd.smoe import SharedSmoe
    img = _image(64, 48, seed=4)
    s = SharedSmoe(img, train_inverse_cov=False, kernels_per_dim=[8, 6], batch_size=[16, 16], use_determinant=True, overlap_of_batches=3,
                   engine_factory=OracleSharedEngine)
    assert s.batch_size == (22, 22) and s.batch_size_valued == (16, 16) and s.overlap == 3        # smoe.py:244-245
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
    s.train(6, val_iter=3, ukl_iter=2)
    p0 = o.shared_init_params(img, [8, 6])
    coords = o.global_batch_coords((64, 48), (16, 16))
    halo = o.global_halo_coords((64, 48), (16, 16), 3)
    tb, _ = blk.image_to_blocks(img, (16, 16))
    cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=48, lr_steer=0.01)
    pn, st, info = o.shared_fit(p0, tb.reshape(12, -1, 1), coords, cfg, 6, val_iter=3, ukl_iter=2, dtype=np.float32,
                                halo_coords=halo)
    _, _, plain = o.shared_fit(p0, tb.reshape(12, -1, 1), coords, cfg, 6, val_iter=3, ukl_iter=2, dtype=np.float32)
    got = s.get_params()
    for k in got:       # the engine double sums the batch gradients in fp64, shared_fit in fp32; where |g| ~ eps
        # Adam amplifies that rounding to a fraction of lr per step (A_corr starts at a zero gradient)
        assert np.allclose(got[k], pn[k][0], rtol=1e-4, atol=1e-3 if k.startswith("A_") else 1e-5), k
    assert np.array_equal(np.array(s.kernel_list_per_batch), info["lists"])
    assert info["lists"].sum() > plain["lists"].sum()              # the halo keeps neighbouring kernels listed
    assert s.get_reconstruction().shape == img.shape


def test_start_batches_selects_the_block_shape():
    img = _image(64, 64)
    s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], start_batches=16, batch_size=[None], engine_factory=OracleEngine)
    assert s.batch_size_valued == (16, 16) and s.num_blocks == 16            # get_batch_shape, smoe.py:229,243
    s1 = Smoe(img[:16, :16], train_inverse_cov=False, kernels_per_dim=[2, 2], engine_factory=OracleEngine)
    assert s1.batch_size_valued == (16, 16) and s1.num_blocks == 1


def test_only_y_gamma_and_use_diff_center():
    img = _image(32, 32, C=3, seed=9)
    s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, use_yuv=True, only_y_gamma=True,
             use_diff_center=True, engine_factory=OracleEngine)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1.0))
    assert s.only_y_gamma and np.array_equal(s.get_params()["musX"], np.zeros((4, 4, 2), np.float32))
    s.train(6, val_iter=3)
    p = s.get_params()
    assert np.abs(p["gamma_e"][..., 0]).max() > 0 and not p["gamma_e"][..., 1:].any()      # smoe.py:725-729
    assert 0 < np.abs(p["musX"]).max() < 0.05                                               # offsets, not centres
    # the same fit without use_diff_center: centres = grid + offsets
    t = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, use_yuv=True, only_y_gamma=True,
             engine_factory=OracleEngine)
    t.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1.0))
    t.train(6, val_iter=3)
    grid = blk.gen_domain_grid([2, 2], 2).astype(np.float32)
    assert np.allclose(t.get_params()["musX"], grid[None] + p["musX"], atol=1e-7)
    # only_y_gamma is dropped when the image is not YUV (smoe_test.py:41-44)
    g = Smoe(img[..., :1], train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], only_y_gamma=True, engine_factory=OracleEngine)
    assert not g.only_y_gamma


def test_training_cli_mirrors_the_reference_flags(tmp_path):
    import steered_mixture_of_experts_amd.smoe as smod
    import steered_mixture_of_experts_amd.smoe_test as cli
    from fake_engine import OracleSharedEngine
    img = _image(32, 48)
    np.save(tmp_path / "img.npy", np.uint8(np.round(img * 255)))
    parser = cli.build_parser()
    d = vars(parser.parse_args(["-i", "x", "-r", "y"]))
    # defaults of the reference CLI (smoe_test.py:262-352) for everything on the hot path
    assert d["iterations"] == 10000 and d["validation_iterations"] == 100 and d["kernels_per_dim"] == [12]
    assert d["base_lr"] == 0.001 and d["lr_div"] == 100 and d["lr_mult"] == 1000 and d["use_determinant"] is True
    assert d["bit_depths"] == [20, 18, 6, 10, 10] and d["batch_size"] == [None] and d["train_inverse_cov"] is False
    f1, f2 = smod._default_engine_factory, smod._default_shared_factory
    smod._default_engine_factory = lambda cfg, device: OracleEngine(cfg, device)
    smod._default_shared_factory = lambda cfg, device: OracleSharedEngine(cfg, device)
    try:
        out = str(tmp_path / "res")
        s = cli.main(parser.parse_args(["-i", str(tmp_path / "img.npy"), "-r", out, "-k", "2", "-bz", "16", "16",
                                        "-n", "4", "-v", "2"]))
        assert s.num_blocks == 6 and s.optimizer2._lr == 1e-5 and s.optimizer3._lr == 1.0
        cp = utils.load_checkpoint(out + "/params_last.pkl")
        assert cp["params"]["nu_e"].shape == (6, 4, 1) and [i for i, _ in cp["losses"]] == [0, 2, 4]
        assert np.load(out + "/reconstruction.npy").shape == (32, 48, 1)
        g = cli.main(parser.parse_args(["-i", str(tmp_path / "img.npy"), "-r", out, "-k", "3", "-bz", "16", "16",
                                        "-n", "2", "-v", "2", "--mode", "shared"]))
        assert g.kernels == 9 and g.num_batches == 6
        q = cli.main(parser.parse_args(["-i", str(tmp_path / "img.npy"), "-r", out, "-k", "2", "-bz", "16", "16",
                                        "-n", "2", "-v", "2", "-ssim", "true"]))
        assert q.ssim_opt and 0.0 < q.get_losses()[-1][1] < 1.0            # 1 - SSIM
        gs = cli.main(parser.parse_args(["-i", str(tmp_path / "img.npy"), "-r", out, "-k", "3", "-bz", "16", "16",
                                         "-n", "2", "-v", "2", "-ssim", "true", "--mode", "shared"]))
        assert gs.ssim_opt and 0.0 < gs.get_losses()[-1][1] < 1.0
        with pytest.raises(NotImplementedError):
            cli.main(parser.parse_args(["-i", str(tmp_path / "img.npy"), "-r", out, "-is", "100"]))
    finally:
        smod._default_engine_factory, smod._default_shared_factory = f1, f2


def test_ssim_opt_fits_one_minus_ssim():
    """ssim_opt (smoe.py:929,980-1011): the facade trains on 1 - SSIM; losses follow oracle.fit with ssim_opt."""
    img = _image(32, 32, seed=12)
    s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, ssim_opt=True,
             engine_factory=OracleEngine)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
    s.train(6, val_iter=3)
    tb, _ = blk.image_to_blocks(img, (16, 16))
    cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=4, lr_steer=0.01, ssim_opt=True)
    p0 = o.init_params(tb, [2, 2])
    pn, _, info = o.fit(p0, tb.reshape(4, -1, 1), o.block_coords((16, 16)), cfg, 6, val_iter=3, dtype=np.float32)
    got = s.get_params()
    for k in got:
        assert np.allclose(got[k], pn[k], rtol=1e-5, atol=1e-6), k
    losses = [v for _, v in s.get_losses()]
    assert 0.0 < losses[-1] < losses[0] < 1.0
    for kw in ({"loss_mask": np.ones((32, 32), np.float32)}, {"batch_size": [12, 12]}):
        with pytest.raises(NotImplementedError):
            Smoe(img, train_inverse_cov=False, **{"kernels_per_dim": [2, 2], "batch_size": [16, 16], "ssim_opt": True,
                         "engine_factory": OracleEngine, **kw})
    with pytest.raises(ValueError):
        Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[4, 16], ssim_opt=True, engine_factory=OracleEngine)


def test_ssim_opt_on_video_blocks():
    """3-d blocks: SYMMETRIC pad by 5 on the three axes and the 11x11x11 window (smoe.py:999-1003); the reference's
    4-frame default cannot be padded, so the blocks here are 6 frames deep."""
    rng = np.random.default_rng(5)
    g = np.stack(np.meshgrid(*[np.linspace(0, 1, n) for n in (16, 16, 6)], indexing="ij"), -1)
    vid = np.clip(0.5 + 0.3 * np.sin(5 * g[..., :1] + 3 * g[..., 1:2] + 2 * g[..., 2:]) * np.ones(3)
                  + 0.02 * rng.standard_normal((16, 16, 6, 3)), 0, 1).astype(np.float32)
    s = Smoe(vid, train_inverse_cov=False, kernels_per_dim=[2, 2, 1], batch_size=[8, 8, 6], use_determinant=True,
             ssim_opt=True, engine_factory=OracleEngine)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
    s.train(6, val_iter=3)
    losses = [v for _, v in s.get_losses()]
    assert 0.0 < losses[-1] < losses[0] < 1.0
    with pytest.raises(ValueError):
        Smoe(vid[:, :, :4], train_inverse_cov=False, kernels_per_dim=[2, 2, 1], batch_size=[8, 8, 4], ssim_opt=True,
             engine_factory=OracleEngine)


@pytest.mark.parametrize("mode,qpis", [(0, True), (2, False), (3, False)])
def test_fake_quantised_fit_through_the_facade(mode, qpis):
    """quantize_pis / quantization_mode 2, 3 (smoe.py:474-538): the facade trains on the fake-quantised graph
    (oracle.fit with the same configuration), counts kernels with qpis > 0 and keeps writing qparams at the
    validation points (smoe.py:1498-1499,1539-1540)."""
    img = _image(32, 32, seed=3)
    kw = dict(quantization_mode=mode, quantize_pis=qpis, bit_depths=[14, 12, 8, 10, 10],
              lower_bounds=[-60, -.3, -1, 0, -4], upper_bounds=[60, 1.3, 2, 2, 4])
    s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, engine_factory=OracleEngine, **kw)
    assert s.quantize_pis                                     # implied by modes >= 2 (smoe_test.py:36-37)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
    s.train(6, val_iter=3)
    tb, _ = blk.image_to_blocks(img, (16, 16))
    cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=4, lr_steer=0.01, quantization_mode=mode,
                         quantize_pis=True, bit_depths=tuple(kw["bit_depths"]), lower_bounds=tuple(kw["lower_bounds"]),
                         upper_bounds=tuple(kw["upper_bounds"]))
    p0 = o.init_params(tb, [2, 2])
    pn, _, info = o.fit(p0, tb.reshape(4, -1, 1), o.block_coords((16, 16)), cfg, 6, val_iter=3, dtype=np.float32)
    got = s.get_params()
    for k in got:
        assert np.allclose(got[k], pn[k], rtol=1e-5, atol=2e-6), k
    assert np.allclose([v for _, v in s.get_losses()], [float(np.mean(l)) for l in info["hist"]["loss"]], rtol=1e-5)
    if mode >= 1:
        assert s.qparams is not None and s.rparams is None
    # a kernel whose pi rounds to 0 on the 10-bit lattice of [0, 2] is not counted (pis_mask = qpis > 0)
    p = s.get_params()
    p["pis"][0, 0] = 0.0004
    s2 = Smoe(img, train_inverse_cov=False, init_params=p, batch_size=[16, 16], use_determinant=True, engine_factory=OracleEngine, **kw)
    s2.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
    assert s2.run_batched(train=False)[2] == 15


@pytest.mark.parametrize("mode", [0, 2, 3])
def test_shared_facade_with_fake_quantised_variables(mode):
    """SharedSmoe with quantize_pis / quantization_mode 2 / 3 (smoe.py:474-530) follows oracle.shared_fit."""
    from fake_engine import OracleSharedEngine
    from steered_mixture_of_experts_amd.smoe import SharedSmoe
    img = _image(32, 48, seed=6)
    kw = dict(quantization_mode=mode, quantize_pis=True, bit_depths=[14, 12, 8, 10, 10],
              lower_bounds=[-60, -.3, -1, 0, -4], upper_bounds=[60, 1.3, 2, 2, 4])
    s = SharedSmoe(img, train_inverse_cov=False, kernels_per_dim=[3, 4], batch_size=[16, 16], use_determinant=True,
                   engine_factory=OracleSharedEngine, **kw)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
    s.train(6, val_iter=3)
    p0 = o.shared_init_params(img, [3, 4])
    coords = o.global_batch_coords((32, 48), (16, 16))
    tb, _ = blk.image_to_blocks(img, (16, 16))
    cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=12, lr_steer=0.01, quantization_mode=mode,
                         quantize_pis=True, bit_depths=tuple(kw["bit_depths"]), lower_bounds=tuple(kw["lower_bounds"]),
                         upper_bounds=tuple(kw["upper_bounds"]))
    pn, _, info = o.shared_fit(p0, tb.reshape(6, -1, 1), coords, cfg, 6, val_iter=3, dtype=np.float32)
    got = s.get_params()
    for k in got:
        assert np.allclose(got[k], pn[k][0], rtol=1e-4, atol=1e-3 if k.startswith("A_") else 1e-5), k
    assert np.allclose([v for _, v in s.get_losses()], info["hist"]["loss"], rtol=1e-5)
    with pytest.raises(ValueError):
        SharedSmoe(img, train_inverse_cov=False, kernels_per_dim=[3, 4], batch_size=[16, 16], engine_factory=OracleSharedEngine,
                   quantization_mode=4)


def test_shared_facade_kernel_count_as_norm_l1():
    """kernel_count_as_norm_l1 in the shared mode (smoe.py:1022-1027): pis_l1 is divided by the image-wide count of
    kernels with qpis > 0 instead of start_pis."""
    from fake_engine import OracleSharedEngine
    from steered_mixture_of_experts_amd.smoe import SharedSmoe
    img = _image(32, 48, seed=6)
    p0 = o.shared_init_params(img, [3, 4])
    p0["pis"][0, 2] = 0.0004                       # rounds to 0 on the pis lattice: not counted
    out = {}
    for kc in (False, True):
        s = SharedSmoe(img, train_inverse_cov=False, init_params={k: v[0].copy() for k, v in p0.items()}, batch_size=[16, 16],
                       use_determinant=True, engine_factory=OracleSharedEngine, quantize_pis=True, kernel_count_as_norm_l1=kc)
        s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
        s.train(4, val_iter=2, pis_l1=0.5)
        cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=12, lr_steer=0.01, quantize_pis=True, pis_l1=0.5,
                             kernel_count_as_norm_l1=kc)
        coords = o.global_batch_coords((32, 48), (16, 16))
        tb, _ = blk.image_to_blocks(img, (16, 16))
        pn, _, info = o.shared_fit(p0, tb.reshape(6, -1, 1), coords, cfg, 4, val_iter=2, dtype=np.float32)
        assert np.allclose([v for _, v in s.get_losses()], info["hist"]["loss"], rtol=1e-5)
        out[kc] = info["hist"]["loss"][0]
    assert out[True] > out[False]                   # 11 kernels counted instead of start_pis = 12


def test_shared_facade_only_y_gamma_and_diff_center():
    """only_y_gamma (gamma_mask, smoe.py:725-729) and use_diff_center (smoe.py:390-394,746-747) in the shared mode."""
    from fake_engine import OracleSharedEngine
    from steered_mixture_of_experts_amd.smoe import SharedSmoe
    img = _image(32, 32, C=3, seed=2)
    s = SharedSmoe(img, train_inverse_cov=False, kernels_per_dim=[3, 3], batch_size=[16, 16], use_determinant=True, use_yuv=True,
                   only_y_gamma=True, use_diff_center=True, engine_factory=OracleSharedEngine)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
    assert np.array_equal(s.get_params()["musX"], np.zeros((9, 2), np.float32))      # offsets start at zero
    s.train(4, val_iter=2)
    p = s.get_params()
    assert np.abs(p["musX"]).max() < 0.02 and np.abs(p["musX"]).max() > 0
    assert not p["gamma_e"][..., 1:].any() and p["gamma_e"][..., 0].any()
    p0 = o.shared_init_params(img, [3, 3])
    tb, _ = blk.image_to_blocks(img, (16, 16))
    cfg = o.OracleConfig(block_shape=(16, 16), channels=3, kernels=9, lr_steer=0.01, use_yuv=True, only_y_gamma=True)
    pn, _, _ = o.shared_fit(p0, tb.reshape(4, -1, 3), o.global_batch_coords((32, 32), (16, 16)), cfg, 4, val_iter=2)
    assert np.allclose(p["musX"] + p0["musX"][0], pn["musX"][0], atol=2e-6)
    assert np.allclose(p["nu_e"], pn["nu_e"][0], atol=2e-6)


def test_inverse_covariance_fit_through_the_facade():
    """train_inverse_cov=True (the reference constructor default): A_init is squared (smoe.py:2162), the fit
    follows oracle.fit with the symmetric form."""
    img = _image(32, 32, seed=8)
    s = Smoe(img, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, train_inverse_cov=True,
             engine_factory=OracleEngine)
    assert np.allclose(np.diagonal(s.get_params()["A_diagonal"], axis1=-2, axis2=-1), 36.0)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.05))
    s.train(6, val_iter=3)
    tb, _ = blk.image_to_blocks(img, (16, 16))
    cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=4, lr_steer=0.05, train_inverse_cov=True)
    p0 = o.init_params(tb, [2, 2])
    p0["A_diagonal"] = p0["A_diagonal"] ** 2
    pn, _, info = o.fit(p0, tb.reshape(4, -1, 1), o.block_coords((16, 16)), cfg, 6, val_iter=3, dtype=np.float32)
    got = s.get_params()
    for k in got:
        assert np.allclose(got[k], pn[k], rtol=1e-5, atol=2e-6), k
    assert s.get_losses()[-1][1] < s.get_losses()[0][1]


def test_constructor_default_is_the_references_inverse_covariance_form():
    """smoe.py:41: Smoe(...) defaults to train_inverse_cov=True (the CLI passes False, smoe_test.py:342); a saved model
    records the form and smoe_reconstruction rebuilds with it."""
    img = _image(16, 16)
    s = Smoe(img, kernels_per_dim=[2, 2], batch_size=[16, 16], engine_factory=OracleEngine)
    assert s.train_inverse_cov and np.allclose(np.diagonal(s.get_params()["A_diagonal"], axis1=-2, axis2=-1), 36.0)
    q = Smoe(img, kernels_per_dim=[2, 2], batch_size=[16, 16], ssim_opt=True, quantize_pis=True, engine_factory=OracleEngine)
    q.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.05))
    q.train(2, val_iter=1)                                   # the default form composes with the SSIM loss
    assert 0 < q.get_losses()[-1][1] < 1


@pytest.mark.parametrize("ic", [False, True])
def test_radial_steering_through_the_facade(ic):
    """radial_as: get_params() returns the reference's (K,) steering variable per block, the fit follows oracle.fit with
    tied diagonals, A_corr stays untouched."""
    img = _image(32, 32, seed=13)
    s = Smoe(img, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, radial_as=True, train_inverse_cov=ic,
             engine_factory=OracleEngine)
    a0 = 36.0 if ic else 6.0
    assert s.get_params()["A_diagonal"].shape == (4, 4) and np.allclose(s.get_params()["A_diagonal"], a0)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.05))
    s.train(6, val_iter=3, u_l1=0.002)
    tb, _ = blk.image_to_blocks(img, (16, 16))
    cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=4, lr_steer=0.05, radial_as=True, train_inverse_cov=ic,
                         u_l1=0.002)
    p0 = o.init_params(tb, [2, 2])
    if ic:
        p0["A_diagonal"] = p0["A_diagonal"] ** 2
    pn, _, _ = o.fit(p0, tb.reshape(4, -1, 1), o.block_coords((16, 16)), cfg, 6, val_iter=3, dtype=np.float32)
    got = s.get_params()
    assert np.allclose(got["A_diagonal"], pn["A_diagonal"][:, :, 0, 0], rtol=1e-5) and not got["A_corr"].any()
    assert np.abs(got["A_diagonal"] - a0).max() > 0.01
    for k in ("nu_e", "musX", "pis"):
        assert np.allclose(got[k], pn[k], rtol=1e-5, atol=2e-6), k
    r = Smoe(img, init_params=got, batch_size=[16, 16], use_determinant=True, radial_as=True, train_inverse_cov=ic,
             engine_factory=OracleEngine)                    # the (K,) layout round-trips through init_params
    assert np.array_equal(r.get_reconstruction(), s.get_reconstruction())


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_radial_steering_with_a_quantisation_mode(mode):
    """radial_as with quantization_mode 1 (quantise the (K,) variable at validation, quantizer.py radial paths), 2
    (fixed-range fake quant inside the graph) and 3 (range from the data, smoe.py:498-504): the fit follows oracle.fit,
    the quantised evaluation runs."""
    img = _image(32, 32, seed=14)
    kw = dict(quantization_mode=mode, quantize_pis=True, bit_depths=[14, 12, 8, 10, 10], lower_bounds=[-60, -.3, -1, 0, -4],
              upper_bounds=[60, 1.3, 2, 2, 4])
    s = Smoe(img, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, radial_as=True, train_inverse_cov=False,
             engine_factory=OracleEngine, **kw)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.05))
    s.train(4, val_iter=2)
    tb, _ = blk.image_to_blocks(img, (16, 16))
    cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=4, lr_steer=0.05, radial_as=True,
                         quantization_mode=mode, quantize_pis=True, bit_depths=tuple(kw["bit_depths"]),
                         lower_bounds=tuple(kw["lower_bounds"]), upper_bounds=tuple(kw["upper_bounds"]))
    pn, _, _ = o.fit(o.init_params(tb, [2, 2]), tb.reshape(4, -1, 1), o.block_coords((16, 16)), cfg, 4, val_iter=2,
                     dtype=np.float32)
    got = s.get_params()
    assert got["A_diagonal"].shape == (4, 4)
    assert np.allclose(got["A_diagonal"], pn["A_diagonal"][:, :, 0, 0], rtol=1e-5) and not got["A_corr"].any()
    assert s.qparams is not None and "A_corr" not in s.qparams and s.qparams["A_diagonal"].shape == (4, 4)
    if mode == 1:
        assert [i for i, _ in s.get_qlosses()] == [0, 2, 4]
        assert s.rparams["A_diagonal"].shape == (4, 4, 2, 2) and not s.rparams["A_corr"].any()


@pytest.mark.parametrize("mode", [2, 3])
def test_fake_quantised_centre_offsets_through_the_facade(mode):
    """use_diff_center with quantization_mode 2 / 3: the facade hands the kernel grid to the engine, the fit follows
    oracle.fit with ``mus_grid``; get_params() reports the offsets."""
    img = _image(32, 32, seed=15)
    kw = dict(quantization_mode=mode, quantize_pis=True, bit_depths=[14, 10, 8, 10, 10], lower_bounds=[-60, -.06, -1, 0, -4],
              upper_bounds=[60, .08, 2, 2, 4])
    s = Smoe(img, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, use_diff_center=True, train_inverse_cov=False,
             engine_factory=OracleEngine, **kw)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.05))
    s.train(4, val_iter=2)
    tb, _ = blk.image_to_blocks(img, (16, 16))
    p0 = o.init_params(tb, [2, 2])
    cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=4, lr_steer=0.05, mus_grid=p0["musX"].copy(),
                         quantization_mode=mode, quantize_pis=True, bit_depths=tuple(kw["bit_depths"]),
                         lower_bounds=tuple(kw["lower_bounds"]), upper_bounds=tuple(kw["upper_bounds"]))
    pn, _, _ = o.fit(p0, tb.reshape(4, -1, 1), o.block_coords((16, 16)), cfg, 4, val_iter=2, dtype=np.float32)
    got = s.get_params()
    assert np.allclose(got["musX"], pn["musX"] - cfg.mus_grid, atol=1e-6) and np.abs(got["musX"]).max() > 0
    for k in ("pis", "nu_e", "A_diagonal"):
        assert np.allclose(got[k], pn[k], rtol=1e-5, atol=1e-6), k
    from fake_engine import OracleSharedEngine
    from steered_mixture_of_experts_amd.smoe import SharedSmoe
    g = SharedSmoe(img, kernels_per_dim=[3, 3], batch_size=[16, 16], use_diff_center=True, train_inverse_cov=False,
                   engine_factory=OracleSharedEngine, **kw)
    g.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.05))
    g.train(3, val_iter=3)
    assert np.abs(g.get_params()["musX"]).max() > 0 and g.get_losses()[-1][1] < g.get_losses()[0][1] * 1.5


def test_shared_facade_with_the_ssim_loss_on_video():
    from fake_engine import OracleSharedEngine
    from steered_mixture_of_experts_amd.smoe import SharedSmoe
    rng = np.random.default_rng(6)
    g = np.stack(np.meshgrid(*[np.linspace(0, 1, n) for n in (16, 16, 6)], indexing="ij"), -1)
    vid = np.clip(0.5 + 0.3 * np.sin(5 * g[..., :1] + 3 * g[..., 1:2] + 2 * g[..., 2:]) + 0.02 * rng.standard_normal((16, 16, 6, 1)),
                  0, 1).astype(np.float32)
    s = SharedSmoe(vid, train_inverse_cov=False, kernels_per_dim=[2, 2, 1], batch_size=[8, 8, 6], use_determinant=True,
                   ssim_opt=True, engine_factory=OracleSharedEngine)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
    s.train(4, val_iter=2)
    losses = [v for _, v in s.get_losses()]
    assert 0.0 < losses[-1] < losses[0]


def test_shared_facade_with_the_ssim_loss():
    from fake_engine import OracleSharedEngine
    from steered_mixture_of_experts_amd.smoe import SharedSmoe
    img = _image(32, 48, seed=11)
    s = SharedSmoe(img, train_inverse_cov=False, kernels_per_dim=[4, 5], batch_size=[16, 16], use_determinant=True, ssim_opt=True,
                   engine_factory=OracleSharedEngine)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
    s.train(4, val_iter=2)
    p0 = o.shared_init_params(img, [4, 5])
    tb, _ = blk.image_to_blocks(img, (16, 16))
    cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=20, lr_steer=0.01, ssim_opt=True)
    pn, _, info = o.shared_fit(p0, tb.reshape(6, -1, 1), o.global_batch_coords((32, 48), (16, 16)), cfg, 4, val_iter=2)
    got = s.get_params()
    for k in got:
        assert np.allclose(got[k], pn[k][0], rtol=1e-4, atol=1e-3 if k.startswith("A_") else 1e-5), k
    assert np.allclose([v for _, v in s.get_losses()], info["hist"]["loss"], rtol=1e-5)
    assert 0 < s.get_losses()[-1][1] < s.get_losses()[0][1] < 1


def test_shared_facade_inverse_covariance_default():
    """SharedSmoe mirrors the reference constructor default train_inverse_cov=True (A_init squared, maha = r^T A r)."""
    from fake_engine import OracleSharedEngine
    from steered_mixture_of_experts_amd.smoe import SharedSmoe
    img = _image(32, 48, seed=14)
    s = SharedSmoe(img, kernels_per_dim=[3, 4], batch_size=[16, 16], use_determinant=True, engine_factory=OracleSharedEngine)
    assert s.train_inverse_cov and np.allclose(s.get_params()["A_diagonal"][:, 0, 0], 64.0)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.05))
    s.train(4, val_iter=2)
    p0 = o.shared_init_params(img, [3, 4])
    p0["A_diagonal"] = p0["A_diagonal"] ** 2
    tb, _ = blk.image_to_blocks(img, (16, 16))
    cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=12, lr_steer=0.05, train_inverse_cov=True)
    pn, _, info = o.shared_fit(p0, tb.reshape(6, -1, 1), o.global_batch_coords((32, 48), (16, 16)), cfg, 4, val_iter=2)
    got = s.get_params()
    for k in got:
        assert np.allclose(got[k], pn[k][0], rtol=1e-4, atol=1e-3 if k.startswith("A_") else 1e-5), k
    assert np.allclose([v for _, v in s.get_losses()], info["hist"]["loss"], rtol=1e-5)


def test_shared_facade_quantises_at_validation_points():
    """quantization_mode 1 in the shared mode (smoe.py:1498-1505,1539-1545): qparams / rparams of the global kernel
    set at every validation, the rescaled parameters are evaluated beside the trained ones."""
    from fake_engine import OracleSharedEngine
    from steered_mixture_of_experts_amd.smoe import SharedSmoe
    img = _image(32, 48, seed=15)
    s = SharedSmoe(img, kernels_per_dim=[3, 4], batch_size=[16, 16], use_determinant=True, train_inverse_cov=False,
                   quantization_mode=1, quantize_pis=True, bit_depths=[12, 10, 8, 10, 8], engine_factory=OracleSharedEngine)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
    s.train(4, val_iter=2)
    assert [i for i, _ in s.get_qlosses()] == [0, 2, 4] and s.qparams["musX"].shape == (1, 12, 2)
    assert s.rparams["A"].shape == (1, 12, 2, 2)
    q, r = s.get_qreconstruction(), s.get_reconstruction()
    assert q.shape == r.shape and 0 < np.abs(q - r).mean() < 0.05
    assert s.get_qmses()[-1][1] >= s.get_mses()[-1][1] * 0.9          # quantised parameters reconstruct slightly worse


def test_shared_facade_with_a_loss_mask():
    """loss_mask in the shared mode (smoe.py:550,932,1674-1677): per-pixel weights on the margin loss."""
    from fake_engine import OracleSharedEngine
    from steered_mixture_of_experts_amd.smoe import SharedSmoe
    img = _image(32, 48, seed=16)
    mask = np.random.default_rng(1).uniform(0, 1, size=(32, 48)).astype(np.float32)
    mask[:8] = 0.0
    s = SharedSmoe(img, kernels_per_dim=[3, 4], batch_size=[16, 16], use_determinant=True, train_inverse_cov=False,
                   loss_mask=mask, engine_factory=OracleSharedEngine)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
    s.train(4, val_iter=2)
    p0 = o.shared_init_params(img, [3, 4])
    tb, _ = blk.image_to_blocks(img, (16, 16))
    mb, _ = blk.image_to_blocks(mask[..., None], (16, 16))
    cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=12, lr_steer=0.01)
    pn, _, info = o.shared_fit(p0, tb.reshape(6, -1, 1), o.global_batch_coords((32, 48), (16, 16)), cfg, 4, val_iter=2,
                               loss_w=mb.reshape(6, -1))
    plain = o.shared_fit(p0, tb.reshape(6, -1, 1), o.global_batch_coords((32, 48), (16, 16)), cfg, 4, val_iter=2)[2]
    got = s.get_params()
    for k in got:
        assert np.allclose(got[k], pn[k][0], rtol=1e-4, atol=1e-3 if k.startswith("A_") else 1e-5), k
    assert np.allclose([v for _, v in s.get_losses()], info["hist"]["loss"], rtol=1e-5)
    assert info["hist"]["loss"][0] < 0.8 * plain["hist"]["loss"][0]            # the mask really weighs the loss


def test_resume_equals_uninterrupted_training(tmp_path):
    """train(2n) == train(n) + checkpoint / restore into a fresh model + train(n), including the best snapshot and the
    per-block stop rule's iteration-0 losses (restore() lost _best_loss_blocks and _loss0)."""
    img = _image(32, 48, seed=21)

    def fresh():
        s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True,
                 engine_factory=OracleEngine)
        s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1.0))
        return s
    a = fresh()
    a.train(8, val_iter=2)
    b = fresh()
    b.train(4, val_iter=2)
    path = b.checkpoint(str(tmp_path / "ck.pkl"))
    c = fresh()
    c.restore(path)
    assert c._best_loss_blocks is not None and c._loss0 is not None and c.best_loss == b.best_loss
    assert np.array_equal(c._loss0.numpy(), b._loss0.numpy())
    c.train(4, val_iter=2)
    pa, pc = a.get_params(), c.get_params()
    ba, bc = a.get_best_params(), c.get_best_params()
    for k in pa:
        assert np.array_equal(pa[k], pc[k]), k
        assert np.array_equal(ba[k], bc[k]), k
    assert c.get_iter() == a.get_iter() == 8


def test_unsupported_kernel_counts_are_padded_with_prior_zero_kernels():
    """A kernel count without its own instantiation runs on the next instantiated count; the padding kernels have prior
    0, so `bool_mask = kernel_list & pis > 0` (smoe.py:480,738) drops them: results equal the unpadded model's."""
    img = _image(32, 32, seed=5)

    class Padded:
        """engine factory that, like libsmoe_hip.so, only has even kernel counts"""
        def __call__(self, cfg, device):
            assert cfg.kernels % 2 == 0
            return OracleEngine(cfg, device)

        @staticmethod
        def padded_kernels(d, C, K):
            return K + (K % 2)
    kw = dict(train_inverse_cov=False, kernels_per_dim=[1, 3], batch_size=[16, 16], use_determinant=True, quantize_pis=True)
    ref = Smoe(img, engine_factory=OracleEngine, **kw)
    pad = Smoe(img, engine_factory=Padded(), **kw)
    assert ref._kp == 3 and pad._kp == 4 and pad.kernels == 3
    for s in (ref, pad):
        s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1e-2))
        s.train(6, val_iter=3, pis_l1=0.1)
    pr, pp = ref.get_params(), pad.get_params()
    for k in pr:
        assert pp[k].shape == pr[k].shape and np.allclose(pp[k], pr[k], rtol=1e-6, atol=1e-7), k
    assert np.array_equal(ref.get_reconstruction(), pad.get_reconstruction())
    assert np.array_equal(ref.get_weight_matrix_argmax(), pad.get_weight_matrix_argmax())
    assert ref.get_weight_matrix().shape == pad.get_weight_matrix().shape
    assert ref.get_num_pis() == pad.get_num_pis() and np.allclose(ref.get_losses(), pad.get_losses(), rtol=1e-6)
    assert len(pad.kernel_list_per_batch[0]) == 3


def test_shared_mode_cli_checkpoint_decodes(tmp_path):
    """CLI --mode shared -> smoe_reconstruction (the shared-mode writer dropped the bounds, quantize_pis and
    train_inverse_cov; decoding crashed on the missing bounds and would have rebuilt another graph)."""
    import steered_mixture_of_experts_amd.smoe as smod
    import steered_mixture_of_experts_amd.smoe_reconstruction as rec
    import steered_mixture_of_experts_amd.smoe_test as cli
    from fake_engine import OracleSharedEngine
    img = _image(32, 48, seed=9)
    np.save(tmp_path / "img.npy", np.uint8(np.round(img * 255)))
    f1, f2, f3 = smod._default_engine_factory, smod._default_shared_factory, rec._shared_engine_factory
    smod._default_engine_factory = lambda cfg, device: OracleEngine(cfg, device)
    smod._default_shared_factory = lambda cfg, device: OracleSharedEngine(cfg, device)
    rec._shared_engine_factory = OracleSharedEngine
    try:
        out = str(tmp_path / "res")
        g = cli.main(cli.build_parser().parse_args(["-i", str(tmp_path / "img.npy"), "-r", out, "-k", "3", "-bz", "16", "16",
                                                    "-n", "2", "-v", "2", "--mode", "shared", "-tiv", "true"]))
        cp = utils.load_checkpoint(out + "/params_last.pkl")
        assert cp["mode"] == "shared" and cp["train_inverse_cov"] is True and cp["quantized_pis"] is True
        assert cp["lower_bounds"] is not None and cp["upper_bounds"] is not None and cp["params"]["pis"].shape == (9,)
        recon, loss, mse = rec.main(str(tmp_path / "img.npy"), str(tmp_path / "dec"), out + "/params_last.pkl")
        assert np.array_equal(recon, g.get_reconstruction())          # the same graph: inverse-covariance form, quantised pis
    finally:
        smod._default_engine_factory, smod._default_shared_factory, rec._shared_engine_factory = f1, f2, f3


def test_pixel_sub_sampling_is_the_reference_subset_fit():
    """sampling_percentage < 100 (smoe.py:1664-1667): the reference feeds a random subset of a block's pixels, drawn with the
    error-proportional probabilities of the last reconstruction pass (smoe.py:906-907); the facade gives the drawn pixels
    the loss weight N / n instead.  Same loss, same gradients: checked against the restatement run on the subset itself."""
    img = _image(32, 32, seed=3)
    s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, engine_factory=OracleEngine)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1e-2))
    s.run_batched(train=False, update_reconstruction=True)
    prob = s._sampl_prob.numpy()
    assert prob.shape == (4, 256) and np.allclose(prob.sum(axis=1), 1.0, atol=1e-5)
    rec = s.get_reconstruction()
    tb, _ = blk.image_to_blocks(img, (16, 16))
    rb, _ = blk.image_to_blocks(rec, (16, 16))
    err = ((rb - tb) ** 2).reshape(4, 256, -1).mean(axis=2)
    assert np.allclose(prob, err / err.sum(axis=1, keepdims=True), rtol=1e-4, atol=1e-9)
    w = s._sample_pixels(25).numpy()
    assert ((w > 0).sum(axis=1) == 64).all() and np.allclose(w[w > 0], 4.0)
    # pixels with larger error are drawn more often
    hits = np.zeros_like(w)
    for _ in range(40):
        hits += s._sample_pixels(25).numpy() > 0
    hi, lo = prob > np.median(prob, axis=1, keepdims=True), prob <= np.median(prob, axis=1, keepdims=True)
    assert hits[hi].mean() > hits[lo].mean()
    # weighted full block == the graph on the subset (loss and every gradient)
    cfg = o.OracleConfig(block_shape=(16, 16), channels=1, kernels=4)
    p0 = o.init_params(tb, [2, 2])
    coords = o.block_coords((16, 16))
    tgt = tb.reshape(4, -1, 1)
    active = np.ones((4, 4), bool)
    full = o.forward(p0, tgt, coords, active, cfg, w, np.float64, want_grads=True)
    for b in range(4):
        sel = np.flatnonzero(w[b] > 0)
        cfgs = o.OracleConfig(block_shape=(len(sel), 1), channels=1, kernels=4)
        pb = {k: v[b:b + 1] for k, v in p0.items()}
        sub = o.forward(pb, tgt[b:b + 1, sel], coords[sel], active[b:b + 1], cfgs, None, np.float64, want_grads=True)
        assert np.allclose(full["loss"][b], sub["loss"][0], rtol=1e-10)
        for k in sub["grads"]:
            assert np.allclose(full["grads"][k][b], sub["grads"][k][0], rtol=1e-9, atol=1e-14), k
    # and the training loop runs on fresh draws
    l0 = s.run_batched(train=False)[0]
    s.train(6, val_iter=3, sampling_percentage=50)
    assert s.get_iter() == 6 and s.get_losses()[-1][1] < l0 * 1.05


def test_sub_sampled_training_pass_as_the_first_call_with_a_regulariser():
    """run_batched(train=True, sampling_percentage < 100, pis_l1 != 0) as the FIRST call.  The sampling
    probabilities start uniform (smoe.py:270-272), so no evaluation pass is nested into the training pass: the engine built
    for (pis_l1, u_l1) stays the one that runs, and the kernel lists are pruned by that pass alone."""
    img = _image(32, 32, seed=5)
    s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, engine_factory=OracleEngine)
    assert np.allclose(s._sampl_prob.numpy(), 1.0 / 256)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1e-2))
    p0 = s.get_params()
    loss, mse, num_pi, _ = s.run_batched(pis_l1=0.1, u_l1=0.01, train=True, sampling_percentage=50)
    eng = s._engine
    assert eng.cfg.pis_l1 == pytest.approx(0.1) and eng.cfg.u_l1 == pytest.approx(0.01)
    assert np.isfinite(loss) and np.isfinite(mse) and num_pi == 16
    assert np.allclose(s._sampl_prob.numpy(), 1.0 / 256)          # a training pass does not touch them (smoe.py:1768)
    p1 = s.get_params()
    assert not np.array_equal(p0["nu_e"], p1["nu_e"])             # the Adam step happened
    # the l1 terms are part of the reported loss: the same pass without them is smaller by pis_l1 sum(pi)/K0 + u_l1 sum(diag A)
    s2 = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, engine_factory=OracleEngine)
    s2.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1e-2))
    s2._sample_gen = None
    l_reg = s2.run_batched(pis_l1=0.1, u_l1=0.01, train=False)[0]
    l_plain = s2.run_batched(train=False)[0]
    assert l_reg - l_plain == pytest.approx(0.1 * 1.0 / 4 + 0.01 * 4 * 2 * 6.0, rel=1e-4)   # per block: sum(pi) = 1, K0 = 4, A = 6 I


def test_ragged_image_mse_counts_the_image_pixels_only():
    """on an image that is not a multiple of the block the reported MSE of a reconstruction pass equals the
    MSE of the cropped reconstruction (what get_psnr() reports), not that of the padded tiling."""
    img = _image(40, 52, seed=4)                       # 16x16 blocks: padded to 48 x 64
    s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True, engine_factory=OracleEngine)
    assert s.padded and s.num_pixel == 40 * 52
    _, mse, _, _ = s.run_batched(train=False, update_reconstruction=True)
    rec = s.get_reconstruction()
    want = float(np.mean((rec.astype(np.float64) - img.astype(np.float64)) ** 2)) * (2 ** 8) ** 2
    assert abs(mse - want) < 1e-6 * want
    assert abs(s.get_psnr() - (-10 * np.log10(want / 65536.0))) < 1e-6
