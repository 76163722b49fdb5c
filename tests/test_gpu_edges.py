"""Edge cases of the hot path through the C ABI: empty inputs, blocks without kernels, dead priors, ragged images,
the largest blocks an instantiation takes, refused shapes."""
import numpy as np
import pytest
import torch

from oracle import smoe_oracle as o
from test_gpu_parity import _bits_to_mask, _close, _engine, _mask_to_bits, _planar, _setup, _to_dev

pytestmark = pytest.mark.gpu


def test_zero_blocks_is_a_no_op():
    """B = 0 (an image sharded over more ranks than it has blocks): every entry point returns without a launch."""
    eng = _engine((16, 16), 1, 4)
    p = eng.new_params(0)
    st = eng.new_adam_state(p)
    act = torch.zeros((0,), dtype=torch.int32, device="cuda")
    T = torch.zeros((0, 1, 256), device="cuda")
    out = eng.forward(T, p, act, want_recon=True, want_argmax=True, want_gate=True)
    assert out["loss"].shape == (0,) and out["recon"].shape == (0, 1, 256)
    eng.fit(T, p, st, act, 5)
    eng.update_kernel_list(p, act)
    s = eng.reduce_scalars(out["loss"], out["sse"], act)
    torch.cuda.synchronize()
    assert s.cpu().numpy().tolist() == [0.0, 0.0, 0.0]
    eng.close()


@pytest.mark.parametrize("tiling", [16, 64])
def test_blocks_without_kernels_and_dead_priors(tiling):
    """An empty kernel list (w = 0 / max(1e-11, 0), smoe.py:819-827 -> reconstruction 0), kernels whose prior is <= 0
    (pis_mask, smoe.py:480,738) and a single surviving kernel: forward values, the pruned list and one Adam step."""
    shape, C, kpd = (16, 16), 1, [2, 2]
    cfg, p, coords, tgt, K = _setup(shape, C, kpd, False, B=24, seed=31)
    active = np.ones((24, K), bool)
    active[0] = False                      # nothing listed
    active[1] = [True, False, False, False]
    p["pis"][2] = 0.0                      # everything listed, nothing alive
    p["pis"][3, :3] = -0.25                # one kernel alive
    p["pis"][4, 1] = 0.0
    ref = o.forward(p, tgt, coords, active, cfg, None, np.float32, want_grads=True)
    eng = _engine(shape, C, K)
    eng.set_tiling(tiling)
    dp = _to_dev(p)
    act = torch.from_numpy(_mask_to_bits(active).view(np.int32)).cuda()
    out = eng.forward(_planar(tgt), dp, act, want_recon=True)
    torch.cuda.synchronize()
    recon = out["recon"].cpu().numpy().transpose(0, 2, 1)
    assert not recon[0].any() and not recon[2].any()
    frac = (np.clip(ref["y"], 0, 1) * 255 + 0.5) % 1.0
    tie = (frac < 3e-4) | (frac > 1 - 3e-4)
    assert (np.abs(recon - ref["recon"])[~tie] < 1e-7).all()
    assert _close(out["loss"].cpu().numpy(), ref["loss"], rtol=3e-5).all()
    got = _bits_to_mask(act.cpu().numpy().view(np.uint32), K)
    assert np.array_equal(got, ref["active_new"])
    assert not got[0].any() and not got[2].any() and got[3].tolist() == [False, False, False, True]
    # one Adam step from the ORIGINAL lists: untouched blocks keep their parameters, their Adam slots stay 0
    act = torch.from_numpy(_mask_to_bits(active).view(np.int32)).cuda()
    st = eng.new_adam_state(dp)
    before = {k: v.clone() for k, v in dp.items()}
    eng.fit(_planar(tgt), dp, st, act, 1)
    torch.cuda.synchronize()
    for name in o.PARAM_NAMES:
        m = st.m[name].cpu().numpy()
        assert not m[0].any() and not m[2].any(), name
        assert torch.equal(dp[name][0], before[name][0]) and torch.equal(dp[name][2], before[name][2]), name
        g = ref["grads"][name]
        scale = np.abs(g).max() + 1e-30
        assert np.abs(m / 0.1 - g).max() / scale < 5e-5, name
    eng.close()


def test_ragged_image_through_the_facade():
    """An image whose sides are not multiples of the block (30x50 with 16x16 blocks): zero padding + loss mask
    (smoe.py:550,932), cropped reconstruction; the GPU facade follows the oracle-backed facade."""
    from fake_engine import OracleEngine
    from steered_mixture_of_experts_amd import blocks as blk
    from steered_mixture_of_experts_amd.smoe import Adam, Smoe
    b = blk.synthetic_blocks(8, (16, 16), 1, 99)
    img = blk.blocks_to_image(b, (32, 64), (16, 16))[:30, :50]

    def run(factory):
        s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True,
                 engine_factory=factory)
        s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1e-2))
        s.train(8, val_iter=4)
        return s

    g, c = run(None), run(OracleEngine)
    assert g.get_reconstruction().shape == (30, 50, 1)
    assert np.allclose([v for _, v in g.get_losses()], [v for _, v in c.get_losses()], rtol=2e-4)
    assert np.allclose([v for _, v in g.get_mses()], [v for _, v in c.get_mses()], rtol=2e-3)
    assert np.abs(g.get_reconstruction() - c.get_reconstruction()).max() <= 1.0 / 255 + 1e-6
    assert (g.get_reconstruction() != c.get_reconstruction()).mean() < 0.01


@pytest.mark.parametrize("shape,C,kpd,yuv", [((64, 64), 1, [2, 2], False), ((32, 64), 3, [2, 2], True),
                                            ((32, 32), 3, [2, 4], True), ((16, 16, 8), 3, [2, 2, 1], True)])
def test_largest_blocks(shape, C, kpd, yuv):
    """Big blocks (4 096 pixels with one channel; 2 048 with three, where the fp32 targets of a workgroup fill the LDS): forward and one
    gentle fit step against the restatement."""
    cfg, p, coords, tgt, K = _setup(shape, C, kpd, yuv, B=5, seed=17, lr_steer=1e-2)
    active = np.ones((5, K), bool)
    ref = o.forward(p, tgt, coords, active, cfg, None, np.float32, want_grads=True)
    eng = _engine(shape, C, K, use_yuv=yuv, lr_steer=1e-2)
    dp = _to_dev(p)
    act = torch.from_numpy(_mask_to_bits(active).view(np.int32)).cuda()
    out = eng.forward(_planar(tgt), dp, act, want_recon=False, update_active=False)
    torch.cuda.synchronize()
    assert _close(out["loss"].cpu().numpy(), ref["loss"], rtol=5e-5).all()
    st = eng.new_adam_state(dp)
    eng.fit(_planar(tgt), dp, st, act, 1)
    torch.cuda.synchronize()
    for name in o.PARAM_NAMES:
        g = ref["grads"][name]
        scale = np.abs(g).max() + 1e-30
        assert np.abs(st.m[name].cpu().numpy() / 0.1 - g).max() / scale < 2e-3, name     # quantiser ties move single pixels
    eng.close()


def test_refused_shapes_say_why():
    from steered_mixture_of_experts_amd import _lib
    for kw, code in ((dict(block_shape=(128, 128), channels=1, kernels=4), _lib.SMOE_ERR_INVALID),      # > 8192 pixels
                     (dict(block_shape=(16, 16), channels=1, kernels=5), _lib.SMOE_ERR_UNSUPPORTED),   # no such instantiation (the facade pads to 6)
                     (dict(block_shape=(16, 16), channels=1, kernels=3, ssim_opt=True), _lib.SMOE_ERR_UNSUPPORTED),   # basic triple: margin loss only
                     (dict(block_shape=(16, 16), channels=1, kernels=9, quantization_mode=3), _lib.SMOE_ERR_UNSUPPORTED),
                     (dict(block_shape=(4, 4), channels=1, kernels=4, ssim_opt=True), _lib.SMOE_ERR_INVALID)):
        with pytest.raises(_lib.SmoeError) as e:
            _engine(kw.pop("block_shape"), kw.pop("channels"), kw.pop("kernels"), **kw)
        assert e.value.code == code and str(e.value)


@pytest.mark.parametrize("tiling", [16, 64, 128, 816, 264])
def test_sub_sampled_pass_prunes_by_the_fed_pixels_only(tiling):
    """Pixel sub-sampling (smoe.py:1664-1667): the reference feeds only the drawn pixels, so a kernel stays on the list iff it
    has influence on a DRAWN pixel (smoe.py:829,1763-1766).  The engine is told that its loss weights are a sample
    (smoe_set_sampling): weight-0 pixels do not vote.  A kernel is confined to a corner of the block; the sample avoids that
    corner: the sampled pass drops the kernel, the same weights as a plain loss mask keep it."""
    shape, C, kpd = (16, 16), 1, [2, 2]
    B = 9
    cfg, p, coords, tgt, K = _setup(shape, C, kpd, False, B, 77, perturb=False)
    p["A_diagonal"][:, 0] *= 6.0                       # kernel 0: narrow, around its centre (0.25, 0.25)
    fed = np.ones((B, 256), bool).reshape(B, 16, 16)
    fed[:, :9, :9] = False                            # nothing drawn in the corner that kernel 0 reaches
    fed = fed.reshape(B, 256)
    n = fed.sum(axis=1)[:, None]
    lw = np.where(fed, 256.0 / n, 0.0).astype(np.float32)
    active = np.ones((B, K), bool)
    ref_s = o.forward(p, tgt, coords, active, cfg, lw, np.float32, want_grads=True, fed=fed)
    ref_m = o.forward(p, tgt, coords, active, cfg, lw, np.float32, want_grads=True)
    assert not ref_s["active_new"][:, 0].any() and ref_m["active_new"][:, 0].all()      # the scenario is what it claims
    for sample, ref in ((True, ref_s), (False, ref_m)):
        eng = _engine(shape, C, K)
        eng.set_tiling(tiling)
        dp = _to_dev(p)
        st = eng.new_adam_state(dp)
        act = torch.from_numpy(_mask_to_bits(active).view(np.int32)).cuda()
        eng.fit(_planar(tgt), dp, st, act, 1, loss_w=torch.from_numpy(lw).cuda(), loss_w_is_sample=sample)
        torch.cuda.synchronize()
        got = _bits_to_mask(act.cpu().numpy().view(np.uint32), K)
        assert np.array_equal(got, ref["active_new"]), (sample, tiling)
        # the gradients are those of the weighted pass either way (weight 0 = no contribution)
        g = st.m["nu_e"].cpu().numpy() / 0.1
        assert np.abs(g - ref["grads"]["nu_e"]).max() < 3e-5 * np.abs(ref["grads"]["nu_e"]).max()
        eng.close()
