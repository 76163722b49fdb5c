"""GPU parity of the shared-kernel image mode (smoe_shared_*) against the CPU
restatement (oracle.shared_pass / shared_fit / shared_readmit): one global kernel set, per-batch
kernel lists, gradients accumulated over the batches of a pass, one Adam step per pass."""
import numpy as np
import pytest
import torch

from oracle import smoe_oracle as o
from steered_mixture_of_experts_amd import blocks as blk

pytestmark = pytest.mark.gpu

CASES = [
    # image shape, batch shape, C, kernels_per_dim, yuv
    ((64, 64), (16, 16), 1, [4, 4], False),
    ((64, 96), (32, 32), 3, [3, 5], True),
    ((96, 64), (32, 64), 1, [6, 4], False),        # 2048-pixel batches: 8 pixels per lane
    ((32, 32, 8), (16, 16, 4), 3, [2, 2, 2], True),
    ((48, 40), (16, 8), 1, [12, 12], False),       # 144 kernels: several LDS chunks of the list
]


def _image(shape, C, seed):
    d = len(shape)
    bs = (16, 16) if d == 2 else (16, 16, 4)
    g = [-(-s // b) for s, b in zip(shape, bs)]
    b = blk.synthetic_blocks(int(np.prod(g)), bs, C, seed)
    full = blk.blocks_to_image(b, tuple(gi * bi for gi, bi in zip(g, bs)), bs)
    return np.ascontiguousarray(full[tuple(slice(0, s) for s in shape)])


def _setup(shape, bshape, C, kpd, yuv, seed=5, perturb=True, **kw):
    img = _image(shape, C, seed)
    p = o.shared_init_params(img, kpd)
    K = p["pis"].shape[1]
    rng = np.random.default_rng(seed)
    if perturb:
        p["A_corr"] = (rng.normal(size=p["A_corr"].shape) * 1.0).astype(np.float32)
        p["gamma_e"] = (rng.normal(size=p["gamma_e"].shape) * 0.1).astype(np.float32)
        p["musX"] = (p["musX"] + rng.normal(size=p["musX"].shape) * 0.01).astype(np.float32)
        p["pis"] = (p["pis"] * rng.uniform(0.5, 1.5, size=p["pis"].shape)).astype(np.float32)
    cfg = o.OracleConfig(block_shape=bshape, channels=C, kernels=K, use_yuv=yuv, **kw)
    coords = o.global_batch_coords(shape, bshape)
    tb, _ = blk.image_to_blocks(img, bshape)
    NB = tb.shape[0]
    tgt = tb.reshape(NB, -1, C)
    return img, p, cfg, coords, tgt, K, NB


def _engine(shape, bshape, C, K, yuv, **kw):
    from steered_mixture_of_experts_amd.engine import SharedConfig, SharedEngine
    return SharedEngine(SharedConfig(image_shape=shape, batch_shape=bshape, channels=C, kernels=K, use_yuv=yuv, **kw))


def _dev(p):
    return {k: torch.from_numpy(np.ascontiguousarray(v[0])).cuda() for k, v in p.items()}


def _bits(mask):
    NB, K = mask.shape
    KW = (K + 31) // 32
    out = np.zeros((NB, KW), np.uint32)
    for k in range(K):
        out[:, k >> 5] |= (mask[:, k].astype(np.uint32) << np.uint32(k & 31))
    return out


def _mask(bits, K):
    return np.stack([(bits[:, k >> 5] >> np.uint32(k & 31)) & 1 for k in range(K)], axis=1).astype(bool)


@pytest.mark.parametrize("shape,bshape,C,kpd,yuv", CASES)
def test_shared_forward_parity(shape, bshape, C, kpd, yuv):
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, yuv, pis_l1=0.2, u_l1=0.003)
    rng = np.random.default_rng(1)
    lists = rng.uniform(size=(NB, K)) < 0.9
    p["pis"][0, 1] = 0.0
    ref = o.shared_pass(p, tgt, coords, lists, cfg, np.float32)
    ref64 = o.shared_pass(p, tgt, coords, lists, cfg, np.float64)
    eng = _engine(shape, bshape, C, K, yuv, pis_l1=0.2, u_l1=0.003)
    assert eng.num_batches == NB
    dp = _dev(p)
    dl = torch.from_numpy(_bits(lists).view(np.int32)).cuda()
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    out = eng.forward(T, dp, dl, want_recon=True, want_argmax=True)
    torch.cuda.synchronize()
    recon = out["recon"].cpu().numpy().transpose(0, 2, 1)
    frac = (np.clip(ref64["y"], 0, 1) * 255 + 0.5) % 1.0
    tie = (frac < 2e-4) | (frac > 1 - 2e-4)
    d = np.abs(recon - ref["recon"])
    assert (d[~tie] < 1e-7).all() and (d <= 1.0001 / 255).all()
    refq = o.forward(o._bcast(p, NB), tgt, coords, lists, cfg, None, np.float32, q_override=recon)
    assert np.allclose(out["loss"].cpu().numpy(), refq["loss"], rtol=3e-5, atol=1e-9)
    assert np.allclose(out["sse"].cpu().numpy(), refq["sse"], rtol=3e-5, atol=1e-9)
    near_tau = (np.abs(ref64["w"] - 0.5 / 256) < 1e-6).any(axis=2)
    new = _mask(dl.cpu().numpy().view(np.uint32), K)
    assert (new == ref["lists_new"])[~near_tau].all()
    srt = np.sort(ref64["wt"], axis=1)
    close_top = (srt[:, -1, :] - srt[:, -2, :]) < 1e-6
    ok = (out["argmax"].cpu().numpy() == ref["argmax"]) | close_top | near_tau.any(axis=1)[:, None]
    assert ok.all()
    eng.close()


@pytest.mark.parametrize("shape,bshape,C,kpd,yuv", CASES)
def test_shared_one_step_gradients_and_adam(shape, bshape, C, kpd, yuv):
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, yuv, pis_l1=0.05, u_l1=0.001)
    lists = np.ones((NB, K), bool)
    eng = _engine(shape, bshape, C, K, yuv, pis_l1=0.05, u_l1=0.001)
    dp = _dev(p)
    dl = eng.new_lists()
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    fw = eng.forward(T, dp, dl, want_recon=True, update_lists=False)
    recon = fw["recon"].cpu().numpy().transpose(0, 2, 1)
    f64 = o.forward(o._bcast(p, NB), tgt, coords, lists, cfg, None, np.float64, want_grads=True, q_override=recon)
    g64 = {k: v.sum(axis=0) for k, v in f64["grads"].items()}
    st = eng.new_adam_state(dp)
    loss = torch.zeros(NB, device="cuda")
    eng.accumulate(T, dp, dl, loss_out=loss)
    eng.apply(dp, st)
    torch.cuda.synchronize()
    assert np.allclose(loss.cpu().numpy(), f64["loss"], rtol=3e-5)
    # gate values on the influence threshold / blends on the clip edge make the gradient discontinuous
    bad = (np.abs(f64["w"] - 0.5 / 256) < 1e-6).any() or ((np.abs(f64["y"]) < 1e-6) | (np.abs(f64["y"] - 1) < 1e-6)).any()
    for name in o.PARAM_NAMES:
        scale = np.abs(g64[name]).max() + 1e-30
        err = np.abs(st.m[name].cpu().numpy() / 0.1 - g64[name]).max() / scale
        assert err < (2e-3 if bad else 3e-5), (name, err)
    assert st.step == 1
    eng.close()


def test_shared_short_fit_and_readmission():
    shape, bshape, C, kpd = (64, 64), (16, 16), 1, [4, 4]
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, False, perturb=False)
    n = 12
    p32, st32, i32 = o.shared_fit(p, tgt, coords, cfg, n, val_iter=6, dtype=np.float32)
    p64, _, i64 = o.shared_fit(p, tgt, coords, cfg, n, val_iter=6, dtype=np.float64)
    eng = _engine(shape, bshape, C, K, False)
    dp = _dev(p)
    dl = eng.new_lists()
    st = eng.new_adam_state(dp)
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    hist = []
    f0 = eng.forward(T, dp, dl, want_recon=False)
    hist.append(float(f0["loss"].mean()))
    for _ in range(2):
        eng.fit(T, dp, st, dl, 6)
        eng.update_kernel_list(dp, dl)
        fv = eng.forward(T, dp, dl, want_recon=False)
        hist.append(float(fv["loss"].mean()))
    torch.cuda.synchronize()
    assert abs(hist[0] - i32["hist"]["loss"][0]) < 1e-7
    assert np.allclose(hist, i32["hist"]["loss"], rtol=0.05)
    assert hist[-1] < hist[0]
    got = {k: v.cpu().numpy()[None] for k, v in dp.items()}
    for name in ("nu_e", "musX", "pis", "gamma_e"):
        dev = np.median(np.abs(got[name] - p32[name]))
        floor = np.median(np.abs(p32[name] - p64[name]))
        assert dev <= 3 * floor + 1e-6, (name, dev, floor)
    lists = _mask(dl.cpu().numpy().view(np.uint32), K)
    assert (lists == i32["lists"]).mean() > 0.97
    # readmission alone, from empty lists
    empty = torch.zeros_like(dl)
    eng.update_kernel_list(dp, empty)
    want = o.shared_readmit({k: v for k, v in got.items()}, np.zeros((NB, K), bool), coords, cfg, np.float32)
    assert np.array_equal(_mask(empty.cpu().numpy().view(np.uint32), K), want)
    eng.close()


def test_shared_batch_sharding_matches_single_launch():
    """Batches processed as two ranges (what two ranks would do) + summed gradient buffers give the
    same accumulated gradient as one launch over all batches."""
    shape, bshape, C, kpd = (64, 64), (16, 16), 1, [4, 4]
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, False)
    eng = _engine(shape, bshape, C, K, False)
    dp = _dev(p)
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    l1 = eng.new_lists()
    eng.accumulate(T, dp, l1)
    full = eng.grad_buffer().clone()
    eng.discard_gradients()
    l2 = eng.new_lists()
    h = NB // 2 + 1
    eng.accumulate(T[:h].contiguous(), dp, l2[:h], first_batch=0)
    a = eng.grad_buffer().clone()
    eng.discard_gradients()
    eng.accumulate(T[h:].contiguous(), dp, l2[h:], first_batch=h)
    b = eng.grad_buffer().clone()
    torch.cuda.synchronize()
    assert float((a + b - full).abs().max()) <= 1e-10 * float(full.abs().max())      # two rank sums vs one: rounding only
    assert float(a.abs().max()) > 0 and float(b.abs().max()) > 0
    assert torch.equal(l1, l2)
    eng.discard_gradients()
    eng.close()


def test_shared_gradient_accumulation_is_bit_deterministic():
    """The accumulated gradient is summed per kernel in batch order (rows per batch + gather step, no atomics): two runs
    are bit-identical, a rank's batches split over several accumulate calls give the same bits as one call, and a 30-step
    fit repeats bit for bit (the fp64 atomicAdd left the order to the scheduler)."""
    shape, bshape, C, kpd = (128, 128), (16, 16), 1, [6, 6]
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, False)
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    eng = _engine(shape, bshape, C, K, False, quantize_pis=True)
    dp = _dev(p)
    bufs = []
    for split in (None, None, (0, 7, 23, NB)):
        lists = eng.new_lists()
        if split is None:
            eng.accumulate(T, dp, lists)
        else:
            for lo, hi in zip(split[:-1], split[1:]):
                eng.accumulate(T[lo:hi].contiguous(), dp, lists[lo:hi], first_batch=lo)
        torch.cuda.synchronize()
        bufs.append(eng.grad_buffer().clone())
        eng.discard_gradients()
    assert float(bufs[0].abs().max()) > 0
    assert torch.equal(bufs[0], bufs[1]) and torch.equal(bufs[0], bufs[2])
    eng.close()
    runs = []
    for _ in range(2):
        eng = _engine(shape, bshape, C, K, False, quantize_pis=True, lr_steer=0.05)
        dq = _dev(p)
        st = eng.new_adam_state(dq)
        lists = eng.new_lists()
        eng.forward(T, dq, lists, want_recon=False)
        eng.fit(T, dq, st, lists, 30)
        torch.cuda.synchronize()
        runs.append(({k: v.cpu().numpy() for k, v in dq.items()}, lists.cpu().numpy().copy()))
        eng.close()
    for k in runs[0][0]:
        assert np.array_equal(runs[0][0][k], runs[1][0][k]), k
    assert np.array_equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize("case", [((128, 128), (16, 16), 1, [6, 6], {}), ((64, 96), (16, 32), 3, [4, 6], {"train_inverse_cov": True}),
                                  ((32, 32, 8), (16, 16, 4), 1, [3, 3, 2], {"overlap": 2})])
def test_one_launch_fit_equals_the_two_launches_per_iteration(case, monkeypatch):
    """smoe_shared_fit as ONE launch (shared_fit_kernel: pass -> grid barrier -> gather + Adam -> grid barrier, n_iters times on
    a resident grid) against the loop of two launches per iteration it replaces (SMOE_SHARED_ONE_LAUNCH=0): the same device
    functions in the same order, so parameters, Adam slots, kernel lists, losses and the beta powers are bit-identical; also
    when the iterations are cut into several calls."""
    shape, bshape, C, kpd, kw = case
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, C == 3)
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    runs = []
    for mode, chunks in (("0", (25,)), ("1", (25,)), ("1", (7, 1, 17))):
        monkeypatch.setenv("SMOE_SHARED_ONE_LAUNCH", mode)
        eng = _engine(shape, bshape, C, K, C == 3, quantize_pis=True, lr_steer=0.05, **kw)
        dq = _dev(p)
        st = eng.new_adam_state(dq)
        lists = eng.new_lists()
        eng.forward(T, dq, lists, want_recon=False)
        loss = torch.zeros(NB, device="cuda")
        for n in chunks:
            eng.fit(T, dq, st, lists, n, loss_out=loss)
        f = eng.forward(T, dq, lists.clone(), want_recon=False)
        torch.cuda.synchronize()
        runs.append(({k: v.cpu().numpy() for k, v in dq.items()}, {k: v.cpu().numpy() for k, v in st.m.items()},
                     {k: v.cpu().numpy() for k, v in st.v.items()}, lists.cpu().numpy().copy(), loss.cpu().numpy(),
                     (float(st.c.beta1_power), float(st.c.beta2_power), int(st.c.step)), f["loss"].cpu().numpy()))
        eng.close()
    ref = runs[0]
    assert np.isfinite(ref[4]).all() and ref[5][2] == 25
    for other in runs[1:]:
        for a, b in zip(ref[:3], other[:3]):
            for k in a:
                assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(ref[3], other[3]) and np.array_equal(ref[4], other[4]) and ref[5] == other[5] and np.array_equal(ref[6], other[6])


def test_shared_facade_on_gpu_matches_the_oracle_backed_facade():
    from fake_engine import OracleSharedEngine
    from steered_mixture_of_experts_amd.smoe import Adam, SharedSmoe
    img = _image((64, 96), 1, 77)

    def run(factory):
        s = SharedSmoe(img, train_inverse_cov=False, kernels_per_dim=[4, 6], batch_size=[16, 32], use_determinant=True, engine_factory=factory)
        s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(1.0))
        s.train(10, val_iter=5)
        return s

    g, c = run(None), run(OracleSharedEngine)
    assert [i for i, _ in g.get_losses()] == [0, 5, 10]
    assert abs(g.get_losses()[0][1] - c.get_losses()[0][1]) < 1e-7
    assert np.allclose([v for _, v in g.get_losses()], [v for _, v in c.get_losses()], rtol=0.03)
    pg, pc = g.get_params(), c.get_params()
    for k in ("nu_e", "musX", "pis"):
        assert np.median(np.abs(pg[k] - pc[k])) < 2e-5, k
    assert abs(g.get_psnr() - c.get_psnr()) < 0.2
    assert g.get_losses()[-1][1] < g.get_losses()[0][1]
    assert g.get_weight_matrix_argmax().shape == (64, 96)


@pytest.mark.parametrize("shape,bshape,C,kpd,ov", [((48, 40), (16, 8), 1, [12, 12], 3),
                                                   ((64, 64), (32, 32), 3, [8, 8], 4),
                                                   ((32, 32, 8), (16, 16, 4), 1, [4, 4, 3], 2)])
def test_shared_overlap_halo(shape, bshape, C, kpd, ov):
    """overlap_of_batches > 0 (smoe.py:244-245): loss, reconstruction and gradients are those of the
    interior (smoe.py:909-923); the halo -- cut from the zero-padded domain (smoe.py:21,28) -- joins the
    influence test that prunes the kernel lists (smoe.py:829,1763-1766) and sets the readmission probes
    (smoe.py:2322-2331)."""
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, C == 3, perturb=True)
    halo = o.global_halo_coords(shape, bshape, ov)
    lists = np.ones((NB, K), bool)
    ref = o.shared_pass(p, tgt, coords, lists, cfg, np.float32, halo_coords=halo)
    plain = o.shared_pass(p, tgt, coords, lists, cfg, np.float32)
    fe64 = o.forward(o._bcast(p, NB), np.zeros(halo.shape[:2] + (C,)), halo.astype(np.float64), lists, cfg, None, np.float64)
    near_tau = (np.abs(fe64["w"] - 0.5 / 256) < 1e-6).any(axis=2)
    assert ref["lists_new"].sum() > plain["lists_new"].sum()       # the halo really keeps more kernels listed
    eng = _engine(shape, bshape, C, K, C == 3, overlap=ov)
    eng0 = _engine(shape, bshape, C, K, C == 3)
    dp = _dev(p)
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    dl, dl0 = eng.new_lists(), eng0.new_lists()
    out = eng.forward(T, dp, dl, want_recon=True)
    out0 = eng0.forward(T, dp, dl0, want_recon=True)
    torch.cuda.synchronize()
    assert torch.equal(out["recon"], out0["recon"]) and torch.equal(out["loss"], out0["loss"])
    new = _mask(dl.cpu().numpy().view(np.uint32), K)
    assert (new == ref["lists_new"])[~near_tau].all()
    assert (new != _mask(dl0.cpu().numpy().view(np.uint32), K)).any()
    # accumulated gradients do not see the halo
    l1, l0 = eng.new_lists(), eng0.new_lists()
    eng.accumulate(T, dp, l1)
    eng0.accumulate(T, dp, l0)
    torch.cuda.synchronize()
    ga, gb = eng.grad_buffer(), eng0.grad_buffer()                 # fp64 atomics: order differs between launches
    assert float((ga - gb).abs().max()) <= 1e-10 * float(gb.abs().max())
    eng.discard_gradients()
    eng0.discard_gradients()
    # readmission probes: min / max / mid of the padded window
    empty = torch.zeros_like(dl)
    eng.update_kernel_list(dp, empty)
    want = o.shared_readmit(p, np.zeros((NB, K), bool), halo, cfg, np.float32)
    assert np.array_equal(_mask(empty.cpu().numpy().view(np.uint32), K), want)
    # a short fit follows the oracle's lists
    cfg2 = o.OracleConfig(block_shape=bshape, channels=C, kernels=K, use_yuv=(C == 3), lr_steer=0.01)
    eng2 = _engine(shape, bshape, C, K, C == 3, overlap=ov, lr_steer=0.01)
    pn, _, info = o.shared_fit(p, tgt, coords, cfg2, 4, val_iter=4, ukl_iter=2, dtype=np.float32, halo_coords=halo)
    dp2, dl2 = _dev(p), eng2.new_lists()
    st2 = eng2.new_adam_state(dp2)
    eng2.forward(T, dp2, dl2, want_recon=False)
    for _ in range(2):
        eng2.fit(T, dp2, st2, dl2, 2)
        eng2.update_kernel_list(dp2, dl2)
    eng2.forward(T, dp2, dl2, want_recon=False)
    torch.cuda.synchronize()
    assert (_mask(dl2.cpu().numpy().view(np.uint32), K) == info["lists"]).mean() > 0.98
    for e in (eng, eng0, eng2):
        e.close()


QKW = dict(bit_depths=(14, 12, 8, 10, 10), lower_bounds=(-60, -.3, -1, 0, -4), upper_bounds=(60, 1.3, 2, 2, 4))


@pytest.mark.parametrize("mode", [0, 2, 3])
@pytest.mark.parametrize("shape,bshape,C,kpd,yuv", [CASES[0], CASES[1], CASES[3]])
def test_shared_fake_quantised_variables(shape, bshape, C, kpd, yuv, mode):
    """quantize_pis / quantization_mode 2 / 3 in the shared-kernel mode (smoe.py:474-530): forward, accumulated
    gradients through the straight-through masks (mode 3: routed to the image-wide extreme elements), one Adam
    step, readmission on the quantised variables."""
    kw = dict(quantization_mode=mode, quantize_pis=True, **QKW)
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, yuv, pis_l1=0.05, u_l1=0.001, **kw)
    d = len(shape)
    p["A_corr"] = p["A_corr"] * np.tril(np.ones((d, d), np.float32), -1)
    p["pis"][0, 1] = 0.0006            # rounds to 0 on the lattice: kernel absent
    p["pis"][0, 2] = 2.3               # clamped, no gradient
    p["musX"][0, 3, 0] = 1.31          # outside the fixed musX range (mode 2)
    lists = np.ones((NB, K), bool)
    ref = o.shared_pass(p, tgt, coords, lists, cfg, np.float32)
    eng = _engine(shape, bshape, C, K, yuv, pis_l1=0.05, u_l1=0.001, **kw)
    dp = _dev(p)
    dl = eng.new_lists()
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    fw = eng.forward(T, dp, dl, want_recon=True, update_lists=False)
    torch.cuda.synchronize()
    recon = fw["recon"].cpu().numpy().transpose(0, 2, 1)
    frac = (np.clip(ref["y"], 0, 1) * 255 + 0.5) % 1.0
    tie = (frac < 3e-4) | (frac > 1 - 3e-4)
    assert (np.abs(recon - ref["recon"])[~tie] < 1e-7).all()
    f32 = o.forward(o._bcast(p, NB), tgt, coords, lists, cfg, None, np.float32, want_grads=True, q_override=recon)
    g32 = {k: v.sum(axis=0) for k, v in f32["grads"].items()}
    assert np.allclose(fw["loss"].cpu().numpy(), f32["loss"], rtol=3e-5)
    st = eng.new_adam_state(dp)
    eng.accumulate(T, dp, dl)
    eng.apply(dp, st)
    torch.cuda.synchronize()
    bad = (np.abs(f32["w"] - 0.5 / 256) < 1e-6).any() or ((np.abs(f32["y"]) < 1e-6) | (np.abs(f32["y"] - 1) < 1e-6)).any()
    for name in o.PARAM_NAMES:
        scale = np.abs(g32[name]).max() + 1e-30
        err = np.abs(st.m[name].cpu().numpy() / 0.1 - g32[name]).max() / scale
        assert err < (2e-3 if bad else 5e-5), (name, err)
    m = {k: v.cpu().numpy() for k, v in st.m.items()}
    assert m["pis"][1] == 0.0 and m["pis"][2] == 0.0
    if mode == 2:
        assert m["musX"][3, 0] == 0.0
    empty = torch.zeros_like(dl)
    got = {k: v.cpu().numpy()[None] for k, v in dp.items()}
    eng.update_kernel_list(dp, empty)
    want = o.shared_readmit(got, np.zeros((NB, K), bool), coords, cfg, np.float32)
    assert np.array_equal(_mask(empty.cpu().numpy().view(np.uint32), K), want) and not want[:, 1].any()
    eng.close()
    if mode == 3:
        # what fell outside the nudged ranges went to the extreme elements: the routing moved gradient mass
        q, back, _ = o.quantize_graph_params(p, cfg, np.float32)
        assert any(b["below"].any() or b["above"].any() for k, b in back.items() if "below" in b)


@pytest.mark.parametrize("kw", [dict(quantization_mode=3, quantize_pis=True, **QKW),
                                dict(kernel_count_as_norm_l1=True, pis_l1=0.5, quantize_pis=True),
                                dict(quantization_mode=3, quantize_pis=True, kernel_count_as_norm_l1=True, pis_l1=0.5,
                                     train_inverse_cov=True, **QKW)])
def test_shared_image_wide_records_follow_the_fit(kw):
    """quantization_mode 3 (image-wide min / max ranges, gradient routing between the kernels) and
    kernel_count_as_norm_l1 (pis_l1 / count(qpis > 0), smoe.py:1022-1027) over a short fit: the records are rebuilt
    from the current parameters before every launch."""
    shape, bshape, C, kpd = (64, 64), (16, 16), 1, [4, 4]
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, False, perturb=False, lr_steer=1e-2, **kw)
    p["pis"][0, 5] = 0.0004            # off the lattice from the start: not counted, not in the ranges
    if kw.get("train_inverse_cov"):
        p["A_diagonal"] = p["A_diagonal"] ** 2
    n = 12
    p32, st32, i32 = o.shared_fit(p, tgt, coords, cfg, n, val_iter=6, dtype=np.float32)
    p64, _, i64 = o.shared_fit(p, tgt, coords, cfg, n, val_iter=6, dtype=np.float64)
    eng = _engine(shape, bshape, C, K, False, lr_steer=1e-2, **kw)
    dp = _dev(p)
    dl = eng.new_lists()
    st = eng.new_adam_state(dp)
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    hist = [float(eng.forward(T, dp, dl, want_recon=False)["loss"].mean())]
    for _ in range(2):
        eng.fit(T, dp, st, dl, 6)
        eng.update_kernel_list(dp, dl)
        hist.append(float(eng.forward(T, dp, dl, want_recon=False)["loss"].mean()))
    torch.cuda.synchronize()
    assert abs(hist[0] - i32["hist"]["loss"][0]) < 2e-6 * max(1.0, abs(hist[0]))
    assert np.allclose(hist, i32["hist"]["loss"], rtol=0.02), (hist, i32["hist"]["loss"])
    got = {k: v.cpu().numpy()[None] for k, v in dp.items()}
    for name in o.PARAM_NAMES:
        dev = np.median(np.abs(got[name] - p32[name]))
        floor = np.median(np.abs(p32[name] - p64[name]))
        assert dev <= 3 * floor + 1e-6, (name, dev, floor)
    eng.close()


@pytest.mark.parametrize("shape,bshape,C,kpd,yuv,ov", [((64, 64), (16, 16), 1, [4, 4], False, 0),
                                                       ((64, 96), (32, 32), 3, [3, 5], True, 0),
                                                       ((96, 64), (32, 64), 1, [6, 4], False, 0),
                                                       ((48, 40), (16, 8), 1, [12, 12], False, 3),
                                                       # 3-d batches: the 11x11x11 window (smoe.py:999-1003)
                                                       ((16, 16, 12), (8, 8, 6), 3, [2, 2, 2], True, 0),
                                                       ((16, 24, 8), (8, 8, 8), 1, [2, 3, 1], False, 0)])
def test_shared_ssim_loss(shape, bshape, C, kpd, yuv, ov):
    """ssim_opt in the shared-kernel mode: loss_pixel = 1 - SSIM of every batch (smoe.py:980-1011; with a halo the
    interior is cropped first, smoe.py:984-991), gradients accumulated over the batches."""
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, yuv, pis_l1=0.05, u_l1=0.001, ssim_opt=True)
    lists = np.ones((NB, K), bool)
    eng = _engine(shape, bshape, C, K, yuv, pis_l1=0.05, u_l1=0.001, ssim_opt=True, overlap=ov)
    dp = _dev(p)
    dl = eng.new_lists()
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    fw = eng.forward(T, dp, dl, want_recon=True, update_lists=False)
    torch.cuda.synchronize()
    recon = fw["recon"].cpu().numpy().transpose(0, 2, 1)
    f64 = o.forward(o._bcast(p, NB), tgt, coords, lists, cfg, None, np.float64, want_grads=True, q_override=recon)
    assert np.abs(fw["loss"].cpu().numpy() - f64["loss"]).max() < 2e-5
    g64 = {k: v.sum(axis=0) for k, v in f64["grads"].items()}
    st = eng.new_adam_state(dp)
    loss = torch.zeros(NB, device="cuda")
    eng.accumulate(T, dp, dl, loss_out=loss)
    eng.apply(dp, st)
    torch.cuda.synchronize()
    assert np.abs(loss.cpu().numpy() - f64["loss"]).max() < 2e-5
    bad = (np.abs(f64["w"] - 0.5 / 256) < 1e-6).any() or ((np.abs(f64["y"]) < 1e-6) | (np.abs(f64["y"] - 1) < 1e-6)).any()
    for name in o.PARAM_NAMES:
        scale = np.abs(g64[name]).max() + 1e-30
        err = np.abs(st.m[name].cpu().numpy() / 0.1 - g64[name]).max() / scale
        assert err < (2e-3 if bad else 1e-4), (name, err)
    eng.close()


@pytest.mark.parametrize("mode", [2, 3])
def test_shared_ssim_loss_on_fake_quantised_variables(mode):
    """ssim_opt together with quantization_mode 2 / 3 in the shared-kernel mode: the variables are quantised at the
    kernels' load sites, so the SSIM instantiation sees them like the margin-loss one."""
    shape, bshape, C, kpd, yuv = (64, 96), (32, 32), 3, [3, 5], True
    kw = dict(quantization_mode=mode, quantize_pis=True, ssim_opt=True, **QKW)
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, yuv, **kw)
    p["A_corr"] = p["A_corr"] * np.tril(np.ones((2, 2), np.float32), -1)
    lists = np.ones((NB, K), bool)
    eng = _engine(shape, bshape, C, K, yuv, **kw)
    dp = _dev(p)
    dl = eng.new_lists()
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    fw = eng.forward(T, dp, dl, want_recon=True, update_lists=False)
    torch.cuda.synchronize()
    recon = fw["recon"].cpu().numpy().transpose(0, 2, 1)
    f64 = o.forward(o._bcast(p, NB), tgt, coords, lists, cfg, None, np.float64, want_grads=True, q_override=recon)
    assert np.abs(fw["loss"].cpu().numpy() - f64["loss"]).max() < 3e-5
    g64 = {k: v.sum(axis=0) for k, v in f64["grads"].items()}
    st = eng.new_adam_state(dp)
    eng.accumulate(T, dp, dl)
    eng.apply(dp, st)
    torch.cuda.synchronize()
    bad = (np.abs(f64["w"] - 0.5 / 256) < 1e-6).any() or ((np.abs(f64["y"]) < 1e-6) | (np.abs(f64["y"] - 1) < 1e-6)).any()
    for name in o.PARAM_NAMES:
        scale = np.abs(g64[name]).max() + 1e-30
        err = np.abs(st.m[name].cpu().numpy() / 0.1 - g64[name]).max() / scale
        assert err < (2e-3 if bad else 2e-4), (name, err)
    eng.close()


@pytest.mark.parametrize("shape,bshape,C,kpd,yuv,ov", [((64, 64), (16, 16), 1, [4, 4], False, 2),
                                                       ((64, 96), (32, 32), 3, [3, 5], True, 0),
                                                       ((32, 32, 8), (16, 16, 4), 3, [2, 2, 2], True, 0)])
def test_shared_inverse_covariance_form(shape, bshape, C, kpd, yuv, ov):
    """train_inverse_cov in the shared-kernel mode (smoe.py:734-735,791-793): forward, kernel lists (with a halo),
    accumulated gradients, readmission."""
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, yuv, pis_l1=0.05, u_l1=0.001, train_inverse_cov=True)
    p["A_diagonal"] = (p["A_diagonal"] ** 2).astype(np.float32)
    p["A_corr"] = (p["A_corr"] * 3.0).astype(np.float32)
    lists = np.ones((NB, K), bool)
    halo = o.global_halo_coords(shape, bshape, ov) if ov else None
    ref = o.shared_pass(p, tgt, coords, lists, cfg, np.float32, halo_coords=halo)
    eng = _engine(shape, bshape, C, K, yuv, pis_l1=0.05, u_l1=0.001, train_inverse_cov=True, overlap=ov)
    dp = _dev(p)
    dl = eng.new_lists()
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    fw = eng.forward(T, dp, dl, want_recon=True)
    torch.cuda.synchronize()
    recon = fw["recon"].cpu().numpy().transpose(0, 2, 1)
    f64 = o.forward(o._bcast(p, NB), tgt, coords, lists, cfg, None, np.float64, want_grads=True, q_override=recon)
    assert np.allclose(fw["loss"].cpu().numpy(), f64["loss"], rtol=3e-5)
    near_tau = (np.abs(f64["w"] - 0.5 / 256) < 1e-6).any(axis=2)
    if not ov:
        assert (_mask(dl.cpu().numpy().view(np.uint32), K) == ref["lists_new"])[~near_tau].all()
    else:
        assert (_mask(dl.cpu().numpy().view(np.uint32), K) == ref["lists_new"]).mean() > 0.995
    g64 = {k: v.sum(axis=0) for k, v in f64["grads"].items()}
    st = eng.new_adam_state(dp)
    full = eng.new_lists()
    eng.accumulate(T, dp, full)
    eng.apply(dp, st)
    torch.cuda.synchronize()
    bad = near_tau.any() or ((np.abs(f64["y"]) < 1e-6) | (np.abs(f64["y"] - 1) < 1e-6)).any()
    for name in o.PARAM_NAMES:
        scale = np.abs(g64[name]).max() + 1e-30
        err = np.abs(st.m[name].cpu().numpy() / 0.1 - g64[name]).max() / scale
        assert err < (2e-3 if bad else 5e-5), (name, err)
    got = {k: v.cpu().numpy()[None] for k, v in dp.items()}
    empty = torch.zeros_like(dl)
    eng.update_kernel_list(dp, empty)
    want = o.shared_readmit(got, np.zeros((NB, K), bool), coords if halo is None else halo, cfg, np.float32)
    assert np.array_equal(_mask(empty.cpu().numpy().view(np.uint32), K), want)
    eng.close()


@pytest.mark.parametrize("ic", [False, True])
def test_shared_radial_steering(ic):
    """radial_as in the shared-kernel mode: tied diagonals (trace gradient, u_l1 counted d times), A_corr untouched."""
    shape, bshape, C, kpd = (64, 64), (16, 16), 1, [4, 4]
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, False, perturb=True, pis_l1=0.05, u_l1=0.002,
                                             radial_as=True, train_inverse_cov=ic)
    a0 = np.abs(p["A_diagonal"][0, :, 0, 0]) ** (2 if ic else 1)
    p["A_diagonal"] = (a0[None, :, None, None] * np.eye(2)).astype(np.float32)
    p["A_corr"] = np.zeros_like(p["A_corr"])
    lists = np.ones((NB, K), bool)
    eng = _engine(shape, bshape, C, K, False, pis_l1=0.05, u_l1=0.002, radial_as=True, train_inverse_cov=ic)
    dp = _dev(p)
    dl = eng.new_lists()
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    recon = eng.forward(T, dp, dl, want_recon=True, update_lists=False)["recon"].cpu().numpy().transpose(0, 2, 1)
    f64 = o.forward(o._bcast(p, NB), tgt, coords, lists, cfg, None, np.float64, want_grads=True, q_override=recon)
    g64 = {k: v.sum(axis=0) for k, v in f64["grads"].items()}
    st = eng.new_adam_state(dp)
    eng.accumulate(T, dp, dl)
    eng.apply(dp, st)
    torch.cuda.synchronize()
    m = {k: v.cpu().numpy() for k, v in st.m.items()}
    scale = np.abs(g64["A_diagonal"]).max()
    assert np.abs(m["A_diagonal"] / 0.1 - g64["A_diagonal"]).max() / scale < 1e-4
    assert not m["A_corr"].any() and not dp["A_corr"].cpu().numpy().any()
    dg = np.diagonal(dp["A_diagonal"].cpu().numpy(), axis1=-2, axis2=-1)
    assert np.all(dg[:, 0] == dg[:, 1]) and np.abs(dg[:, 0] - a0).max() > 0.5
    eng.close()


def test_shared_radial_steering_with_fixed_range_quantisation():
    """radial_as together with quantization_mode 2 in the shared-kernel mode."""
    shape, bshape, C, kpd = (64, 64), (16, 16), 1, [4, 4]
    kw = dict(pis_l1=0.05, u_l1=0.002, radial_as=True, quantization_mode=2, quantize_pis=True, **QKW)
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, False, perturb=True, **kw)
    a0 = np.abs(p["A_diagonal"][0, :, 0, 0])
    a0[5] = 75.0                                       # clamped by the fixed range: no gradient
    p["A_diagonal"] = (a0[None, :, None, None] * np.eye(2)).astype(np.float32)
    p["A_corr"] = np.zeros_like(p["A_corr"])
    lists = np.ones((NB, K), bool)
    eng = _engine(shape, bshape, C, K, False, **kw)
    dp = _dev(p)
    dl = eng.new_lists()
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    recon = eng.forward(T, dp, dl, want_recon=True, update_lists=False)["recon"].cpu().numpy().transpose(0, 2, 1)
    f32 = o.forward(o._bcast(p, NB), tgt, coords, lists, cfg, None, np.float32, want_grads=True, q_override=recon)
    g32 = {k: v.sum(axis=0) for k, v in f32["grads"].items()}
    st = eng.new_adam_state(dp)
    eng.accumulate(T, dp, dl)
    eng.apply(dp, st)
    torch.cuda.synchronize()
    m = {k: v.cpu().numpy() for k, v in st.m.items()}
    for name in ("A_diagonal", "musX", "nu_e", "pis"):
        scale = np.abs(g32[name]).max() + 1e-30
        assert np.abs(m[name] / 0.1 - g32[name]).max() / scale < 1e-4, name
    assert not m["A_corr"].any() and not m["A_diagonal"][5].any()
    dg = np.diagonal(dp["A_diagonal"].cpu().numpy(), axis1=-2, axis2=-1)
    assert np.all(dg[:, 0] == dg[:, 1])
    eng.close()


def test_shared_radial_steering_with_ranges_from_the_data():
    """radial_as together with quantization_mode 3 in the shared-kernel mode (smoe.py:498-504, unshifted input)."""
    shape, bshape, C, kpd = (64, 64), (16, 16), 1, [4, 4]
    kw = dict(pis_l1=0.05, u_l1=0.002, radial_as=True, quantization_mode=3, quantize_pis=True, bit_depths=(12, 10, 6, 10, 8))
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, False, perturb=True, **kw)
    a0 = np.abs(p["A_diagonal"][0, :, 0, 0])
    a0[3] = a0.min()                                   # two tied minima
    a0[7] = a0.min()
    p["A_diagonal"] = (a0[None, :, None, None] * np.eye(2)).astype(np.float32)
    p["A_corr"] = np.zeros_like(p["A_corr"])
    lists = np.ones((NB, K), bool)
    eng = _engine(shape, bshape, C, K, False, **kw)
    dp = _dev(p)
    dl = eng.new_lists()
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    recon = eng.forward(T, dp, dl, want_recon=True, update_lists=False)["recon"].cpu().numpy().transpose(0, 2, 1)
    # identical parameters in every batch: the (linear) fake-quant backward of the sum is the sum of the routed gradients
    f32 = o.forward(o._bcast(p, NB), tgt, coords, lists, cfg, None, np.float32, want_grads=True, q_override=recon)
    g32 = {k: v.sum(axis=0) for k, v in f32["grads"].items()}
    st = eng.new_adam_state(dp)
    eng.accumulate(T, dp, dl)
    eng.apply(dp, st)
    torch.cuda.synchronize()
    m = {k: v.cpu().numpy() for k, v in st.m.items()}
    for name in ("A_diagonal", "musX", "nu_e", "pis"):
        scale = np.abs(g32[name]).max() + 1e-30
        assert np.abs(m[name] / 0.1 - g32[name]).max() / scale < 1e-4, name
    assert not m["A_corr"].any()
    dg = np.diagonal(dp["A_diagonal"].cpu().numpy(), axis1=-2, axis2=-1)
    assert np.all(dg[:, 0] == dg[:, 1])
    eng.close()


@pytest.mark.parametrize("mode", [2, 3])
def test_shared_fake_quantised_centre_offsets(mode):
    """use_diff_center with quantization_mode 2 / 3 in the shared-kernel mode (smoe_shared_set_center_grid)."""
    shape, bshape, C, kpd = (64, 64), (16, 16), 1, [4, 4]
    kw = dict(pis_l1=0.05, u_l1=0.002, quantization_mode=mode, quantize_pis=True, bit_depths=(14, 10, 8, 10, 10),
              lower_bounds=(-60, -.03, -1, 0, -4), upper_bounds=(60, .04, 2, 2, 4))
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, False, perturb=True, **kw)
    d = len(shape)
    p["A_corr"] = p["A_corr"] * np.tril(np.ones((d, d), np.float32), -1)
    from steered_mixture_of_experts_amd import blocks as sblk
    grid = sblk.init_block_params(img[None], kpd, True, False)["musX"][0].astype(np.float32)
    rng = np.random.default_rng(9)
    off = rng.uniform(-0.025, 0.025, size=grid.shape).astype(np.float32)
    off[5, 0] = 0.05                                    # outside the fixed offset range of mode 2
    p["musX"] = (grid + off)[None].astype(np.float32)
    cfg = o.OracleConfig(**{**cfg.__dict__, "mus_grid": grid[None]})
    lists = np.ones((NB, K), bool)
    eng = _engine(shape, bshape, C, K, False, **kw)
    gdev = torch.from_numpy(grid).cuda()
    eng.set_center_grid(gdev)
    dp = _dev(p)
    dl = eng.new_lists()
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    fw = eng.forward(T, dp, dl, want_recon=True, update_lists=False)
    recon = fw["recon"].cpu().numpy().transpose(0, 2, 1)
    ref = o.shared_pass(p, tgt, coords, lists, cfg, np.float32)
    frac = (np.clip(ref["y"], 0, 1) * 255 + 0.5) % 1.0
    tie = (frac < 3e-4) | (frac > 1 - 3e-4)
    assert (np.abs(recon - ref["recon"])[~tie] < 1e-7).all()
    f32 = o.forward(o._bcast(p, NB), tgt, coords, lists, cfg, None, np.float32, want_grads=True, q_override=recon)
    g32 = {k: v.sum(axis=0) for k, v in f32["grads"].items()}
    st = eng.new_adam_state(dp)
    eng.accumulate(T, dp, dl)
    eng.apply(dp, st)
    torch.cuda.synchronize()
    m = {k: v.cpu().numpy() for k, v in st.m.items()}
    for name in o.PARAM_NAMES:
        scale = np.abs(g32[name]).max() + 1e-30
        assert np.abs(m[name] / 0.1 - g32[name]).max() / scale < 1e-4, name
    if mode == 2:
        assert not m["musX"][5, 0]
    empty = torch.zeros_like(dl)
    got = {k: v.cpu().numpy()[None] for k, v in dp.items()}
    eng.update_kernel_list(dp, empty)
    want = o.shared_readmit(got, np.zeros((NB, K), bool), coords, cfg, np.float32)
    assert np.array_equal(_mask(empty.cpu().numpy().view(np.uint32), K), want)
    eng.close()


def test_gradient_buffer_goes_through_an_rccl_all_reduce():
    """Multi-GPU shared mode: the accumulated fp64 gradient buffer (library memory wrapped as a torch tensor) is what
    torch.distributed all-reduces between smoe_shared_accumulate and smoe_shared_apply.  One rank is all this box has:
    the collective must accept the tensor and leave a one-rank sum unchanged, and the step after it must equal the
    step without it."""
    import socket
    import torch.distributed as dist
    shape, bshape, C, kpd = (64, 64), (16, 16), 1, [4, 4]
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, False)
    eng = _engine(shape, bshape, C, K, False)
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    dp1, dp2 = _dev(p), _dev(p)
    l1, l2 = eng.new_lists(), eng.new_lists()
    s1, s2 = eng.new_adam_state(dp1), eng.new_adam_state(dp2)
    eng.accumulate(T, dp1, l1)
    eng.apply(dp1, s1)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        eng.accumulate(T, dp2, l2)
        buf = eng.grad_buffer()
        before = buf.clone()
        dist.all_reduce(buf)
        torch.cuda.synchronize()
        assert buf.dtype == torch.float64 and torch.equal(buf, before) and float(before.abs().max()) > 0
        eng.apply(dp2, s2)
        torch.cuda.synchronize()
        for k in dp1:
            assert torch.equal(dp1[k], dp2[k]), k
    finally:
        if own_group:
            dist.destroy_process_group()
    eng.close()


def test_shared_loss_weights():
    """smoe_shared_set_loss_weights: per-pixel weights on the margin loss and its gradients, indexed by the global
    batch id (so that a range of batches -- a rank's shard -- sees its own weights)."""
    shape, bshape, C, kpd = (64, 96), (32, 32), 3, [3, 5]
    img, p, cfg, coords, tgt, K, NB = _setup(shape, bshape, C, kpd, True, pis_l1=0.05, u_l1=0.001)
    lw = np.random.default_rng(3).uniform(0, 1, size=(NB, tgt.shape[1])).astype(np.float32)
    lw[1] = 0.0
    lists = np.ones((NB, K), bool)
    eng = _engine(shape, bshape, C, K, True, pis_l1=0.05, u_l1=0.001)
    LW = torch.from_numpy(lw).cuda()
    eng.set_loss_weights(LW)
    dp = _dev(p)
    dl = eng.new_lists()
    T = torch.from_numpy(blk.to_planar(tgt.reshape((NB,) + tuple(bshape) + (C,)))).cuda()
    fw = eng.forward(T, dp, dl, want_recon=True, update_lists=False)
    recon = fw["recon"].cpu().numpy().transpose(0, 2, 1)
    f64 = o.forward(o._bcast(p, NB), tgt, coords, lists, cfg, lw, np.float64, want_grads=True, q_override=recon)
    assert np.allclose(fw["loss"].cpu().numpy(), f64["loss"], rtol=3e-5, atol=1e-9)
    g64 = {k: v.sum(axis=0) for k, v in f64["grads"].items()}
    st = eng.new_adam_state(dp)
    h = NB // 2
    eng.accumulate(T[:h].contiguous(), dp, dl[:h], first_batch=0)             # two ranges: the weights follow the batch id
    eng.accumulate(T[h:].contiguous(), dp, dl[h:], first_batch=h)
    eng.apply(dp, st)
    torch.cuda.synchronize()
    bad = (np.abs(f64["w"] - 0.5 / 256) < 1e-6).any() or ((np.abs(f64["y"]) < 1e-6) | (np.abs(f64["y"] - 1) < 1e-6)).any()
    for name in o.PARAM_NAMES:
        scale = np.abs(g64[name]).max() + 1e-30
        err = np.abs(st.m[name].cpu().numpy() / 0.1 - g64[name]).max() / scale
        assert err < (2e-3 if bad else 5e-5), (name, err)
    eng.set_loss_weights(None)
    f1 = eng.forward(T, _dev(p), dl, want_recon=False, update_lists=False)     # same parameters as fw, weights cleared
    assert float(f1["loss"][1]) > float(fw["loss"][1])                         # batch 1 had weight 0: regularisers only
    eng.close()
