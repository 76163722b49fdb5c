"""Host logic and oracle against golden vectors PRODUCED BY THE REFERENCE'S OWN CODE
(tests/golden/ref_*.npz, generator tests/golden/gen_reference_fixtures.py: the reference's pure-numpy functions compiled
unchanged from its sources and run in the build container).  Pins SURVEY 8(a) rows a1-a4 (domain, windows, kernel grid,
expert / prior initialisation, default block shape), the parameter quantiser (f-3, quantizer.py) and plotter.psnr.
The TensorFlow graph itself stays unpinned."""
import os
import types

import numpy as np
import pytest

from oracle import smoe_oracle as o
from oracle.quantizer_oracle import quantize_block
from steered_mixture_of_experts_amd import blocks as blk
from steered_mixture_of_experts_amd import quantizer, utils

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INIT = np.load(os.path.join(GOLD, "ref_init.npz"))
WIN = np.load(os.path.join(GOLD, "ref_windows.npz"))
QNT = np.load(os.path.join(GOLD, "ref_quantizer.npz"))


@pytest.mark.parametrize("name", ["g2", "g34", "rgb", "vid", "b16", "odd"])
def test_initialisation_equals_the_reference(name):
    """gen_domain / generate_kernel_grid / generate_experts / generate_pis (smoe.py:2146-2242,2395-2426)."""
    img = INIT[f"{name}.image"]
    kpd = [int(k) for k in INIT[f"{name}.kpd"]]
    norm, ic = bool(INIT[f"{name}.normalize_pis"]), bool(INIT[f"{name}.train_inverse_cov"])
    d, C = img.ndim - 1, img.shape[-1]
    joint = INIT[f"{name}.joint_domain"]
    assert joint.shape == img.shape[:d] + (d + C,)
    # pixel coordinates: the fp32 values fed to the graph (smoe.py:1677 casts the feed to float32)
    want = joint[..., :d].reshape(-1, d).astype(np.float32)
    assert np.array_equal(o.block_coords(img.shape[:d]), want)
    assert np.array_equal(o.global_batch_coords(img.shape[:d], img.shape[:d])[0], want)
    assert np.array_equal(joint[..., d:], img)
    # kernel grid, steering, experts, priors: host code ...
    p = blk.init_block_params(img[None].astype(np.float32), kpd, norm, ic)
    K = INIT[f"{name}.musX_init"].shape[0]
    assert np.array_equal(p["musX"][0], INIT[f"{name}.musX_init"].astype(np.float32))
    assert np.array_equal(p["A_diagonal"][0], INIT[f"{name}.A_init"].astype(np.float32))
    assert not p["A_corr"].any() and p["A_corr"].shape == (1, K, d, d)
    assert np.array_equal(p["gamma_e"][0], INIT[f"{name}.gamma_e_init"].astype(np.float32))
    assert np.array_equal(p["pis"][0], INIT[f"{name}.pis_init"])
    assert INIT[f"{name}.nu_e_init"].dtype == np.float32
    assert np.allclose(p["nu_e"][0], INIT[f"{name}.nu_e_init"], rtol=0, atol=1e-6)     # fp32 mean of an fp32 vs fp64 image
    # ... and the oracle's own initialiser
    po = o.init_params(img[None], kpd, norm) if not ic else None
    if po is not None:
        assert np.array_equal(po["musX"][0], p["musX"][0]) and np.array_equal(po["A_diagonal"][0], p["A_diagonal"][0])
        assert np.allclose(po["nu_e"][0], INIT[f"{name}.nu_e_init"], rtol=0, atol=1e-6)
        assert np.array_equal(po["pis"][0], INIT[f"{name}.pis_init"])
    assert np.array_equal(o.kernel_grid(kpd, d), INIT[f"{name}.musX_init"])


@pytest.mark.parametrize("name", ["img", "halo", "vid", "vidhalo"])
def test_windows_equal_the_reference_sliding_window(name):
    """sliding_window (smoe.py:18-35): block order, block content, and with a halo the zero-padded window coordinates
    the shared-kernel mode feeds to the influence test."""
    joint = WIN[f"{name}.joint_domain"]
    ov, bs = int(WIN[f"{name}.overlap"]), tuple(int(b) for b in WIN[f"{name}.batch"])
    d = len(bs)
    shape = joint.shape[:d]
    coords, wins = WIN[f"{name}.coords"], WIN[f"{name}.windows"]
    grid = [s // b for s, b in zip(shape, bs)]
    # origins: row-major over the block grid, minus the halo
    want = np.stack(np.meshgrid(*[np.arange(g) * b for g, b in zip(grid, bs)], indexing="ij"), axis=-1).reshape(-1, d) - ov
    assert np.array_equal(coords, want)
    interior = wins[(slice(None),) + tuple(slice(ov, ov + b) for b in bs)]
    blocks, _ = blk.image_to_blocks(joint, bs)
    assert np.array_equal(blocks, interior)
    assert np.array_equal(blk.blocks_to_image(blocks, shape, bs), joint)
    assert np.array_equal(o.global_batch_coords(shape, bs), interior[..., :d].reshape(len(wins), -1, d).astype(np.float32))
    if ov > 0:
        halo = o.global_halo_coords(shape, bs, ov)
        assert np.array_equal(halo, wins[..., :d].reshape(len(wins), -1, d).astype(np.float32))


def test_default_block_shape_and_intervals_equal_the_reference():
    """Smoe.get_batch_shape (smoe.py:2459-2543) incl. its enumeration-order tie break."""
    for si, s in enumerate(WIN["gbs.shapes"]):
        shape = tuple(int(v) for v in s if v > 0)
        for wi, w in enumerate(WIN["gbs.want"]):
            want = tuple(int(v) for v in WIN["gbs.result"][si, wi][:len(shape)])
            assert blk.get_batch_shape(int(w), shape) == want, (shape, int(w))


def _cases():
    return range(int(QNT["ncases"]))


@pytest.mark.parametrize("i", list(range(15)))
def test_quantizer_equals_the_reference(i):
    """quantize_params + rescaler (quantizer.py:4-144, utils.reduce_params) on one model: the vectorised host quantiser
    and the oracle restatement both reproduce the reference's integers and rescaled values."""
    assert int(QNT["ncases"]) == 15
    tag = f"q{i}"
    mode, qpis = int(QNT[f"{tag}.mode"]), bool(QNT[f"{tag}.quantize_pis"])
    names = ("pis", "musX", "A_diagonal", "A_corr", "nu_e", "gamma_e")
    p = {k: QNT[f"{tag}.in.{k}"] for k in names}
    keep = p["pis"] > 0
    lb, ub, bd = [-2500, -.3, -5, 0, -32], [2500, 1.3, 5, 2, 32], [20, 18, 6, 10, 10]
    smoe = types.SimpleNamespace(quantization_mode=mode, quantize_pis=qpis, radial_as=False, bit_depths=bd,
                                 lower_bounds=lb, upper_bounds=ub, use_diff_center=False)
    q = quantizer.quantize_params(smoe, {k: v[None] for k, v in p.items()})
    r = quantizer.rescaler(smoe, q)
    assert np.array_equal(q["used_kernels"][0], keep)
    for k in names:
        assert np.array_equal(q[k][0][keep], QNT[f"{tag}.q.{k}"]), k
        assert np.array_equal(np.broadcast_to(q["lower_bounds"][k][0], QNT[f"{tag}.lb.{k}"].shape), QNT[f"{tag}.lb.{k}"]), k
        assert np.array_equal(np.broadcast_to(q["upper_bounds"][k][0], QNT[f"{tag}.ub.{k}"].shape), QNT[f"{tag}.ub.{k}"]), k
    for k in ("A", "musX", "nu_e", "pis", "gamma_e"):
        assert q["steps"][k] == int(QNT[f"{tag}.steps.{k}"])
        assert np.array_equal(r[k][0][keep], QNT[f"{tag}.r.{k}"]), k
    assert not r["pis"][0][~keep].any()                         # dropped kernels stay absent
    idx, qo, ro = quantize_block(p, bd, mode, qpis, lb, ub)
    assert np.array_equal(idx, keep)
    for k in names:
        assert np.array_equal(qo[k], QNT[f"{tag}.q.{k}"]), k
    for k in ("A", "musX", "nu_e", "pis", "gamma_e"):
        assert np.array_equal(ro[k], QNT[f"{tag}.r.{k}"]), k


@pytest.mark.parametrize("i", list(range(6)))
def test_radial_quantizer_equals_the_reference(i):
    """radial_as through quantize_params + rescaler: one steering value per kernel, no A_corr, A = a * I back."""
    assert int(QNT["nradial"]) == 6
    tag = f"rq{i}"
    mode, qpis = int(QNT[f"{tag}.mode"]), bool(QNT[f"{tag}.quantize_pis"])
    names = ("pis", "musX", "A_diagonal", "A_corr", "nu_e", "gamma_e")
    p = {k: QNT[f"{tag}.in.{k}"] for k in names}
    keep = p["pis"] > 0
    smoe = types.SimpleNamespace(quantization_mode=mode, quantize_pis=qpis, radial_as=True, bit_depths=[20, 18, 6, 10, 10],
                                 lower_bounds=[-2500, -.3, -5, 0, -32], upper_bounds=[2500, 1.3, 5, 2, 32],
                                 use_diff_center=False)
    q = quantizer.quantize_params(smoe, {k: v[None] for k, v in p.items()})
    r = quantizer.rescaler(smoe, q)
    assert "A_corr" not in q
    for k in ("A_diagonal", "musX", "nu_e", "pis", "gamma_e"):
        assert np.array_equal(q[k][0][keep], QNT[f"{tag}.q.{k}"]), k
    for k in ("A", "musX", "nu_e", "pis", "gamma_e"):
        assert np.array_equal(r[k][0][keep], QNT[f"{tag}.r.{k}"]), k
    assert np.array_equal(r["A_diagonal"][0][keep], QNT[f"{tag}.r.A"]) and not r["A_corr"].any()


def test_psnr_equals_the_reference():
    for prec in (8, 10):
        want = QNT[f"psnr.p{prec}"]
        assert np.array_equal(utils.psnr(QNT["psnr.mse"], prec), want)
        assert np.array_equal(blk.psnr(QNT["psnr.mse"], prec), want)


def test_cli_defaults_equal_the_reference():
    """Every flag of the reference's training CLI (smoe_test.py:262-352) exists with the same option strings and the
    same default; the one documented deviation is inc_steps (kernel adding is not built)."""
    import json
    from steered_mixture_of_experts_amd.smoe_test import build_parser
    ref = json.load(open(os.path.join(GOLD, "ref_cli_defaults.json")))
    mine = {a.dest: a for a in build_parser()._actions}
    assert len(ref) == 47
    for dest, want in ref.items():
        assert dest in mine, dest
        assert set(mine[dest].option_strings) == set(want["flags"]), dest
        if dest == "inc_steps":
            assert (mine[dest].default, want["default"]) == (0, 100)
        elif "default" in want:
            assert mine[dest].default == want["default"], dest
        assert bool(mine[dest].required) == bool(want.get("required", False)), dest


def test_checkpoint_written_by_the_reference_decodes(tmp_path):
    """A pickle written by the reference's own utils.save_model (reduce=True: only the kernels with pis > 0; kernels
    leading, ONE model for the image) is read by load_params and decoded by the reconstruction entry point."""
    from fake_engine import OracleSharedEngine
    _decode_reference_checkpoint(tmp_path, OracleSharedEngine)


@pytest.mark.gpu
def test_checkpoint_written_by_the_reference_decodes_on_the_gpu(tmp_path):
    """The same decode through the HIP shared-kernel engine (smoe_shared_forward)."""
    _decode_reference_checkpoint(tmp_path, None)


def test_checkpoint_writer_keeps_the_reference_schema(tmp_path):
    """utils.save_model of this package writes every key the reference's save_model writes (utils.py:18-59)."""
    import pickle
    from fake_engine import OracleEngine
    from steered_mixture_of_experts_amd.smoe import Adam, Smoe
    ref = pickle.load(open(os.path.join(GOLD, "ref_checkpoint.pkl"), "rb"))
    img = np.load(os.path.join(GOLD, "ref_checkpoint_inputs.npz"))["image"][:16, :32, :1]
    s = Smoe(img, train_inverse_cov=False, kernels_per_dim=[2, 2], batch_size=[16, 16], use_determinant=True,
             quantization_mode=1, engine_factory=OracleEngine)
    s.set_optimizer(Adam(1e-3), Adam(1e-5), Adam(0.01))
    s.train(2, val_iter=1)
    mine = utils.save_model(s, str(tmp_path / "m.pkl"), quantize=True)
    assert set(ref) <= set(mine), set(ref) - set(mine)
    assert set(ref["qparams"]) <= set(mine["qparams"]) | {"used_kernels"}, set(ref["qparams"]) - set(mine["qparams"])
    assert set(ref["params"]) == set(mine["params"])


def _decode_reference_checkpoint(tmp_path, factory):
    import pickle
    import steered_mixture_of_experts_amd.smoe_reconstruction as rec
    path = os.path.join(GOLD, "ref_checkpoint.pkl")
    inp = np.load(os.path.join(GOLD, "ref_checkpoint_inputs.npz"))
    cp = pickle.load(open(path, "rb"))
    assert {"params", "mses", "losses", "num_pis", "quantization_mode", "quantized_pis", "lower_bounds", "upper_bounds",
            "use_yuv", "only_y_gamma", "ssim_opt", "use_determinant", "use_diff_center", "qparams"} == set(cp)
    p = utils.load_params(path)
    keep = inp["p.pis"] > 0
    assert keep.sum() == 10 and p["pis"].shape == (10,)
    for k in ("pis", "musX", "A_diagonal", "A_corr", "nu_e", "gamma_e"):
        assert np.array_equal(p[k], inp["p." + k][keep]), k
    assert np.array_equal(cp["qparams"]["used_kernels"], keep)
    # the writer of this package produces the same top-level schema (plus its own extra keys)
    img = inp["image"]
    np.save(tmp_path / "img.npy", np.uint8(np.round(img * 255)))
    rec._shared_engine_factory = factory
    try:
        out = str(tmp_path / "out")
        recon, loss, mse = rec.main(str(tmp_path / "img.npy"), out, path)
    finally:
        rec._shared_engine_factory = None
    # oracle: the whole image as ONE block with these kernels (quantize_pis as recorded in the checkpoint)
    tgt = (np.uint8(np.round(img * 255)).astype(np.float32) / 255.0).reshape(1, -1, 3)
    cfg = o.OracleConfig(block_shape=(32, 48), channels=3, kernels=10, use_yuv=True, use_determinant=True,
                         quantize_pis=True)
    P = {k: v[None].astype(np.float32) for k, v in p.items()}
    f = o.forward(P, tgt, o.block_coords((32, 48))[None], np.ones((1, 10), bool), cfg, None, np.float64)
    want = f["recon"].reshape(32, 48, 3)
    frac = (np.clip(f["y"], 0, 1) * 255 + 0.5) % 1.0
    tie = ((frac < 1e-3) | (frac > 1 - 1e-3)).reshape(32, 48, 3)
    assert recon.shape == want.shape and np.abs(recon - want)[~tie].max() < 1e-6
    assert os.path.exists(out + "/0_reconstruction.npy") or any(n.endswith("_reconstruction.npy") for n in os.listdir(out))
