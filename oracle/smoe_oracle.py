"""CPU restatement (numpy) of the reference's per-block SMoE hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the package
``steered_mixture_of_experts_amd``) may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only
as the checker / the timed CPU baseline.

PARITY UNPINNED for the graph arithmetic.  The reference
(roljon/Steered-Mixture-of-Experts) ships no tests, fixtures or golden vectors,
and its arithmetic lives in TensorFlow 1.x (version unpinned, not vendored, not
installable here), so the forward / gradient / Adam restatement cannot be checked
against outputs of the reference itself.  (The host-side pieces -- coordinates,
windows, initialisers -- ARE pinned bit-exactly against vectors produced by the
reference's own numpy code: tests/test_reference_golden.py.)  It is written from
the reference's source text, function by function (citations below are
``file:line`` under ``/root/reference``), and is cross-checked by
``tests/test_oracle.py`` against (i) ``torch.autograd`` on an independent
forward, (ii) central finite differences, (iii) hand known-answer cases that
follow from the reference's initialisers, and (iv) the plain-C restatement in
``oracle/smoe_oracle.c``.

Semantics: every image block is an independent ``Smoe`` instance (own [0,1]^d
domain, own K kernels, own Adam state); all arrays carry a leading block axis
``B``.  ``dtype`` selects fp64 (master) or fp32 (what TF computes in).

Third-party arithmetic restated here (TensorFlow 1.x, call sites in smoe.py):
  * ``tf.train.AdamOptimizer`` / ``ApplyAdam``:
        alpha = lr*sqrt(1-beta2^t)/(1-beta1^t); m += (g-m)(1-beta1);
        v += (g*g-v)(1-beta2); var -= m*alpha/(sqrt(v)+eps)
    with beta^t kept as a running fp32 product (smoe_test.py:84-88,
    smoe.py:1173-1193).
  * ``tf.quantization.fake_quant_with_min_max_args(x, 0, 1, num_bits=p)``
    (smoe.py:899): scale = 1/(2^p-1), zero point 0, nudged range [0, (2^p-1)*scale],
    out = floor(clamp(x)*inv_scale + 0.5)*scale, gradient 1 inside the range.
  * ``tf.clip_by_value`` (smoe.py:857): gradient 1 on [0,1] inclusive.
  * ``tf.maximum(10e-12, s)`` (smoe.py:821): gradient to ``s`` iff s > 1e-11.
"""
from __future__ import annotations

import dataclasses
import itertools
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

PARAM_NAMES = ("pis", "musX", "A_diagonal", "A_corr", "gamma_e", "nu_e")


@dataclasses.dataclass
class OracleConfig:
    """Hyper-parameters of one fit; defaults = smoe_test.py CLI defaults with
    kernel adding off."""
    block_shape: Tuple[int, ...]          # (bh, bw) or (bh, bw, bt)
    channels: int
    kernels: int
    precision: int = 8                    # utils.py:126-131
    margin: float = 0.5                   # smoe.py:41
    use_determinant: bool = True          # smoe_test.py:293-294
    use_yuv: bool = False                 # smoe_test.py:41-44 (forced False unless C==3)
    train_pis: bool = True
    train_gammas: bool = True
    train_musx: bool = True
    lr_expert: float = 1e-3               # optimizer1 {nu_e, gamma_e, musX}  smoe_test.py:84
    lr_pis: float = 1e-5                  # optimizer2 {pis}                  smoe_test.py:85
    lr_steer: float = 1.0                 # optimizer3 {A_diagonal, A_corr}   smoe_test.py:86
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-8
    grad_clip: Optional[float] = None     # smoe.py:1152-1153
    pis_l1: float = 0.0                   # smoe.py:1027
    u_l1: float = 0.0                     # smoe.py:1044
    start_pis: Optional[int] = None       # smoe.py:264 (K0 of the l1 normaliser)
    only_y_gamma: bool = False            # smoe.py:725-729 (slopes only for channel 0)
    ssim_opt: bool = False                # smoe.py:929,980-1011: loss_pixel = 1 - SSIM (2-d blocks)
    # fake-quantised parameters inside the graph (smoe.py:474-538); order of the 5-tuples: A, musX, nu_e, pis, gamma_e
    train_inverse_cov: bool = False       # smoe.py:734-735,791-793: A symmetric, maha = r^T A r (the ctor default is
                                          # True, the CLI default False, smoe_test.py:342)
    kernel_count_as_norm_l1: bool = False  # smoe.py:1022-1027: the pis l1 term is normalised by count(qpis > 0), not start_pis
    radial_as: bool = False               # smoe.py:349-365,429-434,714-719: ONE steering value per kernel (A = a I),
                                          # A_corr not trainable.  Held here as A_diagonal with equal diagonal entries
    mus_grid: Optional[np.ndarray] = None  # use_diff_center (smoe.py:390-394,746-747): the kernel-grid centres (B,K,d) the
                                          # trained OFFSETS are relative to.  ``musX`` holds grid + offset everywhere (the
                                          # engines' convention); only the fake quantisation needs the offsets themselves
    quantization_mode: int = 0            # 0/1: none in the graph; 2: fixed ranges; 3: min/max of the model's kernels
    quantize_pis: bool = False            # smoe.py:474 (the reference CLI passes True by default, smoe_test.py:304)
    bit_depths: Tuple[int, ...] = (20, 18, 6, 10, 10)                 # smoe_test.py:302
    lower_bounds: Tuple[float, ...] = (-2500, -.3, -5, 0, -32)        # smoe_test.py:306
    upper_bounds: Tuple[float, ...] = (2500, 1.3, 5, 2, 32)           # smoe_test.py:308

    @property
    def dim(self) -> int:
        return len(self.block_shape)

    @property
    def pixels(self) -> int:
        return int(np.prod(self.block_shape))

    @property
    def k0(self) -> int:
        return self.kernels if self.start_pis is None else self.start_pis


# --------------------------------------------------------------------------
# domain / initialisers
# --------------------------------------------------------------------------
def axis_coords(block_shape: Sequence[int]):
    """Per-axis pixel coordinates, smoe.py:2412: ``np.linspace(0, 1, size)``."""
    return [np.linspace(0, 1, int(s)) for s in block_shape]


def block_coords(block_shape: Sequence[int], dtype=np.float32) -> np.ndarray:
    """(N, d) pixel coordinates of one block, 'ij' meshgrid flattened row-major
    (smoe.py:2418-2421, 1650); fed to TF as float32 (smoe.py:545)."""
    grids = np.meshgrid(*axis_coords(block_shape), indexing="ij")
    dom = np.stack(grids, axis=-1).reshape(-1, len(block_shape))
    return dom.astype(np.float32).astype(dtype)


def kernel_grid(kernels_per_dim: Sequence[int], dim: int) -> np.ndarray:
    """(K, d) kernel centres, smoe.py:2402-2415,2424: equal spacing to the
    border, ``linspace(1/(2n), 1-1/(2n), n)`` per axis."""
    kpd = list(kernels_per_dim)
    if len(kpd) == 1:
        kpd = kpd * dim
    coord = [np.linspace((1 / n) / 2, 1 - (1 / n) / 2, n) for n in kpd]
    grids = np.meshgrid(*coord, indexing="ij")
    return np.reshape(np.stack(grids, axis=-1), (int(np.prod(kpd)), dim))


def init_params(blocks: np.ndarray, kernels_per_dim: Sequence[int],
                normalize_pis: bool = True, train_inverse_cov: bool = False
                ) -> Dict[str, np.ndarray]:
    """Initial parameters of every block, as ``Smoe(block_b, kernels_per_dim)``
    would build them.  ``blocks``: (B, *block_shape, C) float32 in [0,1].

    smoe.py:2146-2163 (grid, A_init = diag(2*(k_i+1))), 2165-2235 (nu_e = mean
    of the pixels in the [mu-mu0, mu+mu0) window, Python ``round``),
    2237-2242 (pis = 1/K), 436-437 (A_corr = 0).  Variables are float32
    (smoe.py:388-396).
    """
    B = blocks.shape[0]
    shape = blocks.shape[1:-1]
    C = blocks.shape[-1]
    d = len(shape)
    kpd = list(kernels_per_dim)
    if len(kpd) == 1:
        kpd = kpd * d
    mus = kernel_grid(kpd, d)
    K = mus.shape[0]
    A_proto = np.diag([2.0 * (k + 1) for k in kpd])
    if train_inverse_cov:
        A_proto = A_proto ** 2
    stride = mus[0]
    nu = np.empty((B, K, C), dtype=np.float32)
    for k in range(K):
        sl = [slice(None)]
        for ax in range(d):
            lo = int(round((mus[k, ax] - stride[ax]) * shape[ax]))
            hi = int(round((mus[k, ax] + stride[ax]) * shape[ax]))
            sl.append(slice(lo, hi))
        nu[:, k, :] = np.mean(blocks[tuple(sl)], axis=tuple(range(1, d + 1)))
    pis = np.ones((K,), dtype=np.float32)
    if normalize_pis:
        pis = pis / K
    out = {
        "pis": np.tile(pis, (B, 1)),
        "musX": np.tile(mus.astype(np.float32), (B, 1, 1)),
        "A_diagonal": np.tile(A_proto.astype(np.float32), (B, K, 1, 1)),
        "A_corr": np.zeros((B, K, d, d), dtype=np.float32),
        "gamma_e": np.zeros((B, K, d, C), dtype=np.float32),
        "nu_e": nu,
    }
    return out


def zeros_like_params(p: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    return {k: np.zeros_like(v) for k, v in p.items()}


def new_adam_state(p: Dict[str, np.ndarray]) -> Dict[str, object]:
    """TF1 Adam slots: m, v per variable, beta powers start at beta (they are
    multiplied after each apply)."""
    return {"m": zeros_like_params(p), "v": zeros_like_params(p), "t": 0,
            "b1p": None, "b2p": None}


# --------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------
def _steering(p, T, symmetric=False):
    """A = diag(A_diagonal) + strict_lower(A_corr), smoe.py:732-733; train_inverse_cov adds the transposed
    strict lower part (smoe.py:734-735)."""
    Ad = p["A_diagonal"].astype(T)
    Ac = p["A_corr"].astype(T)
    d = Ad.shape[-1]
    eye = np.eye(d, dtype=bool)
    low = np.tril(np.ones((d, d), dtype=bool), -1)
    A = np.where(eye, Ad, 0) + np.where(low, Ac, 0)
    if symmetric:
        A = A + np.swapaxes(np.where(low, Ac, 0), -1, -2)
    return A


def _maha(r, A, inverse_cov):
    """smoe.py:791-796: r^T A r for train_inverse_cov, |A^T r|^2 otherwise.  Returns (maha, z) with z = A^T r."""
    z = np.einsum("...nl,...lm->...nm", r, A)                # z_m = sum_l r_l A[l,m]
    if inverse_cov:
        return np.sum(z * r, axis=-1), z
    return np.sum(z * z, axis=-1), z


def fake_quant01(y, precision, T):
    """clip (smoe.py:857) then fake_quant_with_min_max_args(0,1,bits)
    (smoe.py:899); TF kernel: floor(clamped*inv_scale+0.5)*scale in fp32."""
    levels = T(2 ** precision - 1)
    scale = T(1) / levels                       # (max-min)/(quant_max-quant_min)
    inv_scale = T(1) / scale
    nudged_max = levels * scale
    yc = np.minimum(np.maximum(y, T(0)), T(1))  # clip_by_value
    cl = np.minimum(np.maximum(yc, T(0)), nudged_max)
    return np.floor(cl * inv_scale + T(0.5)) * scale


# --------------------------------------------------------------------------
# fake-quantised parameters (smoe.py:474-538): TF fake_quant_with_min_max_{args,vars}
# --------------------------------------------------------------------------
def fq_nudge(mn, mx, bits, T):
    """TF Nudge() (fake_quant_ops_functor.h): scale and the nudged range of [mn, mx] for ``bits`` bits, in T.
    A zero-width range gives scale 0; callers treat min == max == 0 separately as TF does."""
    qmin, qmax = T(0), T(2 ** bits - 1)
    mn, mx = np.asarray(mn, dtype=T), np.asarray(mx, dtype=T)
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = (mx - mn) / (qmax - qmin)
        zp = qmin - mn / scale
        rounded = np.sign(zp) * np.floor(np.abs(zp) + T(0.5))             # std::round: half away from zero
        nzp = np.where(zp < qmin, qmin, np.where(zp > qmax, qmax, rounded)).astype(T)
        return scale, ((qmin - nzp) * scale).astype(T), ((qmax - nzp) * scale).astype(T)


def fq_apply(x, mn, mx, bits, T):
    """fake_quant forward: floor((clamp(x) - nudged_min) * (1/scale) + 0.5) * scale + nudged_min and the
    mask of elements inside the nudged range; min == max == 0 -> zeros (everything counts as inside)."""
    scale, nmin, nmax = fq_nudge(mn, mx, bits, T)
    zero = np.logical_and(np.asarray(mn) == 0, np.asarray(mx) == 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        cl = np.minimum(np.maximum(x, nmin), nmax)
        q = np.floor((cl - nmin) * (T(1) / scale) + T(0.5)) * scale + nmin
    inside = np.logical_and(x >= nmin, x <= nmax)
    return np.where(zero, T(0), q).astype(T), np.where(zero, True, inside)


def _fq_minmax_vars(x, sel, bits, offset, T, noshift=False):
    """Mode-3 tensor (smoe.py:497-530): range = min / max of x over the selected elements ``sel`` (bool, same
    shape as x) of each block.  ``offset``: the ``fake_quant(x - min, 0, max - min) + min`` form (A_diagonal,
    nu_e), else ``fake_quant(x, min, max)``.  Returns the q-tensor and the linear map of the backward pass
    as masks: g -> g*between + tie_lo * sum(g*below) + tie_hi * sum(g*above) -- fake_quant_with_min_max_vars
    sends what falls outside the NUDGED range to its min / max input, and reduce_min / reduce_max hand that
    to the extreme elements, split equally over ties.  (Nudging keeps 0 exactly representable: an all-positive
    range [lo, hi] becomes [0, hi - lo], so e.g. centres above hi - lo are clamped -- restated as is.)
    ``noshift`` (radial_as, smoe.py:498-504): ``fake_quant(x, 0, max - min) + min`` -- the input is NOT shifted by the
    minimum (restated as is).  Backward: the range's lower end is the constant 0 (what falls below it is lost), its upper
    end max - min hands sum(g*above) to the maximum and takes it from the minimum, and the added minimum collects
    sum(g): the minimum receives sum(g * ~above)."""
    ax = tuple(range(1, x.ndim))
    lo = np.where(sel, x, np.inf).min(axis=ax, keepdims=True)
    hi = np.where(sel, x, -np.inf).max(axis=ax, keepdims=True)
    none = ~sel.any(axis=ax, keepdims=True)
    lo = np.where(none, 0, lo).astype(T)
    hi = np.where(none, 0, hi).astype(T)
    if noshift:
        v, rmin, rmax, back = x, np.zeros_like(lo), hi - lo, lo
    elif offset:
        v, rmin, rmax, back = x - lo, np.zeros_like(lo), hi - lo, lo
    else:
        v, rmin, rmax, back = x, lo, hi, np.zeros_like(lo)
    scale, nmin, nmax = fq_nudge(rmin, rmax, bits, T)
    zero = np.logical_and(rmin == 0, rmax == 0)              # TF: zeros forward, identity backward
    with np.errstate(divide="ignore", invalid="ignore"):
        cl = np.minimum(np.maximum(v, nmin), nmax)
        q = np.floor((cl - nmin) * (T(1) / scale) + T(0.5)) * scale + nmin
    q = (np.where(zero, T(0), q) + back).astype(T)
    below = np.logical_and(~zero, v < nmin)
    above = np.logical_and(~zero, v > nmax)
    tie_lo = np.logical_and(sel, x == lo).astype(T)
    tie_hi = np.logical_and(sel, x == hi).astype(T)
    tie_lo = tie_lo / np.maximum(tie_lo.sum(axis=ax, keepdims=True), 1)
    tie_hi = tie_hi / np.maximum(tie_hi.sum(axis=ax, keepdims=True), 1)
    to_lo = ~above if noshift else below
    return q, {"between": ~(below | above), "below": to_lo, "above": above, "tie_lo": tie_lo, "tie_hi": tie_hi}


def quantize_graph_params(p, cfg: OracleConfig, T):
    """The q-tensors the graph is built on (smoe.py:474-538) and, per tensor, the backward map of the
    fake-quant ops (see ``route_quant_grads``).  Mode 3 ranges are the min/max over the kernels with
    qpis > 0 (pis_mask, NOT the kernel list)."""
    lb, ub, bd = cfg.lower_bounds, cfg.upper_bounds, cfg.bit_depths
    q = {k: p[k].astype(T) for k in PARAM_NAMES}
    back = {}
    mode = cfg.quantization_mode
    if mode >= 2 or cfg.quantize_pis:
        q["pis"], inside = fq_apply(q["pis"], T(lb[3]), T(ub[3]), bd[3], T)
        back["pis"] = {"between": inside}
    keep = q["pis"] > 0
    grid = None
    if mode >= 2 and cfg.mus_grid is not None:              # smoe.py:746-747: musX = qmusX + musX_grid, qmusX = fq(offset)
        grid = np.broadcast_to(np.asarray(cfg.mus_grid).astype(T), q["musX"].shape)
        q["musX"] = q["musX"] - grid
    if mode == 2:
        for name, i in (("A_diagonal", 0), ("A_corr", 0), ("musX", 1), ("nu_e", 2), ("gamma_e", 4)):
            q[name], inside = fq_apply(q[name], T(lb[i]), T(ub[i]), bd[i], T)
            back[name] = {"between": inside}
    elif mode == 3:
        def sel_all(x):
            return np.broadcast_to(keep.reshape(keep.shape + (1,) * (x.ndim - 2)), x.shape)
        d = q["A_diagonal"].shape[-1]
        diag_sel = np.logical_and(sel_all(q["A_diagonal"]), np.eye(d, dtype=bool))
        # radial_as: the variable is ONE value per kernel (here tiled over the diagonal: sums and tie counts both come
        # out d times the reference's, their quotients equal) and the reference does not shift it (smoe.py:498-504)
        q["A_diagonal"], back["A_diagonal"] = _fq_minmax_vars(q["A_diagonal"], diag_sel, bd[0], True, T, noshift=cfg.radial_as)
        q["A_corr"], back["A_corr"] = _fq_minmax_vars(q["A_corr"], sel_all(q["A_corr"]), bd[0], False, T)
        if cfg.train_musx:
            q["musX"], back["musX"] = _fq_minmax_vars(q["musX"], sel_all(q["musX"]), bd[1], False, T)
        q["nu_e"], back["nu_e"] = _fq_minmax_vars(q["nu_e"], sel_all(q["nu_e"]), bd[2], True, T)
        q["gamma_e"], back["gamma_e"] = _fq_minmax_vars(q["gamma_e"], sel_all(q["gamma_e"]), bd[4], False, T)
    if grid is not None:
        q["musX"] = (q["musX"] + grid).astype(T)
    return q, back, keep


def route_quant_grads(grads, back, T):
    """Gradients w.r.t. the q-tensors -> gradients w.r.t. the variables (registered gradients of
    fake_quant_with_min_max_args / _vars composed with reduce_min / reduce_max)."""
    out = {}
    for k, g in grads.items():
        b = back.get(k)
        if b is None:
            out[k] = g
            continue
        r = np.where(b["between"], g, T(0))
        if "below" in b:
            ax = tuple(range(1, g.ndim))
            r = r + b["tie_lo"] * np.sum(np.where(b["below"], g, T(0)), axis=ax, keepdims=True) \
                  + b["tie_hi"] * np.sum(np.where(b["above"], g, T(0)), axis=ax, keepdims=True)
        out[k] = r.astype(T)
    return out


# --------------------------------------------------------------------------
# SSIM loss (smoe.py:980-1011 -> ops/image_ops_impl.py:77-233), 2-d and 3-d blocks
# --------------------------------------------------------------------------
SSIM_SIZE, SSIM_SIGMA, SSIM_PAD = 11, 1.5, 5                  # image_ops_impl.py:180-181; smoe.py:994
SSIM_C1, SSIM_C2 = 0.01 ** 2, 0.03 ** 2                       # image_ops_impl.py:74-75,110-111 (max_val = 1)


def ssim_window(T=np.float32, ndim=2):
    """_fspecial_gauss (image_ops_impl.py:132-149): softmax over the 11^ndim grid of -0.5 |offset|^2 / sigma^2."""
    c = np.arange(SSIM_SIZE, dtype=T) - T(SSIM_SIZE - 1) / T(2)
    g = np.square(c) * T(-0.5 / SSIM_SIGMA ** 2)
    if ndim == 2:
        gn = g[None, :] + g[:, None]
    else:
        gn = g[None, None, :] + g[None, :, None] + g[:, None, None]
    e = np.exp(gn - gn.max())
    return (e / e.sum()).astype(T)


def _ssim_reduce(img, win):
    """reducer (image_ops_impl.py:203-219): VALID correlation of (B, *spatial) with the window (depthwise_conv2d per
    channel for images, conv3d per channel for volumes)."""
    nd = win.ndim
    v = np.lib.stride_tricks.sliding_window_view(img, (SSIM_SIZE,) * nd, axis=tuple(range(1, nd + 1)))
    if nd == 2:
        return np.einsum("bijuv,uv->bij", v, win)
    return np.einsum("bijkuvw,uvw->bijk", v, win)


def ssim_and_grad(q, t, block_shape, T=np.float32, want_grad=False):
    """custom_ssim on SYMMETRIC-padded blocks (smoe.py:993-1003): q, t (B,N,C) -> ssim (B,C) per channel
    = mean over the window positions (one per block pixel) of luminance * contrast-structure
    (image_ops_impl.py:110-129,228-230), and d ssim_c / d q (B,N,C) when asked.  2-d blocks (images) and 3-d blocks
    (volumes: ndim = 3, conv3d with the 11x11x11 window, smoe.py:999-1003)."""
    B, N, C = q.shape
    bs = tuple(int(v) for v in block_shape)
    nd = len(bs)
    assert nd in (2, 3) and min(bs) >= SSIM_PAD, "SYMMETRIC padding by 5 needs blocks of at least 5 pixels per axis"
    win = ssim_window(T, nd)
    c1, c2 = T(SSIM_C1), T(SSIM_C2)
    pad = ((0, 0),) + ((SSIM_PAD, SSIM_PAD),) * nd
    ssim = np.empty((B, C), dtype=T)
    grad = np.zeros((B, N, C), dtype=T) if want_grad else None
    # index of the block pixel every padded position mirrors
    idx = [np.pad(np.arange(b), SSIM_PAD, mode="symmetric") for b in bs]
    flat = idx[0]
    for l in range(1, nd):
        flat = flat[..., None] * bs[l] + idx[l].reshape((1,) * l + (-1,))
    flat = flat.ravel()
    sp = tuple(range(1, nd + 1))
    for c in range(C):
        x = np.pad(q[:, :, c].reshape((B,) + bs).astype(T), pad, mode="symmetric")
        y = np.pad(t[:, :, c].reshape((B,) + bs).astype(T), pad, mode="symmetric")
        mx, my = _ssim_reduce(x, win), _ssim_reduce(y, win)
        num0 = mx * my * T(2)
        den0 = np.square(mx) + np.square(my)
        lum = (num0 + c1) / (den0 + c1)
        num1 = _ssim_reduce(x * y, win) * T(2)
        den1 = _ssim_reduce(np.square(x) + np.square(y), win)
        cs = (num1 - num0 + c2) / (den1 - den0 + c2)
        ssim[:, c] = np.mean(lum * cs, axis=sp)
        if not want_grad:
            continue
        # per window position: d(lum*cs)/d mu_x, /d E[x^2], /d E[xy]
        D0, D1 = den0 + c1, den1 - den0 + c2
        N0, N1 = num0 + c1, num1 - num0 + c2
        d_mx = cs * (T(2) * my * D0 - N0 * T(2) * mx) / np.square(D0) + \
            lum * (-T(2) * my * D1 + N1 * T(2) * mx) / np.square(D1)
        d_s = -lum * N1 / np.square(D1)
        d_p = lum * T(2) / D1
        # adjoint of the VALID correlation: full correlation of the zero-embedded coefficient maps
        full = ((0, 0),) + ((SSIM_SIZE - 1, SSIM_SIZE - 1),) * nd
        wf = win[(slice(None, None, -1),) * nd]
        ga = _ssim_reduce(np.pad(d_mx, full), wf)
        gb = _ssim_reduce(np.pad(d_s, full), wf)
        gc = _ssim_reduce(np.pad(d_p, full), wf)
        gpad = (ga + T(2) * x * gb + y * gc) / T(N)                  # (B, *padded block)
        gp = gpad.reshape(B, -1)
        gc_flat = np.zeros((B, N), dtype=T)
        for bi in range(B):
            np.add.at(gc_flat[bi], flat, gp[bi])                     # fold the mirrored positions back
        grad[:, :, c] = gc_flat
    return ssim, grad


def forward(p: Dict[str, np.ndarray], target: np.ndarray, coords: np.ndarray,
            active: np.ndarray, cfg: OracleConfig, loss_w: Optional[np.ndarray] = None,
            dtype=np.float32, want_grads: bool = False, q_override: Optional[np.ndarray] = None,
            fed: Optional[np.ndarray] = None):
    """One pass of the reference graph over B independent blocks.

    ``fed`` (B, N) bool: the pixels a sub-sampled training pass feeds (smoe.py:1664-1667 passes only the drawn rows of
    ``img_patch``); the kernels that stay on the list are those with influence on a FED pixel (smoe.py:829,1763-1766).

    p: parameter dict with leading B.  target: (B, N, C).  coords: (N, d), or (B, N, d) when
    every block/batch has its own pixel coordinates (shared-kernel mode: global domain).
    active: (B, K) bool = kernel_list (smoe.py:552).  loss_w: (B, N) or None.
    Returns a dict: y (pre-clip), w (gate), wt (masked gate), recon (quantised),
    loss (B,), sse (B,), mse_op (B,), active_new (B,K), argmax (B,N), num_pi (B,),
    and when ``want_grads`` the analytic gradients ``grads`` (dict like p).
    ``q_override`` (B,N,C): use this quantised reconstruction instead of the computed one
    when forming diff/loss/gradients (test hook: the 8-bit quantiser makes loss and
    gradients discontinuous at rounding ties, so a checker feeds the implementation's own
    lattice values here and checks the lattice values separately).
    """
    T = dtype
    B, N, C = target.shape
    K = p["pis"].shape[1]
    d = coords.shape[-1]
    x = coords.astype(T)
    if x.ndim == 2:
        x = np.broadcast_to(x[None], (B,) + x.shape)        # (B,N,d)
    t = target.astype(T)
    graph_quant = cfg.quantization_mode >= 2 or cfg.quantize_pis
    if graph_quant:                                          # smoe.py:474-538: the graph sees fake-quantised variables
        p, qback, _ = quantize_graph_params(p, cfg, T)
    pis = p["pis"].astype(T)
    mu = p["musX"].astype(T)
    nu = p["nu_e"].astype(T)
    gam = p["gamma_e"].astype(T)
    y_only = cfg.only_y_gamma and cfg.use_yuv and cfg.train_gammas            # smoe.py:725
    if y_only:
        gam = gam.copy()
        gam[..., 1:] = T(0)                                  # qgamma_e * gamma_mask
    ic = cfg.train_inverse_cov
    A = _steering(p, T, ic)                                  # (B,K,d,d)
    lw = np.ones((B, N), dtype=T) if loss_w is None else loss_w.astype(T)

    # smoe.py:480,738: bool_mask = kernel_list & (pis > 0)
    act = np.logical_and(active, pis > 0)                    # (B,K)

    # smoe.py:777-782,796: r = x - mu ; z = A^T r ; maha = |z|^2
    r = x[:, None, :, :] - mu[:, :, None, :]                 # (B,K,N,d)
    maha, z = _maha(r, A, ic)                                # (B,K,N)
    n_exp = np.exp(T(-0.5) * maha)                           # smoe.py:807
    if cfg.use_determinant:                                  # smoe.py:809-815
        n_div = np.prod(np.diagonal(A, axis1=-2, axis2=-1), axis=-1)   # (B,K)
        n_dis = T(np.sqrt(np.power(2 * np.pi, d)))
        n_quo = n_div / n_dis
        Nk = n_quo[:, :, None] * n_exp
    else:
        n_quo = np.ones((B, K), dtype=T)
        Nk = n_exp
    g = Nk * pis[:, :, None]                                 # smoe.py:819
    g = np.where(act[:, :, None], g, T(0))                   # masked-out kernels are absent
    S_raw = np.sum(g, axis=1)                                # (B,N)
    S = np.maximum(T(10e-12), S_raw)                         # smoe.py:821
    w = g / S[:, None, :]                                    # smoe.py:823
    tau = T(0.5 * 1 / (2 ** cfg.precision))                  # smoe.py:825
    M = w > tau                                              # tf.greater: strict
    wt = np.where(M, w, T(0))                                # smoe.py:827

    active_new = np.any(M if fed is None else np.logical_and(M, fed[:, None, :]), axis=2)     # smoe.py:829,836
    # smoe.py:833: argmax over the compacted list, mapped back via indices (1706-1716)
    wt_for_arg = np.where(active_new[:, :, None], wt, T(-1))
    argmax = np.argmax(wt_for_arg, axis=1).astype(np.int64)  # first max
    argmax = np.where(np.any(active_new, axis=1)[:, None], argmax, 0)

    # smoe.py:840-848: e = nu + gamma^T x ; y = sum_k wt e
    e = nu[:, :, None, :] + np.einsum("bklc,bnl->bknc", gam, x)  # (B,K,N,C)
    if not cfg.train_gammas:
        e = np.broadcast_to(nu[:, :, None, :], (B, K, N, C)).astype(T)
    y = np.sum(wt[..., None] * e, axis=1)                    # (B,N,C)
    q = fake_quant01(y, cfg.precision, T)                    # smoe.py:857,899
    if q_override is not None:
        q = q_override.astype(T)

    diff = q - t                                             # smoe.py:905
    sse = np.sum(np.square(diff), axis=(1, 2))               # (B,)
    mse_op = sse / T(N * C) * T((2 ** cfg.precision) ** 2)   # smoe.py:927,1053
    eps = T(cfg.margin * 1 / (2 ** cfg.precision))           # smoe.py:931
    a = np.abs(diff) - eps
    lp = np.maximum(T(0), np.square(a)) * lw[:, :, None]     # smoe.py:932
    if cfg.use_yuv:                                          # smoe.py:933-935
        cw = np.array([6 / 8] + [1 / 8] * (C - 1), dtype=T) / T(N)
    else:                                                    # smoe.py:937
        cw = np.full((C,), 1.0 / (N * C), dtype=T)
    loss_pixel = np.sum(np.sum(lp, axis=1) * cw[None, :], axis=1)
    if cfg.ssim_opt:                                         # smoe.py:929,1006-1010 (loss_weights unused there)
        sw = (np.array([6, 1, 1], dtype=T)[:C] / T(8)) if cfg.use_yuv else np.full((C,), 1.0 / C, dtype=T)
        ssim_c, dssim = ssim_and_grad(q, t, cfg.block_shape, T, want_grad=want_grads)
        loss_pixel = T(1) - np.sum(ssim_c * sw[None, :], axis=1)
    diagA = np.diagonal(A, axis1=-2, axis2=-1)               # (B,K,d)
    # smoe.py:1012,1022-1027: num_pi_op = count_nonzero(pis_mask) (independent of the kernel list; no gradient)
    k0 = np.maximum(np.sum(pis > 0, axis=1), 1).astype(T) if cfg.kernel_count_as_norm_l1 else np.full((B,), cfg.k0, dtype=T)
    reg_pi = T(cfg.pis_l1) * np.sum(np.where(act, pis, T(0)), axis=1) / k0             # smoe.py:1027
    reg_u = T(cfg.u_l1) * np.sum(np.where(act[:, :, None], diagA, T(0)), axis=(1, 2))  # smoe.py:1044
    loss = loss_pixel + reg_pi + reg_u                       # smoe.py:1051

    out = {"y": y, "w": w, "wt": wt, "recon": q, "loss": loss, "sse": sse,
           "mse_op": mse_op, "active_new": active_new, "argmax": argmax,
           "num_pi": np.sum(pis > 0, axis=1), "S": S_raw}
    if not want_grads:
        return out

    # ---- analytic reverse pass ------------------------
    nudged_max = T(2 ** cfg.precision - 1) * (T(1) / T(2 ** cfg.precision - 1))
    inside = np.logical_and(y >= T(0), y <= np.minimum(T(1), nudged_max))   # clip + fake-quant STE
    G = (cw[None, None, :] * T(2) * a * np.sign(diff) * lw[:, :, None]) * inside   # (B,N,C)
    if cfg.ssim_opt:
        G = (-sw[None, None, :] * dssim) * inside
    # experts
    g_nu = np.einsum("bkn,bnc->bkc", wt, G)
    g_gam = np.einsum("bkn,bnl,bnc->bklc", wt, x, G)
    # gate
    h = np.where(M, np.einsum("bknc,bnc->bkn", e, G), T(0))  # (B,K,N)
    dotp = np.sum(h * w, axis=1)                             # (B,N)
    passS = (S_raw > T(10e-12))[:, None, :]                  # tf.maximum tie -> constant
    u = np.where(passS, w * (h - dotp[:, None, :]), h * w)   # dL/dlog g  (if floored: dL/dg*g = h*g/S)
    u = np.where(act[:, :, None], u, T(0))
    safe_pi = np.where(act, pis, T(1))
    g_pi = np.where(act, np.sum(u, axis=2) / safe_pi + T(cfg.pis_l1) / k0[:, None], T(0))
    # steering: dm/dA[l,m] = 2 r_l z_m, dL/dm = -u/2
    if ic:      # maha = r^T A r: dm/dA_ll = r_l^2, dm/dA_corr[l,m] = 2 r_l r_m (the entry sits at (l,m) and (m,l))
        rr = np.einsum("bkn,bknl,bknm->bklm", u, r, r)
        g_A = -T(0.5) * rr * np.where(np.eye(d, dtype=bool), T(1), T(2))
    else:       # dm/dA[l,m] = 2 r_l z_m
        g_A = -np.einsum("bkn,bknl,bknm->bklm", u, r, z)     # (B,K,d,d), valid for l>=m
    if cfg.use_determinant:
        safe_d = np.where(act[:, :, None], diagA, T(1))
        g_A = g_A + np.einsum("bk,bkl,lm->bklm", np.sum(u, axis=2), T(1) / safe_d,
                              np.eye(d, dtype=T))
    g_A = g_A + np.where(act[:, :, None, None], T(cfg.u_l1) * np.eye(d, dtype=T), T(0))
    eye = np.eye(d, dtype=bool)
    low = np.tril(np.ones((d, d), dtype=bool), -1)
    g_Adiag = np.where(eye, g_A, T(0))
    g_Acorr = np.where(low, g_A, T(0))
    if cfg.radial_as:       # a is tiled over the diagonal (smoe.py:714-719): dL/da = sum_l dL/dA_ll, kept on every diagonal entry
        tr = np.trace(g_Adiag, axis1=-2, axis2=-1)[..., None, None]
        g_Adiag = np.where(eye, tr, T(0)).astype(T)
        g_Acorr = np.zeros_like(g_Acorr)                    # A_corr_var is not trainable (smoe.py:434)
    # centres: dm/dmu = -2 A z
    Az = z if ic else np.einsum("bklm,bknm->bknl", A, z)    # dm/dmu = -2 A r (symmetric A) resp. -2 A A^T r
    g_mu = np.einsum("bkn,bknl->bkl", u, Az)
    if not cfg.train_gammas:
        g_gam = np.zeros_like(g_gam)
    if y_only:
        g_gam[..., 1:] = T(0)
    out["grads"] = {"pis": g_pi, "musX": g_mu, "A_diagonal": g_Adiag, "A_corr": g_Acorr,
                    "gamma_e": g_gam, "nu_e": g_nu}
    if graph_quant:                                          # back through the fake-quant ops
        out["grads"] = route_quant_grads(out["grads"], qback, T)
    return out


# --------------------------------------------------------------------------
# optimiser (TF1 Adam, three groups)
# --------------------------------------------------------------------------
def adam_step(p, grads, state, cfg: OracleConfig, dtype=np.float32, frozen=None):
    """One ``session.run(train_op)`` (smoe.py:1788): dense TF1 ApplyAdam on every
    trainable variable of a group whose lr != 0 (smoe.py:1112-1144,1173-1193).
    ``frozen``: optional (B,) bool of blocks that must not move (diverged)."""
    T = dtype
    b1, b2, eps = T(cfg.beta1), T(cfg.beta2), T(cfg.adam_eps)
    if state["b1p"] is None:
        state["b1p"], state["b2p"] = b1, b2
    b1p, b2p = T(state["b1p"]), T(state["b2p"])
    groups = {
        "nu_e": (cfg.lr_expert, True), "gamma_e": (cfg.lr_expert, cfg.train_gammas),
        "musX": (cfg.lr_expert, cfg.train_musx), "pis": (cfg.lr_pis, cfg.train_pis),
        "A_diagonal": (cfg.lr_steer, True), "A_corr": (cfg.lr_steer, not cfg.radial_as),
    }
    newp = {}
    for name in PARAM_NAMES:
        lr, trainable = groups[name]
        var = p[name].astype(T)
        if (not trainable) or lr == 0:
            newp[name] = var
            continue
        gr = grads[name].astype(T)
        if cfg.grad_clip is not None:
            gr = np.clip(gr, T(-cfg.grad_clip), T(cfg.grad_clip))
        m = state["m"][name].astype(T)
        v = state["v"][name].astype(T)
        alpha = T(lr) * np.sqrt(T(1) - b2p) / (T(1) - b1p)
        m_new = m + (gr - m) * (T(1) - b1)
        v_new = v + (gr * gr - v) * (T(1) - b2)
        var_new = var - (m_new * alpha) / (np.sqrt(v_new) + eps)
        if frozen is not None:
            fz = frozen.reshape((-1,) + (1,) * (var.ndim - 1))
            m_new = np.where(fz, m, m_new)
            v_new = np.where(fz, v, v_new)
            var_new = np.where(fz, var, var_new)
        state["m"][name] = m_new
        state["v"][name] = v_new
        newp[name] = var_new
    state["b1p"] = T(b1p * b1)
    state["b2p"] = T(b2p * b2)
    state["t"] += 1
    return newp


# --------------------------------------------------------------------------
# pass / iteration semantics
# --------------------------------------------------------------------------
def readmit(p, active, cfg: OracleConfig, dtype=np.float32):
    """``update_kernel_list`` (smoe.py:2287-2365) for a per-block [0,1]^d domain:
    active |= (pis>0) & any_probe(maha < 800), probes = {min, max, mid}^d of the
    block's coordinates (smoe.py:2322-2333,2350; test at smoe.py:806)."""
    T = dtype
    d = cfg.dim
    axes = axis_coords(cfg.block_shape)
    tt = [(ax.min(), ax.max(), (ax.min() + ax.max()) / 2) for ax in axes]
    probes = np.array(list(itertools.product(*tt))).astype(np.float32).astype(T)   # (3^d, d)
    if cfg.quantization_mode >= 2 or cfg.quantize_pis:      # maha_dist_ind is part of the same graph (smoe.py:806)
        p = quantize_graph_params(p, cfg, T)[0]
    A = _steering(p, T, cfg.train_inverse_cov)
    mu = p["musX"].astype(T)
    r = probes[None, None, :, :] - mu[:, :, None, :]
    maha, _ = _maha(r, A, cfg.train_inverse_cov)
    near = np.any(maha < T(800), axis=2)
    return np.logical_or(active, np.logical_and(near, p["pis"] > 0))


def fit(p, target, coords, cfg: OracleConfig, n_iters: int, val_iter: int = 100,
        ukl_iter: Optional[int] = None, loss_w=None, dtype=np.float32,
        record_every: int = 0):
    """``Smoe.train`` (smoe.py:1485-1603) over B independent blocks.

    Iteration 0: eval pass (prunes ``active`` to the kernels with influence,
    smoe.py:1763-1766).  Each iteration: train pass with the current ``active``,
    ``active <- active'``, Adam.  Every ``ukl_iter``: readmit.  Every
    ``val_iter``: eval pass, per-block best snapshot if the loss improved
    (smoe.py:1574-1576).  Per-block stop on NaN / blow-up (smoe.py:1565-1570).
    Returns (params, state, info).
    """
    T = dtype
    if ukl_iter is None:
        ukl_iter = val_iter
    B = target.shape[0]
    K = p["pis"].shape[1]
    p = {k: v.astype(T) for k, v in p.items()}
    state = new_adam_state(p)
    active = np.ones((B, K), dtype=bool)                    # smoe.py:315
    f0 = forward(p, target, coords, active, cfg, loss_w, T)
    active = f0["active_new"]
    loss0 = f0["loss"].copy()
    best_loss = f0["loss"].copy()
    best = {k: v.copy() for k, v in p.items()}
    stopped = np.zeros((B,), dtype=bool)
    hist = {"iter": [0], "loss": [f0["loss"].copy()], "sse": [f0["sse"].copy()]}
    trace = []
    for i in range(1, n_iters + 1):
        f = forward(p, target, coords, active, cfg, loss_w, T, want_grads=True)
        # reference order: pass (loss at current params) -> prune -> Adam -> divergence test
        active = np.where(stopped[:, None], active, f["active_new"])
        p = adam_step(p, f["grads"], state, cfg, T, frozen=stopped)
        bad = np.logical_or(np.isnan(f["loss"]), f["loss"] + 1 > (loss0 + 100) * 10)
        stopped = np.logical_or(stopped, bad)
        if record_every and i % record_every == 0:
            trace.append((i, f["loss"].copy(), f["sse"].copy()))
        if i % ukl_iter == 0:
            active = readmit(p, active, cfg, T)
        if i % val_iter == 0:
            fv = forward(p, target, coords, active, cfg, loss_w, T)
            active = fv["active_new"]
            better = fv["loss"] < best_loss
            best_loss = np.where(better, fv["loss"], best_loss)
            for k in best:
                bm = better.reshape((-1,) + (1,) * (p[k].ndim - 1))
                best[k] = np.where(bm, p[k], best[k])
            hist["iter"].append(i)
            hist["loss"].append(fv["loss"].copy())
            hist["sse"].append(fv["sse"].copy())
    info = {"active": active, "best": best, "best_loss": best_loss, "loss0": loss0,
            "stopped": stopped, "hist": hist, "trace": trace}
    return p, state, info


def psnr_from_sse(sse_total: float, n_values: int) -> float:
    """plotter.py:14-15 with mse_op of smoe.py:1053: 10*log10((2^p)^2 / (mean(diff^2)*(2^p)^2))."""
    return float(-10.0 * np.log10(sse_total / n_values))


# --------------------------------------------------------------------------
# shared-kernel image mode: ONE global kernel set, per-batch kernel lists,
# gradients accumulated over the batches of a pass, one Adam step per pass
# --------------------------------------------------------------------------
def global_batch_coords(image_shape: Sequence[int], batch_shape: Sequence[int], dtype=np.float32) -> np.ndarray:
    """(NB, Nb, d) global pixel coordinates of every batch in sliding_window order
    (smoe.py:18-35, 2412): linspace(0,1,size) per image axis, fed as float32."""
    d = len(batch_shape)
    axes = [np.linspace(0, 1, int(s)).astype(np.float32).astype(dtype) for s in image_shape]
    grids = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1)                 # (*image, d)
    g = [int(s) // int(b) for s, b in zip(image_shape, batch_shape)]
    split = []
    for gi, bi in zip(g, batch_shape):
        split += [gi, int(bi)]
    perm = list(range(0, 2 * d, 2)) + list(range(1, 2 * d, 2))
    return grids.reshape(split + [d]).transpose(perm + [2 * d]).reshape(int(np.prod(g)), -1, d)


def global_halo_coords(image_shape: Sequence[int], batch_shape: Sequence[int], overlap: int, dtype=np.float32):
    """(NB, Next, d) coordinates of the EXTENDED windows the reference iterates when
    overlap_of_batches > 0 (sliding_window, smoe.py:18-35): the joint domain is zero-padded by
    ``overlap`` on every side, so a window pixel outside the image has ALL coordinates 0."""
    d = len(batch_shape)
    axes = [np.linspace(0, 1, int(s)).astype(np.float32).astype(dtype) for s in image_shape]
    grids = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1)
    pad = [(overlap, overlap)] * d + [(0, 0)]
    padded = np.pad(grids, pad, mode="constant", constant_values=0)
    out = []
    ranges = [range(0, int(s), int(b)) for s, b in zip(image_shape, batch_shape)]
    for org in itertools.product(*ranges):
        sl = tuple(slice(o, o + int(b) + 2 * overlap) for o, b in zip(org, batch_shape))
        out.append(padded[sl].reshape(-1, d))
    return np.stack(out)


def shared_init_params(image: np.ndarray, kernels_per_dim: Sequence[int], normalize_pis: bool = True):
    """Smoe.__init__ without init_params for the whole image (smoe.py:260-262): parameters with a
    leading axis of 1 (one model)."""
    return init_params(image[None], kernels_per_dim, normalize_pis)


def _bcast(p, NB):
    return {k: np.broadcast_to(v, (NB,) + v.shape[1:]) for k, v in p.items()}


def shared_pass(p, target, coords, lists, cfg: OracleConfig, dtype=np.float32, want_grads=False, halo_coords=None,
                loss_w=None):
    """One run_batched pass (smoe.py:1606-1793) in shared-kernel mode.  p: leading axis 1;
    target (NB,Nb,C); coords (NB,Nb,d); lists (NB,K) bool.  Returns the per-batch forward dict plus
    ``loss_val``/``mse_val`` (pixel-weighted means, smoe.py:1758-1759), ``lists_new`` (1763-1766)
    and, with want_grads, ``grads`` = SUM over batches of the per-batch gradients (smoe.py:1150)."""
    NB = target.shape[0]
    f = forward(_bcast(p, NB), target, coords, lists, cfg, loss_w, dtype, want_grads=want_grads)      # loss_w: (NB,Nb) or None
    f["loss_val"] = float(np.mean(f["loss"]))                       # equal-size batches
    f["mse_val"] = float(np.mean(f["mse_op"]))
    f["lists_new"] = f["active_new"]
    if halo_coords is not None:
        # overlap_of_batches > 0: the graph runs on the extended window, the loss (and with it every
        # gradient) is cropped to the interior (smoe.py:909-923); what the halo changes is the
        # influence test over the whole window (smoe.py:829,836) -> the new kernel list
        fe = forward(_bcast(p, NB), np.zeros(halo_coords.shape[:2] + (target.shape[2],), dtype), halo_coords, lists,
                     cfg, None, dtype)
        f["lists_new"] = fe["active_new"]
    if want_grads:
        f["grads"] = {k: np.sum(v, axis=0, keepdims=True) for k, v in f["grads"].items()}
    return f


def shared_readmit(p, lists, coords, cfg: OracleConfig, dtype=np.float32):   # coords: the (extended) windows
    """update_kernel_list (smoe.py:2287-2365): per batch, probes = {min,max,mid}^d of the batch's
    coordinates; list |= (pis>0) & any_probe(maha < 800)."""
    T = dtype
    NB, _, d = coords.shape
    mins, maxs = coords.min(axis=1).astype(np.float64), coords.max(axis=1).astype(np.float64)
    tt = np.stack([mins, maxs, (mins + maxs) / 2], axis=-1)                       # (NB,d,3)
    if cfg.quantization_mode >= 2 or cfg.quantize_pis:      # the probe test is part of the fake-quantised graph
        p = quantize_graph_params(p, cfg, T)[0]
    A = _steering(p, T, cfg.train_inverse_cov)[0]
    mu = p["musX"].astype(T)[0]
    out = lists.copy()
    for b in range(NB):
        probes = np.array(list(itertools.product(*tt[b]))).astype(np.float32).astype(T)
        r = probes[None, :, :] - mu[:, None, :]
        near = np.any(_maha(r, A, cfg.train_inverse_cov)[0] < T(800), axis=1)
        out[b] |= near & (p["pis"][0] > 0)
    return out


def shared_fit(p, target, coords, cfg: OracleConfig, n_iters: int, val_iter: int = 100, ukl_iter=None,
               dtype=np.float32, halo_coords=None, loss_w=None):
    """Smoe.train in shared-kernel mode (smoe.py:1485-1603): iteration-0 eval pass, per iteration a
    train pass (prune lists) + one Adam step on the accumulated gradients, readmission every
    ukl_iter, eval + best snapshot every val_iter."""
    T = dtype
    if ukl_iter is None:
        ukl_iter = val_iter
    NB = target.shape[0]
    K = p["pis"].shape[1]
    p = {k: v.astype(T) for k, v in p.items()}
    state = new_adam_state(p)
    lists = np.ones((NB, K), dtype=bool)                              # smoe.py:315
    f0 = shared_pass(p, target, coords, lists, cfg, T, halo_coords=halo_coords, loss_w=loss_w)
    lists = f0["lists_new"]
    hist = {"iter": [0], "loss": [f0["loss_val"]], "mse": [f0["mse_val"]]}
    best, best_loss = {k: v.copy() for k, v in p.items()}, f0["loss_val"]
    train_losses = []
    for i in range(1, n_iters + 1):
        f = shared_pass(p, target, coords, lists, cfg, T, want_grads=True, halo_coords=halo_coords, loss_w=loss_w)
        lists = f["lists_new"]
        p = adam_step(p, f["grads"], state, cfg, T)
        train_losses.append(f["loss_val"])
        if i % ukl_iter == 0:
            lists = shared_readmit(p, lists, coords if halo_coords is None else halo_coords, cfg, T)
        if i % val_iter == 0:
            fv = shared_pass(p, target, coords, lists, cfg, T, halo_coords=halo_coords, loss_w=loss_w)
            lists = fv["lists_new"]
            if fv["loss_val"] < best_loss:
                best_loss, best = fv["loss_val"], {k: v.copy() for k, v in p.items()}
            hist["iter"].append(i)
            hist["loss"].append(fv["loss_val"])
            hist["mse"].append(fv["mse_val"])
    return p, state, {"lists": lists, "hist": hist, "best": best, "best_loss": best_loss,
                      "train_losses": train_losses}
