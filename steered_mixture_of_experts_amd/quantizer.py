"""Uniform min/max parameter quantiser + rescaler of the codec experiments (reference
quantizer.py:4-145), vectorised over independent blocks.

Per block the arithmetic is the reference's: kernels with ``pis <= 0`` are dropped
(``reduce_params``, utils.py:7-16), bounds are the min/max over the block's remaining kernels
(modes 0/1/3) or the fixed ``lower_bounds/upper_bounds`` (mode 2 / ``quantize_pis``),
``q = round((x - lb) / (ub - lb + 10e-12) * (2^bits - 1))`` and
``r = q / (2^bits - 1) * (ub - lb) + lb``.  Dropped kernels keep their slot (arrays stay
rectangular) with ``q = 0`` and are rescaled to ``pis = 0`` so they stay absent.
``bit_depths`` order: A, musX, nu_e, pis, gamma_e (smoe_test.py:302-303)."""
from __future__ import annotations

from typing import Dict

import numpy as np

_BITS = {"A_diagonal": 0, "A_corr": 0, "musX": 1, "nu_e": 2, "pis": 3, "gamma_e": 4}
_STEP_NAME = {"A_diagonal": "A", "A_corr": "A", "musX": "musX", "nu_e": "nu_e", "pis": "pis", "gamma_e": "gamma_e"}


def _masked_minmax(x: np.ndarray, keep: np.ndarray):
    """min / max over the kernel axis (axis 1) of the kept kernels, keepdims (quantizer.py:9-19)."""
    m = keep.reshape(keep.shape + (1,) * (x.ndim - 2))
    lo = np.where(m, x, np.inf).min(axis=1, keepdims=True)
    hi = np.where(m, x, -np.inf).max(axis=1, keepdims=True)
    none = ~keep.any(axis=1)
    if none.any():                                   # a block with no kernel left: harmless bounds
        lo[none] = 0.0
        hi[none] = 0.0
    return lo, hi


def quantize_params(smoe, params: Dict[str, np.ndarray]) -> Dict[str, object]:
    """quantizer.py:4-83 for every block.  ``params``: get_params() layout with leading B."""
    radial = bool(getattr(smoe, "radial_as", False))       # A_diagonal is the (B, K) vector of the radial values, no A_corr
    mode = smoe.quantization_mode
    bits = list(smoe.bit_depths)
    keep = params["pis"] > 0                                             # reduce_params
    p = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
    lower, upper = {}, {}
    names = tuple(n for n in _BITS if not (radial and n == "A_corr"))         # quantizer.py:12-14,43-45,60-62
    if radial and p["A_diagonal"].ndim == 4:                # engine layout (equal diagonals) -> the radial value
        p["A_diagonal"] = p["A_diagonal"][:, :, 0, 0]
    for name in (n for n in names if n != "pis"):
        if mode <= 1 or mode == 3:
            lower[name], upper[name] = _masked_minmax(p[name], keep)
        else:                                                            # mode 2: fixed bounds
            idx = {"A_diagonal": 0, "A_corr": 0, "musX": 1, "nu_e": 2, "gamma_e": 4}[name]
            shape = (p[name].shape[0], 1) + p[name].shape[2:]
            lower[name] = np.ones(shape) * smoe.lower_bounds[idx]
            upper[name] = np.ones(shape) * smoe.upper_bounds[idx]
    if mode <= 1 and not smoe.quantize_pis:
        lower["pis"], upper["pis"] = _masked_minmax(p["pis"], keep)
    else:
        lower["pis"] = np.ones((p["pis"].shape[0], 1)) * smoe.lower_bounds[3]
        upper["pis"] = np.ones((p["pis"].shape[0], 1)) * smoe.upper_bounds[3]
    steps = {"A": 2 ** bits[0] - 1, "musX": 2 ** bits[1] - 1, "nu_e": 2 ** bits[2] - 1,
             "pis": 2 ** bits[3] - 1, "gamma_e": 2 ** bits[4] - 1}
    q = {"lower_bounds": lower, "upper_bounds": upper, "steps": steps, "used_kernels": keep}
    for name in names:
        normalized = (p[name] - lower[name]) / (upper[name] - lower[name] + 10e-12)
        qv = np.round(normalized * steps[_STEP_NAME[name]])
        m = keep.reshape(keep.shape + (1,) * (qv.ndim - 2))
        q[name] = np.where(m, qv, 0.0)
    return q


def rescaler(smoe, qparams: Dict[str, object]) -> Dict[str, np.ndarray]:
    """quantizer.py:85-145 for every block.  Returns the reference's ``rparams`` keys
    (``A = rA_diagonal + rA_corr``, musX, nu_e, pis, gamma_e) plus the split matrices the
    engine consumes."""
    steps, lo, hi = qparams["steps"], qparams["lower_bounds"], qparams["upper_bounds"]
    keep = qparams["used_kernels"]
    r = {}
    for name in (n for n in _BITS if n in qparams):
        r[name] = qparams[name] / steps[_STEP_NAME[name]] * (hi[name] - lo[name]) + lo[name]
    r["pis"] = np.where(keep, r["pis"], 0.0)
    if "A_corr" not in r:                                   # radial_as (quantizer.py:128-133): A = a * I
        d = r["musX"].shape[-1]
        r["A_diagonal"] = r["A_diagonal"][..., None, None] * np.eye(d)
        r["A_corr"] = np.zeros_like(r["A_diagonal"])
    out = {"A": r["A_diagonal"] + r["A_corr"], "musX": r["musX"], "nu_e": r["nu_e"], "pis": r["pis"],
           "gamma_e": r["gamma_e"], "A_diagonal": r["A_diagonal"], "A_corr": r["A_corr"]}
    if getattr(smoe, "use_diff_center", False):
        out["musX"] = out["musX"] + smoe.musX_init
    return out
