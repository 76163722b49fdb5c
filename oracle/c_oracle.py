"""ctypes wrapper of oracle/libsmoe_oracle.so (plain-C restatement).  TEST INFRASTRUCTURE
ONLY -- see the header of smoe_oracle.c.  Used by tests/ (cross-check of the numpy oracle)
and by bench.py's cpu_baseline leg (the timed CPU "port")."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libsmoe_oracle.so")
NAMES = ("pis", "musX", "A_diagonal", "A_corr", "gamma_e", "nu_e")


class OracleCfg(C.Structure):
    _fields_ = [("dim", C.c_int32), ("channels", C.c_int32), ("kernels", C.c_int32), ("pixels", C.c_int32),
                ("precision", C.c_int32), ("margin", C.c_float), ("use_determinant", C.c_int32),
                ("use_yuv", C.c_int32), ("train_pis", C.c_int32), ("train_gammas", C.c_int32),
                ("train_musx", C.c_int32), ("lr_expert", C.c_float), ("lr_pis", C.c_float),
                ("lr_steer", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
                ("grad_clip", C.c_float), ("pis_l1", C.c_float), ("u_l1", C.c_float), ("start_pis", C.c_int32),
                ("only_y_gamma", C.c_int32), ("quantize_pis", C.c_int32), ("pis_bits", C.c_int32),
                ("pis_lb", C.c_float), ("pis_ub", C.c_float)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", _HERE])
        _lib = C.CDLL(LIB)
    return _lib


def _cfg(cfg):
    """cfg: oracle.smoe_oracle.OracleConfig"""
    if getattr(cfg, "ssim_opt", False):
        raise NotImplementedError("the plain-C restatement covers the margin loss only; SSIM lives in smoe_oracle.py")
    c = OracleCfg()
    c.dim, c.channels, c.kernels, c.pixels = cfg.dim, cfg.channels, cfg.kernels, cfg.pixels
    c.precision, c.margin = cfg.precision, cfg.margin
    c.use_determinant, c.use_yuv = int(cfg.use_determinant), int(cfg.use_yuv)
    c.train_pis, c.train_gammas, c.train_musx = int(cfg.train_pis), int(cfg.train_gammas), int(cfg.train_musx)
    c.lr_expert, c.lr_pis, c.lr_steer = cfg.lr_expert, cfg.lr_pis, cfg.lr_steer
    c.beta1, c.beta2, c.adam_eps = cfg.beta1, cfg.beta2, cfg.adam_eps
    c.grad_clip = cfg.grad_clip or 0.0
    c.pis_l1, c.u_l1, c.start_pis = cfg.pis_l1, cfg.u_l1, cfg.k0
    c.only_y_gamma = int(getattr(cfg, 'only_y_gamma', False))
    if getattr(cfg, "kernel_count_as_norm_l1", False):
        raise NotImplementedError("kernel_count_as_norm_l1 lives in smoe_oracle.py")
    if getattr(cfg, "radial_as", False):
        raise NotImplementedError("the plain-C restatement has the CLI-default form only; radial_as lives in smoe_oracle.py")
    if getattr(cfg, "train_inverse_cov", False):
        raise NotImplementedError("the plain-C restatement has the CLI-default form only; train_inverse_cov lives in smoe_oracle.py")
    if getattr(cfg, "quantization_mode", 0) >= 2:
        raise NotImplementedError("the plain-C restatement has quantize_pis only; modes 2/3 live in smoe_oracle.py")
    c.quantize_pis = int(getattr(cfg, "quantize_pis", False))
    c.pis_bits = int(cfg.bit_depths[3]) if c.quantize_pis else 0
    c.pis_lb = float(cfg.lower_bounds[3]) if c.quantize_pis else 0.0
    c.pis_ub = float(cfg.upper_bounds[3]) if c.quantize_pis else 0.0
    return c


def _ptrs(d):
    arr = (C.c_void_p * 6)()
    for i, n in enumerate(NAMES):
        a = d[n]
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
        arr[i] = a.ctypes.data
    return arr


def _f(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def forward(cfg, coords_dn, target_bcn, p, active_bits, loss_w=None, want_recon=True, update_active=True, threads=1):
    """coords_dn: (d,N) float32; target_bcn: (B,C,N) float32; p: dict of float32 arrays; active_bits: (B,) uint32 in/out."""
    lib = load()
    B = target_bcn.shape[0]
    recon = np.empty_like(target_bcn) if want_recon else None
    loss = np.empty((B,), np.float32)
    sse = np.empty((B,), np.float32)
    c = _cfg(cfg)
    rc = lib.smoe_oracle_forward(C.byref(c), B, _f(coords_dn), _f(target_bcn), _f(loss_w), _ptrs(p), _f(recon),
                                 _f(loss), _f(sse), _f(active_bits), int(update_active), int(threads))
    assert rc == 0
    return {"recon": recon, "loss": loss, "sse": sse}


def fit(cfg, coords_dn, target_bcn, p, m, v, active_bits, n_iters, beta_pow, loss_w=None, diverged=None,
        loss0=None, threads=1):
    """In-place n_iters training iterations; beta_pow: float32[2] in/out."""
    lib = load()
    B = target_bcn.shape[0]
    loss = np.zeros((B,), np.float32)
    sse = np.zeros((B,), np.float32)
    c = _cfg(cfg)
    rc = lib.smoe_oracle_fit(C.byref(c), B, _f(coords_dn), _f(target_bcn), _f(loss_w), _ptrs(p), _ptrs(m), _ptrs(v),
                             int(n_iters), _f(beta_pow), _f(loss), _f(sse), _f(active_bits), _f(diverged), _f(loss0),
                             int(threads))
    assert rc == 0
    return {"loss": loss, "sse": sse}
