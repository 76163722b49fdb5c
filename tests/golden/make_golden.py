#!/usr/bin/env python3
"""Generates tests/golden/*.npz: inputs and expected outputs of the hot path for the four
BASELINE shapes, computed by the fp64 oracle (oracle/smoe_oracle.py).

The reference itself cannot run here (TensorFlow 1.x is not installable offline and the
reference HEAD raises NameError for 2-D inputs), so these fixtures pin the
RESTATEMENT, not the reference: "parity unpinned".  They are data only (inputs + expected
outputs); the GPU tests compare the HIP path against them on the GPU box.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import smoe_oracle as o                                   # noqa: E402
from steered_mixture_of_experts_amd.blocks import synthetic_blocks    # noqa: E402

CASES = [("cfg2_16x16_k4_c1", (16, 16), 1, [2, 2], False),
         ("cfg3_32x32_k8_c3", (32, 32), 3, [2, 4], True),
         ("cfg4_16x16_k4_c3", (16, 16), 3, [2, 2], True),
         ("cfg5_16x16x4_k4_c3", (16, 16, 4), 3, [2, 2, 1], True)]
N_ITERS = 5
B = 4

for idx, (name, shape, C, kpd, yuv) in enumerate(CASES):
    K = int(np.prod(kpd))
    blk = synthetic_blocks(B, shape, C, 20260100 + idx)
    p = o.init_params(blk, kpd)
    rng = np.random.default_rng(idx)
    p["A_corr"] = (rng.normal(size=p["A_corr"].shape) * 0.5).astype(np.float32)
    p["gamma_e"] = (rng.normal(size=p["gamma_e"].shape) * 0.05).astype(np.float32)
    cfg = o.OracleConfig(block_shape=shape, channels=C, kernels=K, use_yuv=yuv)
    coords = o.block_coords(shape, np.float64)
    tgt = blk.reshape(B, -1, C)
    act = np.ones((B, K), bool)
    # TF computes in fp32: the lattice values (and with them sign(q - t) at exactly reconstructed
    # pixels) are those of the fp32 restatement; loss / gradients are then evaluated in fp64 GIVEN
    # these lattice values, and the short fit is the fp32 restatement's.
    f32 = o.forward(p, tgt, o.block_coords(shape), act, cfg, None, np.float32)
    f = o.forward(p, tgt, coords, act, cfg, None, np.float64, want_grads=True, q_override=f32["recon"])
    f["recon"] = f32["recon"]
    pn, _, info = o.fit(p, tgt, o.block_coords(shape), cfg, N_ITERS, val_iter=10 ** 9, dtype=np.float32)
    out = {"block_shape": np.array(shape), "channels": C, "kernels": K, "use_yuv": yuv, "n_iters": N_ITERS,
           "target": tgt.astype(np.float32), "loss": f["loss"], "sse": f["sse"],
           "recon": f["recon"].astype(np.float32), "y": f["y"], "active_after_fit": info["active"]}
    for k in o.PARAM_NAMES:
        out["p_" + k] = p[k]
        out["g_" + k] = f["grads"][k]
        out["fit_" + k] = pn[k]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz")
    np.savez_compressed(path, **out)
    print(name, os.path.getsize(path), "bytes")
