"""32x32 / K = 8 / RGB on the 64-lane tiling: time per 50 iterations of every graph variant (plain, inverse covariance,
quantization_mode 3, SSIM loss)."""
import sys
import numpy as np
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_parity import _engine, _setup, _to_dev, _planar
QKW = dict(bit_depths=(14, 12, 8, 10, 10), lower_bounds=(-60, -.3, -1, 0, -4), upper_bounds=(60, 1.3, 2, 2, 4))
shape, C, kpd, B = (32, 32), 3, [2, 4], 2040
for name, kw in (("plain", {}), ("inverse covariance", dict(train_inverse_cov=True)),
                 ("mode 3", dict(quantization_mode=3, quantize_pis=True, **QKW)), ("mode 2", dict(quantization_mode=2, quantize_pis=True, **QKW)),
                 ("ssim", dict(ssim_opt=True))):
    cfg, p, coords, tgt, K = _setup(shape, C, kpd, True, B, 7, **kw)
    p["A_corr"] = p["A_corr"] * np.tril(np.ones((2, 2), np.float32), -1)
    eng = _engine(shape, C, K, use_yuv=True, **kw)
    eng.set_tiling(64)
    T = _planar(tgt)
    act = torch.full((B,), (1 << K) - 1, dtype=torch.int32, device="cuda")
    ts = []
    for r in range(6):
        dp = _to_dev(p); st = eng.new_adam_state(dp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.fit(T, dp, st, act, 50); e1.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(e0.elapsed_time(e1))
    print(f"{name:20s} {np.median(ts):8.3f} ms per 50 iterations  ({B * 1024 * 50 / np.median(ts) / 1e6:6.1f} Gpx-it/s)", flush=True)
    eng.close()
