"""Diagnostic: GPU vs oracle fp32 vs oracle fp64 errors (gradients after one step; short trajectories)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from oracle import smoe_oracle as o
from steered_mixture_of_experts_amd.blocks import synthetic_blocks
from test_gpu_parity import _setup, _engine, _to_dev, _to_host, _planar, _mask_to_bits, SHAPES

for (shape, C, kpd, yuv) in SHAPES:
    B = 21
    cfg, p, coords, tgt, K = _setup(shape, C, kpd, yuv, B, 300 + len(shape) + C)
    active = np.ones((B, K), dtype=bool)
    eng = _engine(shape, C, K, use_yuv=yuv)
    dp = _to_dev(p); act = torch.from_numpy(_mask_to_bits(active).view(np.int32)).cuda(); T = _planar(tgt)
    fw = eng.forward(T, dp, act, want_recon=True, update_active=False)
    recon = np.transpose(fw["recon"].cpu().numpy(), (0, 2, 1))
    ref = o.forward(p, tgt, coords, active, cfg, None, np.float32, want_grads=True, q_override=recon)
    ref64 = o.forward(p, tgt, coords, active, cfg, None, np.float64, want_grads=True, q_override=recon)
    state = eng.new_adam_state(dp)
    eng.fit(T, dp, state, act, 1)
    torch.cuda.synchronize()
    m = _to_host(state.m)
    print(shape, C, kpd)
    for name in o.PARAM_NAMES:
        scale = np.abs(ref64["grads"][name]).max() + 1e-30
        eg = np.abs(m[name] / 0.1 - ref64["grads"][name]).max() / scale
        er = np.abs(ref["grads"][name] - ref64["grads"][name]).max() / scale
        print(f"   {name:11s} gpu-vs-64 {eg:.2e}   ref32-vs-64 {er:.2e}  scale {scale:.2e}")
    eng.close()

# trajectories
shape, C, kpd = (16, 16), 1, [2, 2]
B = 1024
cfg, p, coords, tgt, K = _setup(shape, C, kpd, False, B, 777, perturb=False)
ps = lambda sse: -10 * np.log10(np.maximum(sse, 1e-12) / (tgt.shape[1] * C))
for n in (5, 20, 50):
    p32, _, i32 = o.fit(p, tgt, coords, cfg, n, val_iter=10 ** 9, dtype=np.float32)
    p64, _, i64 = o.fit(p, tgt, coords, cfg, n, val_iter=10 ** 9, dtype=np.float64)
    eng = _engine(shape, C, K)
    dp = _to_dev(p); state = eng.new_adam_state(dp)
    act = torch.full((B,), (1 << K) - 1, dtype=torch.int32, device="cuda"); T = _planar(tgt)
    eng.forward(T, dp, act, want_recon=False)
    eng.fit(T, dp, state, act, n)
    out = eng.forward(T, dp, act, want_recon=False, update_active=False)
    torch.cuda.synchronize()
    f32 = o.forward(p32, tgt, coords, i32["active"], cfg, None, np.float32)
    f64 = o.forward(p64, tgt, coords, i64["active"], cfg, None, np.float64)
    g = ps(out["sse"].cpu().numpy()); a = ps(f32["sse"]); b = ps(f64["sse"])
    agg = lambda s: -10*np.log10(s.sum()/(B*256))
    print(f"n={n}: median PSNR gpu {np.median(g):.3f} ref32 {np.median(a):.3f} ref64 {np.median(b):.3f} | agg gpu {agg(out['sse'].cpu().numpy()):.3f} ref32 {agg(f32['sse']):.3f} ref64 {agg(f64['sse']):.3f}"
          f" | median|d| gpu-32 {np.median(np.abs(g-a)):.4f} 32-64 {np.median(np.abs(a-b)):.4f} gpu-64 {np.median(np.abs(g-b)):.4f}")
    got = _to_host(dp)
    for name in ("nu_e", "musX", "A_diagonal"):
        print(f"     {name}: median|gpu-32| {np.median(np.abs(got[name]-p32[name])):.2e}  median|32-64| {np.median(np.abs(p32[name]-p64[name])):.2e}")
    eng.close()
