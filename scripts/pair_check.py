"""Pair mode (one block on both wavefronts of a workgroup, smoe_set_tiling 128) against the plain 64-lane kernel:
same start, same iterations -> parameters, and the time per launch."""
import sys
import numpy as np
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_parity import _engine, _setup, _to_dev, _to_host, _planar

def run(shape, C, kpd, B, tiling, n=100, reps=5, **kw):
    cfg, p, coords, tgt, K = _setup(shape, C, kpd, False, B, 7, **kw)
    eng = _engine(shape, C, K, **kw)
    eng.set_tiling(tiling)
    T = _planar(tgt)
    act = torch.full((B,), (1 << K) - 1, dtype=torch.int32, device="cuda")
    out = None
    times = []
    for r in range(reps + 3):
        dp = _to_dev(p)
        st = eng.new_adam_state(dp)
        loss = torch.zeros(B, device="cuda")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.fit(T, dp, st, act, n, loss_out=loss)
        e1.record()
        torch.cuda.synchronize()
        if r >= 3:
            times.append(e0.elapsed_time(e1))
        out = (_to_host(dp), loss.cpu().numpy(), _to_host(st.m))
    eng.close()
    return out, float(np.median(times))

if True:
  for shape, C, kpd, B, kw in [((16, 16), 1, [2, 2], 1024, {}), ((16, 16), 1, [2, 2], 1536, {}), ((16, 16), 1, [2, 2], 2048, {}),
                               ((16, 16), 3, [2, 2], 1024, {}), ((32, 32), 3, [2, 4], 1020, {}), ((16, 16, 4), 3, [2, 2, 1], 1020, {}),
                             ((16, 12), 1, [2, 2], 777, dict(pis_l1=0.2, u_l1=0.003))]:
      (pa, la, ma), ta = run(shape, C, kpd, B, 64, **kw)
      (pb, lb, mb), tb = run(shape, C, kpd, B, 128, **kw)
      (_, l1a, m1a), _ = run(shape, C, kpd, B, 64, n=1, reps=0, **kw)
      (_, l1b, m1b), _ = run(shape, C, kpd, B, 128, n=1, reps=0, **kw)
      N = int(np.prod(shape))
      dev = max(np.abs(m1a[k] - m1b[k]).max() / (np.abs(m1a[k]).max() + 1e-30) for k in m1a)
      print(f"{shape} C{C} K{int(np.prod(kpd))} B={B}: g64 {ta:.3f} ms ({B*N*100/ta/1e6:.1f} G)  pair {tb:.3f} ms ({B*N*100/tb/1e6:.1f} G)"
            f"  one step: max rel gradient dev {dev:.2e}  loss dev {np.abs(l1a-l1b).max():.2e}", flush=True)
