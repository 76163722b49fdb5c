import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_parity import _engine, _setup, _to_dev, _planar
shape, C, kpd, B = (16, 16), 1, [2, 2], 65536
cfg, p, coords, tgt, K = _setup(shape, C, kpd, False, B, 7, quantize_pis=True)
eng = _engine(shape, C, K, quantize_pis=True)
T = _planar(tgt)
act = torch.full((B,), 15, dtype=torch.int32, device="cuda")
dp = _to_dev(p); st = eng.new_adam_state(dp)
for _ in range(30): eng.fit(T, dp, st, act, 100)     # clocks
torch.cuda.synchronize()
for n in (1, 2, 5, 10, 20, 50, 100):
    ts = []
    for r in range(7):
        dp = _to_dev(p); st = eng.new_adam_state(dp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.fit(T, dp, st, act, n); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(n, "iters:", round(float(np.median(ts)) * 1000, 1), "us", flush=True)
import time
dp = _to_dev(p); st = eng.new_adam_state(dp)
torch.cuda.synchronize()
hs = []
for r in range(20):
    t0 = time.perf_counter(); eng.fit(T, dp, st, act, 1); hs.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print("host time of one eng.fit call: median", round(float(np.median(hs)) * 1e6, 1), "us  min", round(min(hs) * 1e6, 1), "iters")
# GPU time of a 1-iteration launch when the queue is kept busy (launch latency hidden): 50 back-to-back launches
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for r in range(50): eng.fit(T, dp, st, act, 1)
e1.record(); torch.cuda.synchronize()
print("back-to-back 1-iteration launches:", round(e0.elapsed_time(e1) / 50 * 1000, 1), "us each iters")
