import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R,'tests'))
import numpy as np, torch
from oracle import smoe_oracle as o
from steered_mixture_of_experts_amd.blocks import synthetic_blocks
from test_gpu_parity import _setup, _engine, _to_dev, _to_host, _planar, _mask_to_bits, SHAPES
np.set_printoptions(precision=2, linewidth=220)
for (shape, C, kpd, yuv) in SHAPES[:2]:
    B = 21
    cfg, p, coords, tgt, K = _setup(shape, C, kpd, yuv, B, 300 + len(shape) + C)
    active = np.ones((B, K), dtype=bool)
    for tiling in (16, 64):
        eng = _engine(shape, C, K, use_yuv=yuv); eng.set_tiling(tiling)
        dp = _to_dev(p); act = torch.from_numpy(_mask_to_bits(active).view(np.int32)).cuda(); T = _planar(tgt)
        fw = eng.forward(T, dp, act, want_recon=True, update_active=False)
        recon = np.transpose(fw["recon"].cpu().numpy(), (0, 2, 1))
        ref64 = o.forward(p, tgt, coords, active, cfg, None, np.float64, want_grads=True, q_override=recon)
        refn = o.forward(p, tgt, coords, active, cfg, None, np.float64, want_grads=True)
        state = eng.new_adam_state(dp)
        loss = torch.zeros(B, device="cuda"); sse = torch.zeros(B, device="cuda")
        eng.fit(T, dp, state, act, 1, loss_out=loss, sse_out=sse)
        torch.cuda.synchronize()
        m = _to_host(state.m)
        print(shape, C, 'tiling', tiling)
        for name in ("nu_e", "pis", "A_diagonal"):
            scale = np.abs(ref64["grads"][name]).reshape(B,-1).max(1) + 1e-30
            eg = np.abs(m[name] / 0.1 - ref64["grads"][name]).reshape(B,-1).max(1) / scale
            print('  ', name, 'per-block rel err:', eg)
        print('   sse gpu-fit vs oracle(q=fwd recon):', (sse.cpu().numpy()-ref64['sse'])/ (1/255.)**2)
        print('   loss rel diff:', (loss.cpu().numpy()-ref64['loss'])/ref64['loss'])
        y=ref64['y']; print('   n(y<0|y>1):', ((y<0)|(y>1)).sum(axis=(1,2)))
        eng.close()
