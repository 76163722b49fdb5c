"""Evaluation pass (smoe_forward) at the headline size: time and effective HBM rate per requested output set and tiling."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from steered_mixture_of_experts_amd import blocks as blk                                    # noqa: E402
from steered_mixture_of_experts_amd.engine import BlockEngine, EngineConfig                 # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _warm import warm_block_engine                                                                       # noqa: E402


def main():
    B, shape, C, kpd = 65536, (16, 16), int(os.environ.get("C", "1")), [2, 2]
    K, N = 4, 256
    blocks = blk.synthetic_blocks(B, shape, C, 7)
    p0 = blk.init_block_params(blocks, kpd)
    for tiling in (16, 32, 64):
        eng = BlockEngine(EngineConfig(block_shape=shape, channels=C, kernels=K, use_yuv=(C == 3), quantize_pis=True))
        eng.set_tiling(tiling)
        T = torch.from_numpy(blk.to_planar(blocks)).cuda()
        dp = {k: torch.from_numpy(v).cuda() for k, v in p0.items()}
        act = torch.full((B,), 15, dtype=torch.int32, device="cuda")
        for name, kw in (("loss only", dict(want_recon=False)),
                         ("recon", dict(want_recon=True)),
                         ("recon+argmax+gate", dict(want_recon=True, want_argmax=True, want_gate=True))):
            for _ in range(3):
                eng.forward(T, dp, act, **kw)
            torch.cuda.synchronize()
            warm_block_engine(eng, T, dp, act, iters=600, other_tiling=False, restore_tiling=tiling)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                eng.forward(T, dp, act, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            byt = B * N * C * 4 + B * K * 14 * 4
            if kw.get("want_recon"):
                byt += B * N * C * 4
            if kw.get("want_gate"):
                byt += B * N * K * 4 + B * N
            print(json.dumps({"tiling": tiling, "C": C, "outputs": name, "ms": round(ms, 4), "MB": round(byt / 1e6, 1),
                              "GBps": round(byt / ms / 1e6, 1)}))
        eng.close()


if __name__ == "__main__":
    main()
